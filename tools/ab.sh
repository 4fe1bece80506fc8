#!/bin/bash
# Run bench.py once per tuning variant on the GPU box, interleaved, same process conditions.
# usage: tools/ab.sh [extra bench.py arguments, e.g. --workload lbfgs_rosen256]; the in-tree libFL.so runs as "intree".
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for so in $R/fortran-library_amd/lib/libFL.so $R/fortran-library_amd/lib/variants/libFL_*.so; do
  FL_LIBRARY=$so python $R/bench.py --cpu-sample 0 --no-two-loop --steps 3 --warmup 1 "$@" 2>/tmp/ab_err.txt | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$(basename $so)', round(r['value']/1e6,2), 'Mit/s', round(r['ms_per_step'],1), 'ms')"
done; done
