#!/bin/bash
# Run tools/bench_configs.py <configs...> once per tuning variant on the GPU box (two interleaved rounds), no CPU leg.
# usage: tools/ab_cfg.sh c5 [c2 ...]     -- the in-tree libFL.so runs as libFL.so, variants as libFL_<name>.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for so in $R/fortran-library_amd/lib/libFL.so $R/fortran-library_amd/lib/variants/libFL_*.so; do
  FL_LIBRARY=$so python $R/tools/bench_configs.py "$@" --cpu-seconds 0 2>/tmp/ab_err.txt | python -c "
import json,sys
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    r=json.loads(ln)
    rate = r.get('inner_iterations_per_s') or r.get('iterations_per_s') or 0
    print('$(basename $so)', r['config'][:28], round(r['ms'],2), 'ms', round(rate/1e6,2), 'Mit/s', 'f_evals', r.get('f_evals'), 'it', r.get('inner_iterations', r.get('iterations')), r.get('f_evals_per_problem_min_mean_max',''))"
done; done
