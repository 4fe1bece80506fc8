#!/bin/bash
# Build tuning variants of libFL.so (bench instantiation only) into fortran-library_amd/lib/variants/.
# usage: tools/variants.sh name1:"-DFLAG ..." name2:"..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/fortran-library_amd
mkdir -p $P/lib/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DFL_ONLY_BENCH $flags \
     -shared -o $P/lib/variants/libFL_$name.so $P/csrc/fl_solver_kernels.hip $P/csrc/fl_aux_kernels.hip &
done
wait
ls -la $P/lib/variants
