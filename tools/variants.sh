#!/bin/bash
# Build tuning variants of libFL.so (bench instantiation only) into fortran-library_amd/lib/variants/.
# usage: tools/variants.sh name1:"-DFLAG ..." name2:"..."
# fl_rci.hip does not depend on the tuning flags: the object of the regular build is reused.
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/fortran-library_amd
mkdir -p $P/lib/variants /tmp/flvar
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math ${FL_FULL:--DFL_ONLY_BENCH}"  # FL_FULL=" " builds every instantiation
build() {
  name=$1; flags=$2; d=/tmp/flvar/$name; mkdir -p $d
  for src in fl_solver_kernels fl_aux_kernels fl_bfgs_gemm fl_dense_kernels; do
    /opt/rocm/bin/hipcc $FLAGS $flags -c $P/csrc/$src.hip -o $d/$src.o 2>$d/$src.err || { echo "FAILED $name $src"; grep error $d/$src.err | head -3; }
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/variants/libFL_$name.so $d/*.o $P/csrc/fl_rci.o $P/csrc/fl_blas_kernels.o $P/csrc/fl_chol_blocked.o $P/csrc/fl_general.o $P/csrc/fl_linalg.o 2>$d/link.err || { echo "LINK FAILED $name"; head -3 $d/link.err; }
}
for spec in "$@"; do
  build "${spec%%:*}" "${spec#*:}" &
done
wait
ls $P/lib/variants
