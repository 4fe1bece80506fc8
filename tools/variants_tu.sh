#!/bin/bash
# Build tuning variants of libFL.so in which ONE translation unit is recompiled with extra flags and every other object
# is the regular build's (make -C fortran-library_amd first).  For kernels outside bench.py's headline instantiation
# (tools/variants.sh covers that one): e.g. the one-wave geometry that runs BASELINE config 5.
#   tools/variants_tu.sh fl_solver_g18 name1:"-DFL_SPEC_K=1" name2:"-DFL_SPEC_K=2" ...
# -> fortran-library_amd/lib/variants/libFL_<name>.so ; run with FL_LIBRARY=<that file> (tools/ab_cfg.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/fortran-library_amd
TU=$1; shift
mkdir -p $P/lib/variants /tmp/flvar
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
OTHERS=$(ls $P/csrc/*.o | grep -v "/$TU.o")
build() {
  name=$1; flags=$2; d=/tmp/flvar/$name; mkdir -p $d
  /opt/rocm/bin/hipcc $FLAGS $flags -c $P/csrc/$TU.hip -o $d/$TU.o 2>$d/$TU.err || { echo "FAILED $name"; grep error $d/$TU.err | head -3; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/variants/libFL_$name.so $d/$TU.o $OTHERS 2>$d/link.err || { echo "LINK FAILED $name"; head -3 $d/link.err; }
}
for spec in "$@"; do
  build "${spec%%:*}" "${spec#*:}" &
done
wait
ls $P/lib/variants
