#!/bin/bash
# libFL_timers.so: the fused kernels of the geometries named (default: 1x8) with FL_PHASE_TIMERS
# (csrc/fl_solver_launch.hpp) -- per-problem time by phase into the buffer FL_PHASE_BUFFER names; tools/phase_timers.py reads it.
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/fortran-library_amd
TUS=${@:-fl_solver_g18}
mkdir -p $P/lib/variants /tmp/flvar/timers
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DFL_PHASE_TIMERS"
OTHERS=$(ls $P/csrc/*.o)
for tu in fl_solver_kernels $TUS; do
  OTHERS=$(echo "$OTHERS" | grep -v "/$tu.o")
  /opt/rocm/bin/hipcc $FLAGS -c $P/csrc/$tu.hip -o /tmp/flvar/timers/$tu.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $P/lib/variants/libFL_timers.so /tmp/flvar/timers/*.o $OTHERS
ls -la $P/lib/variants/libFL_timers.so
