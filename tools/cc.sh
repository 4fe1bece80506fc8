#!/bin/bash
# compile the named translation units of fortran-library_amd/csrc in parallel (objects in place) and print their kernels' resources
# usage: tools/cc.sh fl_solver_g18r fl_solver_g14r [-- extra flags]
P=/root/repo/fortran-library_amd
TUS=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; TUS+=("$1"); shift; done
for tu in "${TUS[@]}"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-variable "${EXTRA[@]}" -c $P/csrc/$tu.hip -o $P/csrc/$tu.o > /tmp/cc_$tu.log 2>&1 || { echo "FAILED $tu"; grep -E "error" -A4 /tmp/cc_$tu.log | head -40; } ) &
done
wait
for tu in "${TUS[@]}"; do [ -f $P/csrc/$tu.o ] && python /root/repo/tools/kernel_resources.py $P/csrc/$tu.o --demangle | cut -c1-150; done
