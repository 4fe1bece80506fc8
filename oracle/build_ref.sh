#!/bin/bash
# oracle/build_ref.sh -- builds oracle/_ref/libfl_ref_la.so: the REFERENCE's own LinearAlgebra module, compiled
# UNMODIFIED from where it lies (/root/reference/source/LinearAlgebra.f90: no #include, no dependency besides
# LAPACK/BLAS), linked against the real MKL runtime of this image (/opt/conda/lib/libmkl_rt.so), plus the bind(C)
# doors of oracle/ref_la_capi.f90.  Outputs go to oracle/_ref/ only (git-ignored; they travel to the GPU box with
# gpurun like any built .so).  Test infrastructure: tools/make_la_golden.py and tests/ use it, the product never.
#
# NonlinearOptimization.f90 is NOT built: its line 15 #includes Intel's closed mkl_rci.f90, absent from this image,
# and no stand-in is written for it.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
ref="${FL_REFERENCE:-/root/reference}"
src="$ref/source/LinearAlgebra.f90"
out="$here/_ref"
mkl="${FL_MKL_DIR:-/opt/conda/lib}"
fc="${FC:-amdflang}"
if [ ! -f "$src" ]; then echo "build_ref: $src not present (GPU box: the prebuilt oracle/_ref travels)"; exit 0; fi
if ! command -v "$fc" >/dev/null || [ ! -e "$mkl/libmkl_rt.so" ]; then echo "build_ref: $fc or $mkl/libmkl_rt.so missing"; exit 0; fi
mkdir -p "$out"
if [ "$out/libfl_ref_la.so" -nt "$src" ] && [ "$out/libfl_ref_la.so" -nt "$here/ref_la_capi.f90" ]; then exit 0; fi
cd "$out"
"$fc" -cpp -O2 -fPIC -w -c "$src" -o LinearAlgebra.o            # writes linearalgebra.mod next to it
"$fc" -O2 -fPIC -c "$here/ref_la_capi.f90" -o ref_la_capi.o
"$fc" -shared -o libfl_ref_la.so ref_la_capi.o LinearAlgebra.o -L"$mkl" -lmkl_rt -Wl,-rpath,"$mkl"
echo "build_ref: $out/libfl_ref_la.so"
