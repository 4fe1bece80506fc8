"""The binding stubs INTEGRATION.md shows to a maintainer of the reference are compiled, not just printed.

Round 1 documented a Fortran `fl_options` one field short of the C struct (fl_default_options would have written 4
bytes past it).  Here the Fortran block of section 1 is compiled with amdflang and linked against libFL.so, the C++
block of section 2 is compiled with g++ against include/fl_nlopt.h, and the size of the Fortran derived type is
compared with the C struct's.  Reference interfaces the stubs stand for: NonlinearOptimization.f90:398-400,
cpp/NonlinearOptimization.hpp:278-392."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = open(os.path.join(ROOT, "INTEGRATION.md")).read()
LIBDIR = os.path.join(ROOT, "fortran-library_amd", "lib")


def _block(lang, nth=0):
    return re.findall(r"```" + lang + r"\n(.*?)```", DOC, flags=re.S)[nth]


def _c_sizeof(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "fl_nlopt.h"\nint main(void){printf("%zu\\n", sizeof(fl_options));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    return int(subprocess.check_output([str(exe)]).decode())


@pytest.mark.skipif(shutil.which("amdflang") is None, reason="no Fortran compiler")
def test_fortran_stub_compiles_links_and_has_the_c_struct_size(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libFL.so")):
        pytest.skip("libFL.so not built")
    (tmp_path / "stub.f90").write_text(_block("fortran") + """
program size_check
    use iso_c_binding
    use NonlinearOptimization_MI355X
    implicit none
    type(fl_options)::o
    call fl_default_options(o,2)   ! writes every field of the C struct: must fit the Fortran type
    print '(I0,1X,I0,1X,I0)', c_sizeof(o), o%memory, o%exact_step
end program
""")
    exe = tmp_path / "stub"
    subprocess.check_call(["amdflang", "-O1", "stub.f90", "-o", str(exe), "-L", LIBDIR, "-lFL", "-Wl,-rpath," + LIBDIR,
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], cwd=tmp_path)
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == _c_sizeof(tmp_path) == 72
    assert (int(out[1]), int(out[2])) == (10, 20)  # Memory=10 (NO.f90:420), ExactStep=20 (NO.f90:653)


def test_cpp_stub_compiles_against_the_header(tmp_path):
    (tmp_path / "stub.cpp").write_text(_block("cpp") + "\nint main() { return 0; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I", os.path.join(ROOT, "include"), str(tmp_path / "stub.cpp")])


def test_fortran_shim_options_type_matches_too():
    """the shipped shim (fortran-library_amd/fortran/NonlinearOptimization.f90) declares the same twelve fields"""
    shim = open(os.path.join(ROOT, "fortran-library_amd", "fortran", "NonlinearOptimization.f90")).read()
    m = re.search(r"type,\s*bind\(C\)\s*::\s*fl_options(.*?)end type", shim, flags=re.S | re.I)
    assert m
    fields = re.findall(r"::[ \t]*([\w, \t]+)", m.group(1))
    names = [x.strip() for f in fields for x in f.split(",") if x.strip()]
    assert names == ["strong", "max_iteration", "precision", "min_step_length", "wolfe_c1", "wolfe_c2", "increment",
                     "memory", "cg_method", "fused_f_fd", "clamp", "exact_step"]
