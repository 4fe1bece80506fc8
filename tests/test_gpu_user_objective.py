"""-m gpu: a CALLER-COMPILED objective inside the fused solver kernels (include/fl_user_objective.hpp, the GPU form of the
reference's "pass your own f, fd": NO.f90:33-38).  tests/user_objective_caller.hip restates the diagonal quadratic as a user
functor, is compiled here with hipcc against the installed headers + libFL.so, and must reproduce the built-in
FL_OBJ_DIAGQUAD bit for bit (minimiser, objective, g.g, iteration / evaluation counts, status) for L-BFGS, CG, SD and
quasi-Newton BFGS in two geometries; a geometry that does not belong to n is refused."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_user_functor_equal_to_diagquad_reproduces_the_builtin_bit_for_bit():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = os.path.join(ROOT, "fortran-library_amd", "lib")
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "user_objective_caller")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           os.path.join(ROOT, "tests", "user_objective_caller.hip"), "-o", exe, "-L" + lib, "-lFL",
                           "-Wl,-rpath," + lib])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = p.stdout.decode()
    assert p.returncode == 0 and "ALL OK" in text, text
    assert text.count("reproduces bit for bit") == 5, text


def test_streaming_functor_beyond_n_4096_reproduces_the_builtin_bit_for_bit():
    """include/fl_user_stream_objective.hpp (included twice around the class): the vectors-in-HBM kernel around a caller's
    objective that is asked one element pair at a time"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = os.path.join(ROOT, "fortran-library_amd", "lib")
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "user_stream_objective_caller")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           os.path.join(ROOT, "tests", "user_stream_objective_caller.hip"), "-o", exe, "-L" + lib, "-lFL",
                           "-Wl,-rpath," + lib])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = p.stdout.decode()
    assert p.returncode == 0 and "ALL OK" in text, text
    assert text.count("reproduces bit for bit") == 4, text
