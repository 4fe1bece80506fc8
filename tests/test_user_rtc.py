"""CPU check of the run-time compiled objectives (csrc/fl_user_rtc.hip): the kernel's headers embedded in libFL.so and a
caller's functor compile with hiprtc for gfx950 -- no GPU and nothing of the source tree needed -- for every solver and both
kinds of objective (element-wise; neighbour-coupled with LDS and barriers); a source that does not compile comes back as
FL_ERR_INVALID_ARGUMENT with the compiler's message pointing into the caller's text."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import user_sources as US


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


@pytest.mark.parametrize("name,src,cls,solver,n,tune", [
    ("quadratic L-BFGS 2x8", US.DIAGQUAD, "MyQuadratic", 2, 1024, 2), ("quadratic CG 1x16", US.DIAGQUAD, "MyQuadratic", 1, 1024, 2),
    ("quadratic SD 1x8", US.DIAGQUAD, "MyQuadratic", 0, 300, 2), ("quadratic BFGS 1x4", US.DIAGQUAD, "MyQuadratic", 3, 256, 2),
    ("Rosenbrock L-BFGS 1x4", US.ROSENBROCK, "MyRosenbrock", 2, 256, 4), ("Rosenbrock L-BFGS 8x8", US.ROSENBROCK, "MyRosenbrock", 2, 4096, 4),
    # n > 4096: the streaming functor in the vectors-in-HBM kernel
    ("stream quadratic L-BFGS", US.STREAM_DIAGQUAD, "MyBigQuadratic", 2, 5000, 4), ("stream quadratic BFGS", US.STREAM_DIAGQUAD, "MyBigQuadratic", 3, 6001, 4),
    ("stream Rosenbrock CG", US.STREAM_ROSENBROCK, "MyBigRosenbrock", 1, 100000, 4), ("stream Rosenbrock SD", US.STREAM_ROSENBROCK, "MyBigRosenbrock", 0, 4097, 4)])
def test_objective_sources_compile_for_gfx950_without_a_gpu(name, src, cls, solver, n, tune):
    rc, log = _nlo().compile_check(src, cls, n, solver, tune)
    assert rc == 0, (name, log[:2000])


def test_a_source_that_does_not_compile_is_reported_with_the_compilers_message():
    NLO = _nlo()
    rc, log = NLO.compile_check(US.BROKEN, "Oops", 256)
    assert rc == -1 and "undeclared_name" in log and "objective:" in log, log[:1000]
    assert NLO.compile_check(US.DIAGQUAD, "MyQuadratic", 5000)[0] == -1      # beyond the register path: the streaming functor's interface
    assert NLO.compile_check(US.STREAM_DIAGQUAD, "MyBigQuadratic", 20000, solver=3)[0] == -2  # dense BFGS: H up to n = 16384
    assert NLO.compile_check(US.DIAGQUAD, "MyQuadratic", 256, solver=4)[0] == -1  # NewtonRaphson needs a Hessian functor
    assert NLO.compile_check(US.DIAGQUAD, "NoSuchClass", 256)[0] == -1


def test_compiled_code_objects_are_cached_on_disk(tmp_path, monkeypatch):
    NLO = _nlo()
    monkeypatch.setenv("FL_RTC_CACHE_DIR", str(tmp_path))
    assert NLO.compile_check(US.DIAGQUAD, "MyQuadratic", 512, 2, 2)[0] == 0
    files = list(tmp_path.glob("fl_user_*.co"))
    assert len(files) == 1 and files[0].stat().st_size > 10000
    stamp = files[0].stat().st_mtime_ns
    assert NLO.compile_check(US.DIAGQUAD, "MyQuadratic", 512, 2, 2)[0] == 0  # (served from the file)
    assert files[0].stat().st_mtime_ns == stamp and len(list(tmp_path.glob("fl_user_*"))) == 1
    assert NLO.compile_check(US.DIAGQUAD + "\n// changed\n", "MyQuadratic", 512, 2, 2)[0] == 0
    assert len(list(tmp_path.glob("fl_user_*.co"))) == 2


def test_constraints_functor_sources_compile_too():
    """(through the Python wrapper's GPU-less path there is only the plain kernel: compile the constrained program by hand)"""
    import ctypes as C
    NLO = _nlo()
    NLO.FL.fl_user_compile_check_auglag.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
    log = C.create_string_buffer(1 << 16)
    for cons in (b"MySpheres", b""):
        rc = NLO.FL.fl_user_compile_check_auglag((US.DIAGQUAD + US.BLOCK_SPHERES).encode(), b"MyQuadratic", cons, 2, 512, 2, b"gfx950", log, len(log))
        assert rc == 0, log.value.decode()[:2000]
