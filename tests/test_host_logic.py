"""CPU tests of the product's host-side logic (no GPU):
  * libFL.so loads and exports every symbol include/fl_nlopt.h declares;
  * the line-search state machine (fl_linesearch.hpp, compiled for the host) takes exactly
    the trials of the oracle's structured restatement of Wolfe / StrongWolfe(+_fdwithf)
    over thousands of randomised searches, including bad step guesses that force the
    shrink, grow, zoom and bail-out branches.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "fl_nlopt.h")).read()
    names = sorted(set(re.findall(r"\b(fl_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 9
    so = os.path.join(ROOT, "fortran-library_amd", "lib", "libFL.so")
    assert os.path.exists(so), "build it: make -C fortran-library_amd"
    lib = C.CDLL(so)
    for nme in names:
        assert hasattr(lib, nme), nme
    assert lib.fl_version() >= 100
    # the reference's own mangled entry points (include/fl_legacy.h; cpp/NonlinearOptimization.hpp:278-393)
    leg = open(os.path.join(ROOT, "include", "fl_legacy.h")).read()
    lnames = sorted(set(re.findall(r"\b(__nonlinearoptimization_MOD_[a-z_]+|nonlinearoptimization_mp_[a-z_]+_)\s*\(", leg)))
    assert len(lnames) == 28  # 10 routines + 4 line searchers, gfortran and ifort manglings
    for nme in lnames:
        assert hasattr(lib, nme), nme
    # LinearAlgebra symbols the reference's C++ header binds (cpp/FortranLibrary.hpp:48-63)
    for nme in ("__linearalgebra_MOD_my_dgemm", "__linearalgebra_MOD_my_dgemm_t", "__linearalgebra_MOD_my_dsyev",
                "linearalgebra_mp_my_dgemm_t_", "linearalgebra_mp_my_dsyev_", "fl_dsysv_batched", "fl_dsyev_values",
                "fl_dsyev_vectors", "fl_dsyev_vectors_workspace_bytes", "fl_dsyev_jacobi"):
        assert hasattr(lib, nme), nme
    # import-time symbols of the reference's Python package (FortranLibrary/General.py:4-16)
    for nme in ("__general_MOD_showtime", "general_mp_showtime_", "__general_MOD_dscientificnotation", "general_mp_dscientificnotation_"):
        assert hasattr(lib, nme), nme
    xv, iv = C.c_double(12345.678), C.c_int(7)
    lib.__general_MOD_dscientificnotation(C.byref(xv), C.byref(iv))
    assert iv.value == 4 and abs(xv.value - 1.2345678) < 1e-12
    xv = C.c_double(0.00042)
    lib.general_mp_dscientificnotation_(C.byref(xv), C.byref(iv))
    assert iv.value == -4 and abs(xv.value - 4.2) < 1e-12
    t, e, t2, e2 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for n, (T, E) in {10: (64, 2), 256: (64, 4), 512: (64, 8), 1024: (128, 8), 2048: (256, 8), 4096: (512, 8)}.items():
        assert lib.fl_reduction_geometry(n, C.byref(t), C.byref(e)) == 0 and (t.value, e.value) == (T, E)
    # beyond the register path: 1024 threads, 2*ceil(ceil(n/2)/1024) slots per thread (csrc/fl_big.hpp)
    # the fused SD / CG kernels: one wave x 16 elements for 512 < n <= 1024, everything else as the layout geometry; the
    # padded row length threads*ept never depends on the solver
    for solver in range(5):
        for n in (10, 256, 257, 400, 512, 513, 700, 1024, 1025, 2048, 2049, 4096, 5000):
            assert lib.fl_reduction_geometry(n, C.byref(t), C.byref(e)) == 0
            assert lib.fl_reduction_geometry_for(solver, n, C.byref(t2), C.byref(e2)) == 0
            want = (t.value // 2, 16) if solver in (0, 1) and 512 < n <= 4096 else ((128, 4) if solver == 4 and 256 < n <= 512 else (t.value, e.value))
            assert (t2.value, e2.value) == want and t2.value * e2.value == t.value * e.value
    assert lib.fl_reduction_geometry(100000, C.byref(t), C.byref(e)) == 0 and (t.value, e.value) == (1024, 98)
    assert lib.fl_reduction_geometry(4097, C.byref(t), C.byref(e)) == 0 and (t.value, e.value) == (1024, 6)
    assert lib.fl_reduction_geometry((1 << 27) + 1, C.byref(t), C.byref(e)) == -2
    lib.fl_workspace_bytes.restype = C.c_size_t
    assert lib.fl_workspace_bytes(2, 65536, 1024, 10) == 65536 * 20 * 1024 * 8
    assert lib.fl_workspace_bytes(2, 3, 5000, 10) == 3 * (20 + 4) * 6144 * 8  # ring + the four vector rows


def _driver():
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libls_driver.so")
    src = os.path.join(ROOT, "tests", "ls_machine_driver.cpp")
    hpp = os.path.join(ROOT, "fortran-library_amd", "csrc", "fl_linesearch.hpp")
    O.lib()
    olib = os.path.join(ROOT, "oracle", "libfl_oracle.so")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hpp), os.path.getmtime(olib)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so,
                               "-L" + os.path.dirname(olib), "-lfl_oracle", "-Wl,-rpath," + os.path.dirname(olib)])
    return C.CDLL(so)


@pytest.mark.parametrize("strong,fused", [(1, 0), (1, 1), (0, 0)])
@pytest.mark.parametrize("kind", [O.QUARTIC, O.ROSENBROCK, O.DIAGQUAD])
def test_line_search_machine_equals_oracle(strong, fused, kind):
    drv = _driver()
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(100 * strong + 10 * fused + kind)
    n = 24
    args = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, dp, dp,
            dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    drv.ls_machine_run.argtypes = args
    drv.ls_oracle_run.argtypes = args
    states = set()
    for trial in range(1500):
        d = 1.0 + rng.uniform(0, 99, n)
        b = rng.uniform(-1, 1, n)
        x = rng.uniform(-1.5, 1.5, n) if kind != O.ROSENBROCK else 1 + rng.uniform(-0.5, 0.5, n)
        r = O.solve_batch(O.SD, kind, x, d=d, b=b, opts=O.defaults(maxit=0))  # f, via the oracle objective
        # gradient by central differences is not needed: use a descent direction from the objective itself
        P = O.solve_batch  # noqa: F841
        g = _grad(kind, x, d, b)
        p = -g * rng.uniform(0.5, 1.5, n) if trial % 3 else -g
        phid0 = float(g @ p)
        fx0 = float(r["f"][0])
        a0 = float(10.0 ** rng.uniform(-6, 3))  # from far too short to far too long
        c1, c2, incr = 1e-4, (0.9 if trial % 2 else 0.45), float(rng.choice([1.05, 1.5, 2.0]))
        outs = []
        for fn in (drv.ls_machine_run, drv.ls_oracle_run):
            xx = x.copy()
            gg = np.zeros(n)
            a = C.c_double(a0)
            fx = C.c_double(fx0)
            nf, ng = C.c_int(0), C.c_int(0)
            fn(strong, fused, c1, c2, incr, kind, n, xx.ctypes.data_as(dp), p.ctypes.data_as(dp), C.byref(a),
               C.byref(fx), phid0, d.ctypes.data_as(dp), b.ctypes.data_as(dp), gg.ctypes.data_as(dp), C.byref(nf),
               C.byref(ng))
            outs.append((a.value, fx.value, nf.value, ng.value, xx, gg))
        m, o = outs
        assert m[:4] == o[:4], (trial, m[:4], o[:4])
        assert np.array_equal(m[4], o[4]) and np.array_equal(m[5], o[5])
        states.add((m[2] > 3, m[3] > 2))
    assert len(states) >= 2  # short and long searches both exercised


def _grad(kind, x, d, b):
    if kind == O.QUARTIC:
        return 4 * x ** 3
    if kind == O.DIAGQUAD:
        return d * x - b
    g = np.zeros_like(x)
    u = x[1:] - x[:-1] ** 2
    g[:-1] += -400 * x[:-1] * u - 2 * (1 - x[:-1])
    g[1:] += 200 * u
    return g


def test_strong_wolfe_quirk_path_is_exercised_and_matches():
    """StrongWolfe (no f_fd) keeps looping after zoom in its grow branch (NO.f90:1507-1514) while
    StrongWolfe_fdwithf returns (1628-1632): with Increment = 3, c2 = 0.1 the two variants end differently in
    hundreds of searches -- and in every one the machine equals the oracle's structured restatement."""
    drv = _driver()
    dp = C.POINTER(C.c_double)
    args = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, dp, dp,
            dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    drv.ls_machine_run.argtypes = args
    drv.ls_oracle_run.argtypes = args
    rng = np.random.default_rng(0)
    n, differ = 24, 0
    for kind in (O.QUARTIC, O.DIAGQUAD):
        for _ in range(600):
            d = 1 + rng.uniform(0, 99, n)
            b = rng.uniform(-1, 1, n)
            x = rng.uniform(-1.5, 1.5, n)
            g = _grad(kind, x, d, b)
            p = -g * rng.uniform(0.5, 1.5, n)
            fx0 = float(O.solve_batch(O.SD, kind, x, d=d, b=b, opts=O.defaults(maxit=0))["f"][0])
            a0 = float(10.0 ** rng.uniform(-6, 0))
            res = []
            for fn, fused in ((drv.ls_machine_run, 0), (drv.ls_oracle_run, 0), (drv.ls_machine_run, 1), (drv.ls_oracle_run, 1)):
                xx, gg = x.copy(), np.zeros(n)
                a, fx, nf, ng = C.c_double(a0), C.c_double(fx0), C.c_int(0), C.c_int(0)
                fn(1, fused, 1e-4, 0.1, 3.0, kind, n, xx.ctypes.data_as(dp), p.ctypes.data_as(dp), C.byref(a), C.byref(fx),
                   float(g @ p), d.ctypes.data_as(dp), b.ctypes.data_as(dp), gg.ctypes.data_as(dp), C.byref(nf), C.byref(ng))
                res.append((a.value, fx.value, nf.value, ng.value, xx.tobytes()))
            assert res[0] == res[1] and res[2] == res[3]
            differ += res[0][:2] != res[2][:2]
    assert differ > 50


def test_reference_python_package_imports_against_this_library():
    """Drop-in at the Python boundary: the reference's own ctypes package does CDLL('libFL.so') and
    probes __general_MOD_showtime when imported (FortranLibrary/General.py:4-7).  With this build's lib directory on
    the loader path it imports and its General functions work.  Only where the reference checkout is mounted."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "FortranLibrary")):
        pytest.skip("reference checkout not mounted (GPU box)")
    lib = os.path.join(ROOT, "fortran-library_amd", "lib")
    env = dict(os.environ, PYTHONPATH=ref, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    code = "import FortranLibrary as F; F.ShowTime(); print('SN', *F.dScientificNotation(1234.5))"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd="/tmp")
    assert out.returncode == 0, out.stderr
    assert " year " in out.stdout and "SN 1.2345" in out.stdout and out.stdout.strip().endswith(" 3")


def test_augmented_lagrangian_launch_plan_by_batch():
    """include/fl_nlopt.h fl_augmented_lagrangian_launch_plan: helper waves by batch for the start of the launch, unfinished
    problems handed to stages with more waves; only where the kernel has helper-wave forms; independent of a GPU being present
    (256 CUs assumed without one)."""
    import FortranLibrary.NonlinearOptimization as NLO
    P = NLO.augmented_lagrangian_launch_plan
    assert P(NLO.LBFGS_, NLO.DIAGQUAD, 256, 512, 8) == [(4, 0)]
    assert P(NLO.LBFGS_, NLO.DIAGQUAD, 1024, 512, 8) == [(2, 512), (4, 0)]
    assert P(NLO.LBFGS_, NLO.DIAGQUAD, 8192, 512, 8) == [(1, 1536), (2, 512), (4, 0)]
    assert P(NLO.CG, NLO.QUARTIC, 8192, 256, 8) == [(1, 1536), (2, 512), (4, 0)]
    # no helper-wave form: another objective, a dense inner solver, two waves per problem, blocks that are not lane groups
    assert P(NLO.LBFGS_, NLO.ROSENBROCK, 8192, 512, 8) == [(1, 0)]
    assert P(NLO.BFGS_, NLO.DIAGQUAD, 8192, 512, 8) == [(1, 0)]
    assert P(NLO.LBFGS_, NLO.DIAGQUAD, 8192, 1024, 8) == [(2, 0)]
    assert P(NLO.LBFGS_, NLO.DIAGQUAD, 8192, 200, 5) == [(1, 0)]
    os.environ["FL_AUG_STAGED"] = "0"
    try:
        assert P(NLO.LBFGS_, NLO.DIAGQUAD, 8192, 512, 8) == [(1, 0)] and P(NLO.LBFGS_, NLO.DIAGQUAD, 1024, 512, 8) == [(2, 0)]
    finally:
        del os.environ["FL_AUG_STAGED"]


def test_update_form_and_cooperative_queries_answer_without_a_device():
    """fl_bfgs_deferred_updates: the fused BFGS kernels keep 8 rank-2 updates pending for n > 128 (also beyond the register path, up
    to the dense H's n = 16384), none for n <= 128; fl_cooperative_groups_for: 1 where no device answers (the cooperative form
    needs the CU count) and for everything that never shares a problem"""
    lib = C.CDLL(os.path.join(ROOT, "fortran-library_amd", "lib", "libFL.so"))
    f = lib.fl_bfgs_deferred_updates
    f.argtypes = [C.c_int]
    assert [f(n) for n in (0, 1, 10, 128, 129, 256, 1024, 4096, 4097, 16384, 16385)] == [0, 0, 0, 0, 8, 8, 8, 8, 8, 8, 0]
    g = lib.fl_cooperative_groups_for
    g.argtypes = [C.c_int] * 4
    assert g(2, 2, 1, 1 << 20) >= 1      # (1 here: no GPU in the CPU test container; > 1 on a device)
    assert g(2, 2, 1, 4096) == 1 and g(3, 2, 1, 16000) == 1 and g(2, 1, 1, 1 << 20) == 1 and g(2, 2, 0, 1 << 20) == 1
