"""GPU tests of the drop-in boundary for existing callers:
  * the reference's mangled entry points (include/fl_legacy.h; cpp/NonlinearOptimization.hpp:278-393)
    driven with HOST callbacks, called the way FL::NO::* calls them (everything by reference, every
    optional passed, -1/0 logicals, hidden string length) and the way a Fortran caller does (absent
    optionals = NULL);
  * the batched reverse-communication API (fl_rci_*) with a torch objective.
Callbacks evaluate the oracle's objective functions in the kernels' summation order, so the result must
equal the oracle's (and therefore the fused kernels') BIT FOR BIT, including the callback counts.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

F_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))
FD_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))
FFD_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int), ("d", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double))]


def _callbacks(kind, n, d=None, b=None):
    """host callbacks with the reference's signatures, evaluating the oracle objective in GPU order"""
    import FortranLibrary.NonlinearOptimization as NLO
    lib = O.lib()
    T, E = NLO.reduction_geometry(n)
    lib.flo_set_sum_mode(O.TREE, T, E)
    dp = C.POINTER(C.c_double)
    P = Problem(kind, d.ctypes.data_as(dp) if d is not None else None, b.ctypes.data_as(dp) if b is not None else None)
    lib.flo_prob_f.argtypes = [dp, dp, C.c_int, C.c_void_p]
    lib.flo_prob_fd.argtypes = [dp, dp, C.c_int, C.c_void_p]
    cnt = {"f": 0, "fd": 0, "f_fd": 0}

    def f(fx, x, dim):
        cnt["f"] += 1
        lib.flo_prob_f(fx, x, dim[0], C.byref(P))

    def fd(g, x, dim):
        cnt["fd"] += 1
        lib.flo_prob_fd(g, x, dim[0], C.byref(P))

    def f_fd(fx, g, x, dim):
        cnt["f_fd"] += 1
        lib.flo_prob_f(fx, x, dim[0], C.byref(P))
        lib.flo_prob_fd(g, x, dim[0], C.byref(P))
        return 0

    return F_CB(f), FD_CB(fd), FFD_CB(f_fd), cnt, (T, E), P


def _common(strong=True, warning=False, maxit=1000, precision=1e-15, minstep=1e-15, c1=1e-4, c2=0.9, incr=1.05):
    # cpp/README.md:15-18: logical is a 4-byte integer, the header sends -1 / 0
    vals = [C.c_int32(-1 if strong else 0), C.c_int32(-1 if warning else 0), C.c_int(maxit), C.c_double(precision),
            C.c_double(minstep), C.c_double(c1), C.c_double(c2), C.c_double(incr)]
    return vals, [C.byref(v) for v in vals]


def _fl():
    import FortranLibrary
    return FortranLibrary.FL


@pytest.mark.parametrize("with_ffd", [False, True])
def test_legacy_lbfgs_and_cg_host_callbacks_bitexact(with_ffd):
    FL = _fl()
    n = 10
    x0 = np.full(n, -1.2)
    x0[1::2] = 1.0
    f, fd, ffd, cnt, (T, E), _ = _callbacks(O.ROSENBROCK, n)
    dim = C.c_int(n)
    dp = C.POINTER(C.c_double)
    # L-BFGS, Fortran-style call: Memory present, other optionals absent (NULL) -> reference defaults
    x = x0.copy()
    mem = C.c_int(10)
    FL.__nonlinearoptimization_MOD_lbfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), C.byref(mem), ffd if with_ffd else None,
                                         None, C.byref(C.c_int32(0)), None, None, None, None, None, None)
    ref = O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, use_ffd=with_ffd, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] + cnt["f_fd"] == ref["nf"][0] and cnt["fd"] + cnt["f_fd"] == ref["ng"][0]
    # ConjugateGradient the way FL::NO::ConjugateGradient calls it (hpp:421-447): all optionals, hidden length
    for method in (b"DY", b"PR"):
        for k in cnt:
            cnt[k] = 0
        x = x0.copy()
        vals, refs = _common(c2=0.45)
        FL.__nonlinearoptimization_MOD_conjugategradient(f, fd, x.ctypes.data_as(dp), C.byref(dim), method,
                                                         ffd if with_ffd else None, *refs, C.c_int(2))
        ref = O.solve_batch(O.CG, O.ROSENBROCK, x0, opts=O.defaults(c2=0.45, method=0 if method == b"DY" else 1),
                            use_ffd=with_ffd, sum_mode=O.TREE, threads=T, ept=E)
        assert np.array_equal(x, ref["x"][0]), method
        assert cnt["f"] + cnt["f_fd"] == ref["nf"][0] and cnt["fd"] + cnt["f_fd"] == ref["ng"][0]


def test_legacy_entry_points_beyond_the_register_path_n5001():
    """The reference's routines take any dim; beyond n = 4096 the machine's vectors live in HBM (csrc/fl_big.hpp,
    rci_step_big_kernel).  Host callbacks, odd n: equal to the oracle bit for bit, callback counts included."""
    FL = _fl()
    n = 5001
    rng = np.random.default_rng(5)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, n)
    f, fd, ffd, cnt, (T, E), _ = _callbacks(O.ROSENBROCK, n)
    assert T == 1024
    dim = C.c_int(n)
    dp = C.POINTER(C.c_double)
    x = x0.copy()
    vals, refs = _common(maxit=25, precision=1e-9)
    mem = C.c_int(5)
    FL.__nonlinearoptimization_MOD_lbfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), C.byref(mem), ffd, *refs)
    ref = O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, opts=O.defaults(maxit=25, precision=1e-9, memory=5), use_ffd=True,
                        sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] + cnt["f_fd"] == ref["nf"][0] and cnt["fd"] + cnt["f_fd"] == ref["ng"][0]
    for k in cnt:
        cnt[k] = 0
    x = x0.copy()
    vals, refs = _common(maxit=25, precision=1e-9, c2=0.45)
    FL.__nonlinearoptimization_MOD_conjugategradient_basic(f, fd, x.ctypes.data_as(dp), C.byref(dim), b"DY", *refs, C.c_int(2))
    ref = O.solve_batch(O.CG, O.ROSENBROCK, x0, opts=O.defaults(maxit=25, precision=1e-9, c2=0.45), sum_mode=O.TREE,
                        threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] == ref["nf"][0] and cnt["fd"] == ref["ng"][0]
    for k in cnt:
        cnt[k] = 0
    x = x0.copy()
    vals, refs = _common(maxit=12, precision=1e-9, strong=False)
    FL.__nonlinearoptimization_MOD_steepestdescent(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, *refs)
    ref = O.solve_batch(O.SD, O.ROSENBROCK, x0, opts=O.defaults(maxit=12, precision=1e-9, strong=0), sum_mode=O.TREE,
                        threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] == ref["nf"][0] and cnt["fd"] == ref["ng"][0]


def test_legacy_test_cpp_sequence_quartic_dim10():
    """the calls of the reference's test/test.cpp:84-125 that are on this path: SteepestDescent,
    ConjugateGradient (basic and with f_fd), BFGS -- quartic, dim 10; criterion 'close to 0'"""
    FL = _fl()
    n = 10
    rng = np.random.default_rng(5)
    x0 = rng.random(n)
    f, fd, ffd, cnt, (T, E), _ = _callbacks(O.QUARTIC, n)
    dim = C.c_int(n)
    dp = C.POINTER(C.c_double)
    # SteepestDescent (both manglings)
    for sym in ("__nonlinearoptimization_MOD_steepestdescent", "nonlinearoptimization_mp_steepestdescent_"):
        x = x0.copy()
        vals, refs = _common(maxit=300)
        getattr(FL, sym)(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, *refs)
        ref = O.solve_batch(O.SD, O.QUARTIC, x0, opts=O.defaults(maxit=300), sum_mode=O.TREE, threads=T, ept=E)
        assert np.array_equal(x, ref["x"][0])
    # ConjugateGradient_basic: no clamps, no f_fd
    x = x0.copy()
    vals, refs = _common(c2=0.45)
    FL.__nonlinearoptimization_MOD_conjugategradient_basic(f, fd, x.ctypes.data_as(dp), C.byref(dim), b"DY", *refs,
                                                           C.c_int(2))
    ref = O.solve_batch(O.CG, O.QUARTIC, x0, opts=O.defaults(c2=0.45, clamp=0), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0]) and np.linalg.norm(x) < 1e-3
    # BFGS with ExactStep=0 (test.cpp passes fdd; the device path runs the quasi-Newton branch)
    x = x0.copy()
    vals, refs = _common()
    es = C.c_int(0)
    FL.__nonlinearoptimization_MOD_bfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, C.byref(es), ffd, *refs)
    o = O.defaults(exact_step=0)
    ref = O.solve_batch(O.BFGS, O.QUARTIC, x0, opts=o, use_ffd=True, bfgs_form=1, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0]) and np.linalg.norm(x) < 1e-3
    # unknown method: reference prints and stops; the library prints and returns, x untouched
    x = x0.copy()
    FL.__nonlinearoptimization_MOD_conjugategradient(f, fd, x.ctypes.data_as(dp), C.byref(dim), b"XX", None, *refs,
                                                     C.c_int(2))
    assert np.array_equal(x, x0)


def test_rci_batched_torch_objective_equals_fused_kernel():
    """ask/tell loop with the objective evaluated by torch on the GPU for the whole batch.  torch's sums differ
    from the kernel's order, so compare against the fused kernel by tolerance; status/convergence must hold."""
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n = 32, 96
    rng = np.random.default_rng(1)
    kappa = np.exp(rng.uniform(np.log(10), np.log(100), B))
    d = torch.tensor(1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :], device=dev)
    b = torch.tensor(rng.uniform(-1, 1, (B, n)), device=dev)

    def fun(x):
        return 0.5 * (d * x * x).sum(1) - (b * x).sum(1), d * x - b

    for solver in (NLO.LBFGS_, NLO.CG, NLO.BFGS_):
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        out = NLO.minimize_rci(solver, x, fun, Precision=1e-6)
        assert np.all(out["status"].cpu().numpy() == O.CONVERGED)
        xs = b / d
        assert float(((x - xs).norm(dim=1) / xs.norm(dim=1)).max()) < 1e-5
        x2 = torch.zeros(B, n, dtype=torch.float64, device=dev)
        fused = {NLO.LBFGS_: NLO.LBFGS, NLO.CG: NLO.ConjugateGradient, NLO.BFGS_: NLO.BFGS}[solver]
        extra = {"ExactStep": 0} if solver == NLO.BFGS_ else {}
        ref = fused(NLO.DIAGQUAD, x2, d, b, Precision=1e-6, **extra)
        assert torch.allclose(out["f"], ref["f"], rtol=1e-10, atol=0)


@pytest.mark.parametrize("solver_name", ["LBFGS", "CG", "SD"])
def test_rci_full_and_compact_modes_walk_the_same_path_in_fewer_rounds(solver_name):
    """fl_rci_step_flags(FL_RCI_BOTH) answers same-point requests inside the kernel, fl_rci_step_compact steps only the
    listed problems with the step's arrays in list order: same minimisers, iteration and evaluation COUNTS (the
    reference's callback pattern) as the classic one-request-per-step loop, bit for bit -- in fewer rounds, and the
    objective sees only running problems once the batch thins out (ragged iteration counts: kappa from 2 to 1e4)."""
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n = 96, 200
    rng = np.random.default_rng(3)
    kappa = np.exp(rng.uniform(np.log(2), np.log(1e4), B))
    d = torch.tensor(1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :], device=dev)
    b = torch.tensor(rng.uniform(-1, 1, (B, n)), device=dev)
    solver = {"LBFGS": NLO.LBFGS_, "CG": NLO.CG, "SD": NLO.SD}[solver_name]
    kw = dict(Precision=1e-6, MaxIteration=150 if solver_name == "SD" else 1000)
    seen = {"rows": 0, "calls": 0, "epochs": set()}
    cache = {}

    def f_of(xx, dd, bb):
        dx = dd * xx
        return 0.5 * (dx * xx).sum(1) - (bb * xx).sum(1), dx - bb

    def fun_compact(xc, req, ids, epoch):
        if cache.get("epoch") != epoch:  # gather the problems' data once per change of the list
            i = ids.long()
            cache.update(epoch=epoch, d=d[i], b=b[i])
        seen["rows"] += xc.shape[0]
        seen["calls"] += 1
        seen["epochs"].add(epoch)
        return f_of(xc, cache["d"], cache["b"])
    xa = torch.zeros(B, n, dtype=torch.float64, device=dev)
    a = NLO.minimize_rci(solver, xa, lambda xx: f_of(xx, d, b), **kw)
    xf = torch.zeros(B, n, dtype=torch.float64, device=dev)
    f = NLO.minimize_rci(solver, xf, lambda xx, rq: f_of(xx, d, b), mode="full", **kw)
    xc = torch.zeros(B, n, dtype=torch.float64, device=dev)
    c = NLO.minimize_rci(solver, xc, fun_compact, mode="compact", check_every=4, **kw)
    for o, xx in ((f, xf), (c, xc)):
        assert torch.equal(xx, xa)
        for k in ("f", "gg", "iters", "status", "nf", "ng"):
            assert torch.equal(o[k], a[k]), k
    assert f["steps"] < a["steps"] and c["steps"] <= f["steps"]  # (steps are counted up to the next look at the requests)
    assert len(seen["epochs"]) > 1 and seen["rows"] < 0.8 * B * c["steps"]  # the tail is evaluated for the running problems only


@pytest.mark.parametrize("solver_name,n,groups", [("LBFGS", 10000, 4), ("CG", 10000, 4), ("LBFGS", 20001, 8), ("SD", 9000, 2)])
def test_cooperative_large_n_solve_bitexact(solver_name, n, groups, monkeypatch):
    """n > 4096 with few problems: several workgroups share ONE problem (BigSolver's cooperative form, FL_COOP_GROUPS):
    each owns a range of the vector, all run the same scalar machine and meet in every reduction through device-scope
    counters.  Sums are taken per workgroup and added left to right -- the oracle's tree order with `groups` -- so the
    minimiser, iteration and evaluation counts equal the oracle's bit for bit; with one workgroup per problem the bits
    are those of the plain path."""
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B = 2
    rng = np.random.default_rng(n)
    kappa = np.array([30.0, 200.0])
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    b = rng.uniform(-1, 1, (B, n))
    solver = {"LBFGS": NLO.LBFGS_, "CG": NLO.CG, "SD": NLO.SD}[solver_name]
    osolver = {"LBFGS": O.LBFGS, "CG": O.CG, "SD": O.SD}[solver_name]
    kw = dict(Precision=1e-6, MaxIteration=40 if solver_name == "SD" else 300)
    T, E = NLO.reduction_geometry(n)
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_prob_eval_batch.argtypes = [C.c_int] * 4 + [dp, dp, dp] + [C.c_int] * 4 + [dp] * 4
    P = lambda a: a.ctypes.data_as(dp)
    f_h, g_h = np.zeros(B), np.zeros((B, n))

    def fun(xdev):  # the caller's objective, in the summation order of the run at hand (lib.flo_set_sum_groups)
        xh = np.ascontiguousarray(xdev.cpu().numpy())
        lib.flo_prob_eval_batch(O.DIAGQUAD, B, n, 0, P(xh), P(d), P(b), O.TREE, T, E, 1, P(f_h), P(g_h), None, None)
        return torch.tensor(f_h, device=dev), torch.tensor(g_h, device=dev)
    oo = O.defaults(precision=1e-6, maxit=kw["MaxIteration"], c2=0.45 if solver_name == "CG" else 0.9)
    res = {}
    for want in (groups, 1):
        monkeypatch.setenv("FL_COOP_GROUPS", str(want))
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        lib.flo_set_sum_groups(1)
        # (the handle reports how many workgroups really share a problem: no empty ones)
        probe = NLO.minimize_rci(solver, x, fun, max_steps=0, **kw)
        G = probe["cooperative_groups"]
        assert (G > 1) == (want > 1)
        lib.flo_set_sum_groups(G)
        try:
            x = torch.zeros(B, n, dtype=torch.float64, device=dev)
            out = NLO.minimize_rci(solver, x, fun, check_every=1, **kw)
            o = O.solve_batch(osolver, O.DIAGQUAD, np.zeros((B, n)), d=d, b=b, opts=oo, sum_mode=O.TREE, threads=T, ept=E, nthreads=1)
        finally:
            lib.flo_set_sum_groups(1)
        assert np.array_equal(x.cpu().numpy().view(np.uint64), o["x"].view(np.uint64)), want
        assert np.array_equal(out["iters"].cpu().numpy(), o["iters"]) and np.array_equal(out["nf"].cpu().numpy(), o["nf"])
        assert np.array_equal(out["ng"].cpu().numpy(), o["ng"]) and int(o["iters"].min()) > 3
        res[want] = x.cpu().numpy()
    assert np.abs(res[groups] - res[1]).max() < 1e-6  # same minimiser, different (each reproducible) summation orders


def test_fortran_use_fortranlibrary_smoke():
    """`use FortranLibrary` from Fortran (amdflang): the shim module forwards to libFL.so, the solver runs on the
    GPU and calls the Fortran callbacks on the host.  Mirrors test/test.f90:330-413: residuals close to 0."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran-library_amd", "fortran", "test_nlopt")
    if not os.path.exists(exe):
        pytest.skip("Fortran smoke test not built (make -C fortran-library_amd/fortran)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Mission complete" in out.stdout
    vals = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) == 2 and parts[0] not in ("Steepest", "Conjugate", "Mission"):
            try:
                vals[parts[0]] = float(parts[1])
            except ValueError:
                pass
    assert len(vals) == 19, out.stdout
    for k, v in vals.items():
        assert v < (0.2 if k == "SD" else 1e-7 if k.startswith("AugLag") else 1.0 if k.endswith("f/f0") else 1e-3), (k, v)  # steepest descent on a quartic crawls; the rest reach ~1e-5


def test_fortran_batched_entries_rci_loop_and_compiled_objective():
    """tests/fortran/test_batched.f90: a Fortran program with a BATCH of problems -- AugmentedLagrangian_batched on device
    memory (BASELINE config 5's shape) equal to fl_multi_solve bit for bit, a reverse-communication loop whose objective is
    evaluated in Fortran on the host, and the objective handed over as HIP source text (fl_user_compile) equal to the
    built-in one bit for bit.  The program checks itself and prints 'Mission complete'."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fortran-library_amd", "fortran", "test_batched")
    if not os.path.exists(exe):
        pytest.skip("Fortran batched test not built (make -C fortran-library_amd/fortran)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Mission complete" in out.stdout and "FAILED" not in out.stdout, out.stdout
    assert out.stdout.count("same bits") == 2 and " T" in out.stdout, out.stdout


C_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))


@pytest.mark.parametrize("solver,osolver", [(b"LBFGS", O.LBFGS), (b"ConjugateGradient", O.CG)])
def test_legacy_augmented_lagrangian_unit_sphere_like_test_cpp(solver, osolver):
    """test/test.cpp:112-125 / test/test.f90:452-478: quartic, dim 10, constraint x.x = 1, AugmentedLagrangian with
    host callbacks f, fd, c, cd through the mangled symbol; must equal the oracle bit for bit; |x| - 1 close to 0."""
    FL = _fl()
    n, m = 10, 1
    rng = np.random.default_rng(3)
    x0 = rng.random(n)
    f, fd, ffd, cnt, (T, E), P = _callbacks(O.QUARTIC, n)
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_prob_c.argtypes = [dp, dp, C.c_int, C.c_int, C.c_void_p]
    lib.flo_prob_cd.argtypes = [dp, dp, C.c_int, C.c_int, C.c_void_p]
    c = C_CB(lambda cx, x, M, N: lib.flo_prob_c(cx, x, M[0], N[0], None))
    cd = C_CB(lambda cdx, x, M, N: lib.flo_prob_cd(cdx, x, M[0], N[0], None))
    x = x0.copy()
    N_, M_ = C.c_int(n), C.c_int(m)
    lam0 = np.zeros(m)
    miu0 = C.c_double(1.0)
    es, mem = C.c_int(0), C.c_int(10)
    vals, refs = _common(precision=1e-8, c2=0.45 if osolver == O.CG else 0.9)
    FL.__nonlinearoptimization_MOD_augmentedlagrangian(f, fd, c, cd, x.ctypes.data_as(dp), C.byref(N_), C.byref(M_), solver,
                                                       lam0.ctypes.data_as(dp), C.byref(miu0), None, None, C.byref(es),
                                                       C.byref(mem), b"DY", None, *refs, C.c_int(len(solver)), C.c_int(2))
    assert abs(np.linalg.norm(x) - 1.0) < 1e-7
    ref = O.auglag_batch(osolver, O.QUARTIC, x0, m, opts=O.defaults(precision=1e-8, c2=0.45 if osolver == O.CG else 0.9),
                         sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] + cnt["f_fd"] == ref["nf"][0] and cnt["fd"] + cnt["f_fd"] == ref["ng"][0]


def test_cpp_caller_links_and_runs_against_libFL():
    """A C++ program that declares the mangled symbols like the reference header (C++ references, -1/0 logicals,
    hidden string lengths) links with plain g++ against libFL.so and runs the reference test's optimiser calls."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "legacy_cpp_caller.cpp")
    exe = os.path.join(root, "tests", "_build", "legacy_cpp_caller")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(root, "fortran-library_amd", "lib")
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", src, "-o", exe, "-L" + lib, "-lFL", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert out.returncode == 0, out.stderr
    assert "Mission complete" in out.stdout
    vals = dict(line.split() for line in out.stdout.splitlines() if len(line.split()) == 2 and line[0] != "M")
    assert len(vals) == 5
    for k, v in vals.items():
        assert float(v) < (0.2 if k == "SD" else 1e-3 if not k.startswith("AugLag") else 1e-7), (k, v)


def test_cpp_program_written_against_the_FL_NO_header():
    """cpp/FortranLibrary.hpp (namespace FL::NO, the reference header's names, argument order and defaults; plus
    the new LBFGS): a user program compiles with plain g++, links -lFL and passes the reference test's checks."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "header_cpp_caller.cpp")
    hdr = os.path.join(root, "fortran-library_amd", "cpp", "NonlinearOptimization.hpp")
    exe = os.path.join(root, "tests", "_build", "header_cpp_caller")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(root, "fortran-library_amd", "lib")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", exe, "-L" + lib, "-lFL", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", "")))
    assert out.returncode == 0, out.stderr
    assert "Mission complete" in out.stdout
    vals = dict(line.split() for line in out.stdout.splitlines() if len(line.split()) == 2 and line[0] != "M")
    assert len(vals) == 8, out.stdout
    for k, v in vals.items():
        assert float(v) < (0.2 if k == "SD" else 1e-7 if k.startswith("AugLag") else 1e-3), (k, v)


FDD_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int))


@pytest.mark.parametrize("kind,n", [(O.QUARTIC, 10), (O.ROSENBROCK, 10), (O.ROSENBROCK, 40)])
def test_legacy_newton_and_bfgs_with_host_hessian_callback(kind, n):
    """NewtonRaphson (hpp:344-358) and BFGS with ExactStep > 0: the caller's fdd runs on the host, the Hessian is
    shipped to the GPU (FL_REQ_H), Cholesky solve / inverse there; equal to the oracle bit for bit."""
    FL = _fl()
    rng = np.random.default_rng(n)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, n) if kind == O.ROSENBROCK else 0.1 * np.arange(1, n + 1)
    f, fd, ffd, cnt, (T, E), P = _callbacks(kind, n)
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_prob_fdd.argtypes = [dp, dp, C.c_int, C.c_void_p]
    nh = {"n": 0}

    def fdd_py(H, x, dim):
        nh["n"] += 1
        lib.flo_prob_fdd(H, x, dim[0], C.byref(P))
        return 0
    fdd = FDD_CB(fdd_py)
    dim = C.c_int(n)
    x = x0.copy()
    vals, refs = _common()
    FL.__nonlinearoptimization_MOD_newtonraphson(f, fd, x.ctypes.data_as(dp), C.byref(dim), fdd, ffd, *refs)
    ref = O.solve_batch(4, kind, x0, use_ffd=True, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert nh["n"] == ref["iters"][0] + 1 or ref["status"][0] != O.MAXIT
    x = x0.copy()
    es = C.c_int(5)
    FL.__nonlinearoptimization_MOD_bfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), fdd, C.byref(es), None, *refs)
    ref = O.solve_batch(O.BFGS, kind, x0, opts=O.defaults(exact_step=5), bfgs_form=1, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    # without fdd the reference differentiates f' numerically (MKL djacobi, NO.f90:676, 1067); here central
    # differences of the caller's fd on the host: 2n gradient calls per Hessian, same minimiser
    x_fdd = x.copy()
    x = x0.copy()
    before = cnt["fd"]
    FL.__nonlinearoptimization_MOD_bfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, C.byref(es), None, *refs)
    assert cnt["fd"] - before >= 2 * n
    assert np.linalg.norm(x - x_fdd) < 1e-6 * max(1.0, np.linalg.norm(x_fdd)) or kind == O.QUARTIC
    x = x0.copy()
    FL.__nonlinearoptimization_MOD_newtonraphson(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, None, *refs)
    if kind == O.ROSENBROCK:
        assert np.max(np.abs(x - 1.0)) < 1e-7
    else:
        assert np.linalg.norm(x) < 1e-3


def test_legacy_bfgs_default_numerical_hessian_rosenbrock_n10():
    """BASELINE.md section 2 probe (= BASELINE config 1 with its defaults): BFGS, ExactStep=20, no fdd -- numerical
    Hessian by MKL's djacobi + Cholesky inverse at the start and every 20 iterations -- on Rosenbrock n=10 from the
    standard start ends at f=0, x=1 after 844 f / 845 fd callbacks.  The library's central differences use djacobi's own
    step rule (fl_djacobi, pinned to the real MKL routine by tests/golden/mkl_djacobi.npz), so the legacy symbol with
    host callbacks reproduces the oracle bit for bit -- minimiser AND callback counts -- and the reference's counts."""
    FL = _fl()
    n = 10
    x0 = np.full(n, -1.2)
    x0[1::2] = 1.0
    f, fd, ffd, cnt, (T, E), P = _callbacks(O.ROSENBROCK, n)
    dp = C.POINTER(C.c_double)
    dim = C.c_int(n)
    x = x0.copy()
    vals, refs = _common()
    FL.__nonlinearoptimization_MOD_bfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, None, None, *refs)
    assert np.max(np.abs(x - 1.0)) < 1e-8
    fx = C.c_double(1.0)
    O.lib().flo_prob_f(C.byref(fx), x.ctypes.data_as(dp), n, C.byref(P))
    assert fx.value < 1e-20
    ref = O.solve_batch(O.BFGS, O.ROSENBROCK, x0, opts=O.defaults(), bfgs_form=1 + 4096, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert (cnt["f"], cnt["fd"]) == (ref["nf"][0], ref["ng"][0])
    seq = O.solve_batch(O.BFGS, O.ROSENBROCK, x0, opts=O.defaults(), bfgs_form=4096)  # the reference's own order: 844 / 845
    assert (seq["nf"][0] + 1, seq["ng"][0]) == (844, 845)
    assert abs(cnt["f"] - seq["nf"][0]) <= 40 and abs(cnt["fd"] - seq["ng"][0]) <= 40  # (summation order moves the path a little)


@pytest.mark.parametrize("n", [10, 300, 1024, 5001])
def test_public_line_searchers_bitexact(n):
    """Wolfe, Wolfe_fdwithf, StrongWolfe, StrongWolfe_fdwithf are public procedures of the reference module
    (NO.f90:1286, 1373, 1462, 1582): one search along a given p through the mangled symbols, host callbacks, trial
    points and phi'(a) on the GPU -- step, objective, point, gradient and callback counts equal the oracle's
    structured restatement bit for bit (kernel summation order), from far too short to far too long first steps."""
    import test_host_logic as H
    FL = _fl()
    drv = H._driver()
    dp = C.POINTER(C.c_double)
    drv.ls_oracle_run.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, dp, dp, dp, dp,
                                  C.c_double, dp, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(n)
    kind = O.ROSENBROCK
    f, fd, ffd, cnt, (T, E), P = _callbacks(kind, n)
    dim = C.c_int(n)
    names = [("__nonlinearoptimization_MOD_wolfe", 0, 0), ("__nonlinearoptimization_MOD_wolfe_fdwithf", 0, 1),
             ("__nonlinearoptimization_MOD_strongwolfe", 1, 0), ("nonlinearoptimization_mp_strongwolfe_fdwithf_", 1, 1)]
    for trial in range(6):
        x0 = 1.0 + 0.3 * rng.uniform(-1, 1, n)
        g0 = np.zeros(n)
        fx0 = C.c_double(0.0)
        O.lib().flo_prob_f(C.byref(fx0), x0.ctypes.data_as(dp), n, C.byref(P))
        O.lib().flo_prob_fd(g0.ctypes.data_as(dp), x0.ctypes.data_as(dp), n, C.byref(P))
        p = -g0 * rng.uniform(0.5, 1.5, n)
        phid0 = float(g0 @ p)
        a0 = float(10.0 ** rng.uniform(-5, 0))
        c1, c2, incr = 1e-4, (0.9 if trial % 2 else 0.45), float(rng.choice([1.05, 1.5, 2.0]))
        for name, strong, with_ffd in names:
            for k in cnt:
                cnt[k] = 0
            x, fdx = x0.copy(), np.zeros(n)
            a, fx = C.c_double(a0), C.c_double(fx0.value)
            args = [C.byref(C.c_double(c1)), C.byref(C.c_double(c2)), f, fd]
            if with_ffd:
                args.append(ffd)
            args += [x.ctypes.data_as(dp), C.byref(a), p.ctypes.data_as(dp), C.byref(fx), C.byref(C.c_double(phid0)),
                     fdx.ctypes.data_as(dp), C.byref(dim), C.byref(C.c_double(incr))]
            getattr(FL, name)(*args)
            xo, go = x0.copy(), np.zeros(n)
            ao, fo, nf, ng = C.c_double(a0), C.c_double(fx0.value), C.c_int(0), C.c_int(0)
            drv.ls_oracle_run(strong, 1 if (strong and with_ffd) else 0, c1, c2, incr, kind, n, xo.ctypes.data_as(dp),
                              p.ctypes.data_as(dp), C.byref(ao), C.byref(fo), phid0, None, None, go.ctypes.data_as(dp),
                              C.byref(nf), C.byref(ng))
            assert a.value == ao.value and fx.value == fo.value, (name, trial)
            assert np.array_equal(x, xo) and np.array_equal(fdx, go), (name, trial)
            assert cnt["f"] + cnt["f_fd"] == nf.value and cnt["fd"] + cnt["f_fd"] == ng.value, (name, trial)


@pytest.mark.parametrize("solver,with_hessians,exact", [(b"NewtonRaphson", True, 0), (b"NewtonRaphson", False, 0),
                                                        (b"BFGS", True, 5), (b"BFGS", False, 20), (b"BFGS", False, 0)])
def test_legacy_augmented_lagrangian_dense_inner_solvers(solver, with_hessians, exact):
    """AugmentedLagrangian with UnconstrainedSolver = 'NewtonRaphson' / 'BFGS' (NO.f90:2074-2149): with fdd and cdd
    the inner solver receives Ldd (NO.f90:2229-2240, composed on the host next to the callbacks), without them the
    Hessian of the Lagrangian comes from central differences of Ld (the reference: MKL djacobi).  Quartic, dim 10,
    unit sphere: |x| - 1 close to 0 and x a constrained stationary point (f' parallel to x)."""
    FL = _fl()
    n, m = 10, 1
    rng = np.random.default_rng(3)
    x0 = rng.random(n)
    f, fd, ffd, cnt, (T, E), P = _callbacks(O.QUARTIC, n)
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_prob_c.argtypes = [dp, dp, C.c_int, C.c_int, C.c_void_p]
    lib.flo_prob_cd.argtypes = [dp, dp, C.c_int, C.c_int, C.c_void_p]
    lib.flo_prob_fdd.argtypes = [dp, dp, C.c_int, C.c_void_p]
    c = C_CB(lambda cx, x, M, N: lib.flo_prob_c(cx, x, M[0], N[0], None))
    cd = C_CB(lambda cdx, x, M, N: lib.flo_prob_cd(cdx, x, M[0], N[0], None))

    def fdd_py(H, x, dim):
        lib.flo_prob_fdd(H, x, dim[0], C.byref(P))
        return 0

    def cdd_py(cddx, x, M, N):  # c = x.x - 1: c'' = 2 I
        nn = N[0]
        for i in range(nn * nn):
            cddx[i] = 0.0
        for i in range(nn):
            cddx[i * nn + i] = 2.0
        return 0
    CDD_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))
    fdd, cdd = FDD_CB(fdd_py), CDD_CB(cdd_py)
    x = x0.copy()
    N_, M_ = C.c_int(n), C.c_int(m)
    lam0 = np.zeros(m)
    miu0 = C.c_double(1.0)
    es, mem = C.c_int(exact), C.c_int(10)
    vals, refs = _common(precision=1e-8, maxit=60)
    FL.__nonlinearoptimization_MOD_augmentedlagrangian(f, fd, c, cd, x.ctypes.data_as(dp), C.byref(N_), C.byref(M_), solver,
                                                       lam0.ctypes.data_as(dp), C.byref(miu0), fdd if with_hessians else None,
                                                       cdd if with_hessians else None, C.byref(es), C.byref(mem), b"DY", None,
                                                       *refs, C.c_int(len(solver)), C.c_int(2))
    assert abs(np.linalg.norm(x) - 1.0) < 1e-7
    g = 4.0 * x ** 3
    lam = (g @ x) / (2.0 * x @ x)  # f' = lambda c' at a constrained stationary point
    assert np.linalg.norm(g - lam * 2.0 * x) < 1e-5


@pytest.mark.parametrize("n", [1, 2, 11, 64, 130, 300, 700])
def test_dsysv_batched_bitexact_and_against_numpy(n):
    """fl_dsysv_batched (My_dsysv, LA.f90:695-703): symmetric indefinite systems given by their lower triangle --
    KKT-shaped ([H, C; C^T, 0]) and random ones; bit for bit the oracle's elimination, numpy's solution to rounding,
    info > 0 and an untouched right-hand side for a singular matrix."""
    import FortranLibrary.NonlinearOptimization as NLO
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_dsysv.argtypes = [dp, dp, C.c_int]
    lib.flo_dsysv.restype = C.c_int
    rng = np.random.default_rng(n)
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    B = 4
    A = np.zeros((B, n, n))
    for k in range(B):
        if k % 2 == 0 and n >= 4:  # KKT shape: SPD block, constraint Jacobian, zero block
            m = max(1, n // 5)
            h = rng.standard_normal((n - m, n - m))
            A[k, :n - m, :n - m] = h @ h.T / (n - m) + np.eye(n - m)
            cj = rng.standard_normal((n - m, m))
            A[k, :n - m, n - m:] = -cj
            A[k, n - m:, :n - m] = -cj.T
        else:
            s = rng.standard_normal((n, n))
            A[k] = s + s.T
    if n >= 2:
        A[3] = 0.0  # singular: first column has no non-zero entry
        A[3, 1:, 1:] = np.eye(n - 1)
    rhs = rng.standard_normal((B, n))
    Ap = np.zeros((B, n, ld))
    for k in range(B):
        Ap[k, :, :n] = np.tril(A[k]).T  # column-major [col][row]: lower triangle only, upper left as garbage-free zeros
    dev = torch.device("cuda:0")
    Ad, bd = torch.tensor(Ap, device=dev), torch.tensor(rhs, device=dev)
    info = NLO.dsysv(Ad, bd).cpu().numpy()
    sol = bd.cpu().numpy()
    for k in range(B):
        a = np.ascontiguousarray(np.tril(A[k]).T)  # column-major n x n with the lower triangle
        bb = rhs[k].copy()
        oi = lib.flo_dsysv(a.ctypes.data_as(dp), bb.ctypes.data_as(dp), n)
        assert info[k] == oi
        assert np.array_equal(sol[k], bb)
        if oi == 0:
            assert np.allclose(sol[k], np.linalg.solve(A[k], rhs[k]), rtol=1e-7, atol=1e-9)
        else:
            assert np.array_equal(sol[k], rhs[k])
    if n >= 2:
        assert info[3] == 1


def test_legacy_lagrangian_multiplier_unit_sphere():
    """LagrangianMultiplier (NO.f90:1950-1993) through its mangled symbol: quartic, dim 10, constraint x.x = 1, from a
    start near the constrained minimiser (the method is a Newton iteration on the KKT system: local).  Callbacks on
    the host, the symmetric indefinite solve on the GPU; equal to the oracle bit for bit; |x| = 1, f' = lambda c'."""
    FL = _fl()
    n, m = 10, 1
    f, fd, ffd, cnt, (T, E), P = _callbacks(O.QUARTIC, n)
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    for name in ("flo_prob_c", "flo_prob_cd", "flo_prob_cdd"):
        getattr(lib, name).argtypes = [dp, dp, C.c_int, C.c_int, C.c_void_p]
    lib.flo_prob_fdd.argtypes = [dp, dp, C.c_int, C.c_void_p]
    CDD_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))
    c = C_CB(lambda cx, x, M, N: lib.flo_prob_c(cx, x, M[0], N[0], None))
    cd = C_CB(lambda cdx, x, M, N: lib.flo_prob_cd(cdx, x, M[0], N[0], None))
    cdd = CDD_CB(lambda cddx, x, M, N: lib.flo_prob_cdd(cddx, x, M[0], N[0], None))
    fdd = FDD_CB(lambda H, x, dim: lib.flo_prob_fdd(H, x, dim[0], C.byref(P)))
    rng = np.random.default_rng(1)
    x0 = np.full(n, 1.0 / np.sqrt(n)) * (1.0 + 0.05 * rng.uniform(-1, 1, n))
    lam0 = np.array([0.2])
    x, lam = x0.copy(), lam0.copy()
    N_, M_ = C.c_int(n), C.c_int(m)
    FL.__nonlinearoptimization_MOD_lagrangianmultiplier(fd, fdd, c, cd, cdd, x.ctypes.data_as(dp), lam.ctypes.data_as(dp),
                                                        C.byref(N_), C.byref(M_), C.byref(C.c_int32(0)), C.byref(C.c_int(50)),
                                                        C.byref(C.c_double(1e-12)))
    assert abs(np.linalg.norm(x) - 1.0) < 1e-10
    assert np.linalg.norm(4.0 * x ** 3 - lam[0] * 2.0 * x) < 1e-10
    lib.flo_lagrangian_multiplier.argtypes = [C.c_void_p] * 5 + [dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    lib.flo_lagrangian_multiplier.restype = C.c_int
    xo, lo = x0.copy(), lam0.copy()
    addr = lambda fn: C.cast(fn, C.c_void_p)
    its = lib.flo_lagrangian_multiplier(addr(lib.flo_prob_fd), addr(lib.flo_prob_fdd), addr(lib.flo_prob_c), addr(lib.flo_prob_cd),
                                        addr(lib.flo_prob_cdd), xo.ctypes.data_as(dp), lo.ctypes.data_as(dp), n, m, 50, 1e-12,
                                        C.cast(C.byref(P), C.c_void_p))
    assert 1 <= its < 50
    assert np.array_equal(x, xo) and np.array_equal(lam, lo)


def test_linear_algebra_symbols_of_the_cpp_header():
    """__linearalgebra_MOD_my_dgemm(_t) and __linearalgebra_MOD_my_dsyev (cpp/FortranLibrary.hpp:48-63): host arrays
    in and out like the reference, rocBLAS / rocSOLVER underneath; numpy to rounding."""
    FL = _fl()
    rng = np.random.default_rng(0)
    dp = C.POINTER(C.c_double)
    M, K, N = 37, 53, 29
    A = np.asfortranarray(rng.standard_normal((M, K)))
    At = np.asfortranarray(A.T.copy())  # K x M
    B = np.asfortranarray(rng.standard_normal((K, N)))
    Cm = np.asfortranarray(np.zeros((M, N)))
    iM, iK, iN = C.c_int(M), C.c_int(K), C.c_int(N)
    FL.__linearalgebra_MOD_my_dgemm(A.ctypes.data_as(dp), B.ctypes.data_as(dp), Cm.ctypes.data_as(dp), C.byref(iM), C.byref(iK), C.byref(iN))
    assert np.allclose(Cm, A @ B, rtol=1e-13, atol=1e-13)
    Cm[:] = 0.0
    FL.__linearalgebra_MOD_my_dgemm_t(At.ctypes.data_as(dp), B.ctypes.data_as(dp), Cm.ctypes.data_as(dp), C.byref(iM), C.byref(iK), C.byref(iN))
    assert np.allclose(Cm, A @ B, rtol=1e-13, atol=1e-13)
    n = 41
    S = rng.standard_normal((n, n))
    S = np.asfortranarray(S + S.T)
    S0 = S.copy()
    w = np.zeros(n)
    i_n = C.c_int(n)
    FL.__linearalgebra_MOD_my_dsyev(b"V", S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(i_n), C.c_int(1))
    assert np.allclose(w, np.linalg.eigvalsh(S0), rtol=1e-12, atol=1e-12) and np.all(np.diff(w) >= 0)
    assert np.allclose(S0 @ S, S * w[None, :], rtol=1e-10, atol=1e-10)      # A v_k = w_k v_k
    assert np.allclose(S.T @ S, np.eye(n), atol=1e-12)                       # normalised eigenvectors


RES_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))
JAC_CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))


@pytest.mark.parametrize("analytic", [True, False])
def test_trust_region_least_squares_chained_rosenbrock(analytic):
    """TrustRegion / TrustRegion_basic (NO.f90:1728, 2348; hpp:358-366): f'(x) = 0 for the chained Rosenbrock residuals
    r = (10 (x_{i+1} - x_i^2), 1 - x_i), M = 2 (N - 1) >= N, with the analytical Jacobian and with central differences;
    the solution is x = 1 with zero residual.  Also with bounds that cut the solution off: the projected optimum."""
    FL = _fl()
    n = 12
    m = 2 * (n - 1)

    def res(r, x, M, N):
        for i in range(n - 1):
            r[2 * i] = 10.0 * (x[i + 1] - x[i] * x[i])
            r[2 * i + 1] = 1.0 - x[i]

    def jac(J, x, M, N):  # column-major M x N
        for k in range(m * n):
            J[k] = 0.0
        for i in range(n - 1):
            J[i * m + 2 * i] = -20.0 * x[i]
            J[(i + 1) * m + 2 * i] = 10.0
            J[i * m + 2 * i + 1] = -1.0
        return 0
    r_cb, j_cb = RES_CB(res), JAC_CB(jac)
    dp = C.POINTER(C.c_double)
    x = np.full(n, -1.2)
    x[1::2] = 1.0
    M_, N_ = C.c_int(m), C.c_int(n)
    w, mi, ms = C.c_int32(0), C.c_int(200), C.c_int(50)
    pr, mn = C.c_double(1e-10), C.c_double(1e-15)
    if analytic:
        FL.__nonlinearoptimization_MOD_trustregion_basic(r_cb, j_cb, x.ctypes.data_as(dp), C.byref(M_), C.byref(N_), C.byref(w),
                                                         C.byref(mi), C.byref(ms), C.byref(pr), C.byref(mn))
    else:  # Fortran-style general routine: Jacobian absent (numerical), the other optionals absent too
        FL.__nonlinearoptimization_MOD_trustregion(r_cb, x.ctypes.data_as(dp), C.byref(M_), C.byref(N_), None, None, None,
                                                   C.byref(w), None, None, C.byref(pr), None)
    assert np.max(np.abs(x - 1.0)) < 1e-8
    lo, up = np.full(n, -2.0), np.full(n, 0.5)
    x = np.full(n, -1.2)
    x[1::2] = 0.3
    FL.__nonlinearoptimization_MOD_trustregion(r_cb, x.ctypes.data_as(dp), C.byref(M_), C.byref(N_), j_cb if analytic else None,
                                               lo.ctypes.data_as(dp), up.ctypes.data_as(dp), C.byref(w), C.byref(mi),
                                               C.byref(ms), C.byref(pr), C.byref(mn))
    assert np.all(x <= 0.5 + 1e-15) and np.all(x >= -2.0)
    rr = (C.c_double * m)()
    res(rr, x, None, None)
    x0 = np.full(n, -1.2)
    x0[1::2] = 0.3
    r0 = (C.c_double * m)()
    res(r0, x0, None, None)
    assert sum(v * v for v in rr) < 0.5 * sum(v * v for v in r0)


def test_legacy_bfgs_beyond_the_register_path_n4500():
    """__nonlinearoptimization_MOD_bfgs with dim > 4096 and ExactStep = 0: host callbacks, inverse Hessian in HBM,
    deferred rank-2 updates through the reverse-communication kernel (pending updates parked between launches)."""
    FL = _fl()
    n = 4500
    rng = np.random.default_rng(45)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, n)
    f, fd, ffd, cnt, (T, E), _ = _callbacks(O.ROSENBROCK, n)
    assert T == 1024
    dim = C.c_int(n)
    dp = C.POINTER(C.c_double)
    x = x0.copy()
    vals, refs = _common(maxit=11, precision=1e-9)
    es = C.c_int(0)
    FL.__nonlinearoptimization_MOD_bfgs(f, fd, x.ctypes.data_as(dp), C.byref(dim), None, C.byref(es), ffd, *refs)
    ref = O.solve_batch(O.BFGS, O.ROSENBROCK, x0, opts=O.defaults(maxit=11, precision=1e-9, exact_step=0), use_ffd=True,
                        bfgs_form=108, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x, ref["x"][0])
    assert cnt["f"] + cnt["f_fd"] == ref["nf"][0] and cnt["fd"] + cnt["f_fd"] == ref["ng"][0]


def test_dsysv_and_two_loop_entry_points_refuse_dimensions_beyond_the_register_path():
    """n > 4096: fl_reduction_geometry answers with the vectors-in-HBM layout (1024 threads), which the indefinite
    solver and the stand-alone recursion do not have: FL_ERR_UNSUPPORTED_SIZE (-2), nothing launched (round 1 fell
    through to the 8 x 8 instantiation and returned FL_OK with garbage)."""
    FL = _fl()
    dev = torch.device("cuda:0")
    n = 5000
    A = torch.zeros(1, 8, dtype=torch.float64, device=dev)  # never dereferenced
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert FL.fl_dsysv_batched(1, n, p(A), p(A), p(info), None) == -2  # (dposv / dpotri: blocked path, any n)
    assert FL.fl_lbfgs_two_loop_batched(1, n, 10, 9, p(A), p(A), p(A), p(A), None) == -2
