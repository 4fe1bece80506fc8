"""GPU parity of the run-time compiled objectives (fl_user_compile / fl_user_solve, csrc/fl_user_rtc.hip): an objective a
caller hands over as HIP source text runs inside the fused kernel -- and when it restates a built-in objective, every output
equals the built-in kernel's bit for bit (same geometry, same summation order), for the element-wise diagonal quadratic and
for the NEIGHBOUR-COUPLED chained Rosenbrock (the LDS_DOUBLES / barrier contract of the functor interface), across
geometries and solvers; and equals the oracle.  At the headline size the compiled form keeps >= 0.9 of the built-in kernel's
iterations per second.  Reference interface: callbacks f, fd (NO.f90:33-38)."""
import numpy as np
import pytest

import oracle_lib as O
import user_sources as US

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _quads(B, n, seed):
    rng = np.random.default_rng(seed)
    kappa = np.exp(rng.uniform(np.log(10), np.log(300), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / max(n - 1, 1))[None, :]
    return d, rng.uniform(-1, 1, (B, n))


def _same(a, b, keys=("f", "gg", "iters", "status", "nf", "ng")):
    for k in keys:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("solver,n,kw", [("LBFGS", 1024, {}), ("LBFGS", 256, {"Memory": 4}), ("LBFGS", 2048, {}), ("CG", 1024, {}),
                                         ("CG", 1000, {"Method": "PR"}), ("SD", 300, {"MaxIteration": 150}), ("BFGS", 256, {}),
                                         ("LBFGS", 37, {"Strong": False})])
def test_quadratic_given_as_source_equals_the_builtin_bit_for_bit(solver, n, kw):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B = 12
    d, b = _quads(B, n, n)
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    code = {"SD": NLO.SD, "CG": NLO.CG, "LBFGS": NLO.LBFGS_, "BFGS": NLO.BFGS_}[solver]
    obj = NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", n, solver=code, tune_like=NLO.DIAGQUAD)
    assert obj.geometry == NLO.reduction_geometry(n, code)
    half = torch.tensor([0.5], dtype=torch.float64, device=dev)
    kw = dict({"Precision": 1e-8, "MaxIteration": 400}, **kw)
    xu = torch.zeros(B, n, dtype=torch.float64, device=dev)
    ou = obj.solve(xu, dd, bb, half, **kw)
    xb = torch.zeros_like(xu)
    fn = {"SD": NLO.SteepestDescent, "CG": NLO.ConjugateGradient, "LBFGS": NLO.LBFGS, "BFGS": NLO.BFGS}[solver]
    ob = fn(NLO.DIAGQUAD, xb, dd, bb, **(dict(kw, ExactStep=0) if solver == "BFGS" else kw))
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    _same(ou, ob)
    assert int(ou["iters"].min()) > 3
    # the parameter block reaches the functor: half = 1.5 shifts every term of the first sum by 1 -> f by a constant
    xs = torch.zeros_like(xu)
    sh = obj.solve(xs, dd, bb, torch.tensor([1.5], dtype=torch.float64, device=dev), **kw)
    torch.cuda.synchronize()
    T, E = obj.geometry
    assert bool(torch.isfinite(sh["f"]).all()) and bool(((sh["f"] - ou["f"]) > 0.25 * T * E).all())  # (0.5 T E where both converged)


@pytest.mark.parametrize("solver,n", [("LBFGS", 256), ("LBFGS", 1000), ("LBFGS", 4096), ("CG", 100), ("CG", 700), ("BFGS", 130)])
def test_neighbour_coupled_rosenbrock_given_as_source_equals_the_builtin_and_the_oracle(solver, n):
    """x staged through the functor's LDS scratch with two barriers per evaluation, across 1 .. 8 waves"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, (6, n))
    code = {"CG": NLO.CG, "LBFGS": NLO.LBFGS_, "BFGS": NLO.BFGS_}[solver]
    obj = NLO.compile_objective(US.ROSENBROCK, "MyRosenbrock", n, solver=code)
    kw = {"Precision": 1e-9, "MaxIteration": 120}
    xu = torch.tensor(x0, device=dev)
    ou = obj.solve(xu, **kw)
    xb = torch.tensor(x0, device=dev)
    fn = {"CG": NLO.ConjugateGradient, "LBFGS": NLO.LBFGS, "BFGS": NLO.BFGS}[solver]
    ob = fn(NLO.ROSENBROCK, xb, **(dict(kw, ExactStep=0) if solver == "BFGS" else kw))
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    _same(ou, ob)
    if solver != "BFGS":
        T, E = obj.geometry
        osolver = O.CG if solver == "CG" else O.LBFGS
        o = O.solve_batch(osolver, O.ROSENBROCK, x0, opts=O.defaults(precision=1e-9, maxit=120, c2=0.45 if solver == "CG" else 0.9),
                          sum_mode=O.TREE, threads=T, ept=E)
        assert np.array_equal(xu.cpu().numpy().view(np.uint64), o["x"].view(np.uint64))
        assert np.array_equal(ou["nf"].cpu().numpy(), o["nf"]) and np.array_equal(ou["iters"].cpu().numpy(), o["iters"])


def test_compiled_objective_keeps_the_fused_kernels_speed_on_the_headline_family():
    """L-BFGS m = 10, n = 1024, the benched quadratics: the objective given as a source string against the built-in kernel on
    16 384 problems -- same bits, >= 0.9 of its iterations per second (the reverse-communication form: 1/22)"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B, n = 16384, 1024
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty_like(d)
    NLO.synth_diag_spectrum(20261003, d, 10.0, 1000.0)
    NLO.synth_uniform(20261003, b, -1.0, 1.0)
    obj = NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", n, solver=NLO.LBFGS_, tune_like=NLO.DIAGQUAD)
    ws = NLO.workspace(B, n, 10, dev)
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)

    def timed(fn):
        best = None
        for _ in range(3):
            x.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        return out, best, x.clone()
    ob, mb, xb = timed(lambda: NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000))
    ou, mu, xu = timed(lambda: obj.solve(x, d, b, None, workspace_=ws, Precision=1e-6, MaxIteration=3000))
    assert torch.equal(xu, xb)
    _same(ou, ob)
    print(f"built-in {mb:.2f} ms, compiled from source {mu:.2f} ms")
    assert mb / mu >= 0.9, (mb, mu)


def test_an_objective_that_returns_nan_ends_the_problem_instead_of_hanging_the_kernel():
    """FL_STATUS_NOT_FINITE (include/fl_nlopt.h): the reference's line searchers never return on a NaN objective (their loops
    end on comparisons, NO.f90:1557-1579) -- on the GPU that would be a kernel that never ends.  The machine stops the
    problem at the first NaN value instead; finite problems of the same batch are not affected."""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    src = US.DIAGQUAD.replace("const double t0 = dx * x[k] + (half - 0.5)", "const double t0 = dx * x[k] + (half < 0.0 && x[k] > 0.25 ? __builtin_nan(\"\") : 0.0)")
    assert src != US.DIAGQUAD
    n, B = 256, 8
    d, b = _quads(B, n, 3)
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    obj = NLO.compile_objective(src, "MyQuadratic", n, solver=NLO.LBFGS_, tune_like=NLO.DIAGQUAD)
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    ok = obj.solve(x, dd, bb, torch.tensor([0.5], dtype=torch.float64, device=dev), Precision=1e-8, MaxIteration=300)
    xn = torch.zeros_like(x)
    bad = obj.solve(xn, dd, bb, torch.tensor([-1.0], dtype=torch.float64, device=dev), Precision=1e-8, MaxIteration=300)
    torch.cuda.synchronize()
    assert int((ok["status"] == 0).sum()) == B
    # (the iterates run into x > 0.25 somewhere on the way to b / d: every problem meets a NaN and stops there)
    assert bool((bad["status"] == 4).all()) and bool(torch.isnan(bad["f"]).all()), bad["status"]
    assert bool((bad["nf"] <= ok["nf"]).all())


@pytest.mark.parametrize("inner,n,M", [("LBFGS", 512, 8), ("CG", 512, 8), ("LBFGS", 256, 4), ("LBFGS", 1024, 8), ("LBFGS", 200, 5)])
def test_augmented_lagrangian_around_an_objective_given_as_source_equals_the_builtin(inner, n, M):
    """fl_user_compile_auglag / fl_user_solve_auglag: the caller's objective (source text) inside the fused augmented-Lagrangian
    kernel with the library's block-sphere constraints -- BASELINE config 5's shape and others; restating the diagonal quadratic
    it equals fl_augmented_lagrangian_batched bit for bit (incl. the speculative objective-only trials: an element-wise
    objective takes part in them)."""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B = 6
    rng = np.random.default_rng(n + M)
    kappa = np.exp(rng.uniform(np.log(2), np.log(10), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    b = rng.uniform(-1, 1, (B, n))
    x0 = 0.05 + 0.1 * rng.random((B, n))
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    code = NLO.LBFGS_ if inner == "LBFGS" else NLO.CG
    obj = NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", n, solver=code, tune_like=NLO.DIAGQUAD, constrained=True)
    xu = torch.tensor(x0, device=dev)
    ou = obj.solve_auglag(xu, M, dd, bb, None, Precision=1e-8, MaxIteration=60)
    xb = torch.tensor(x0, device=dev)
    ob = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xb, M, dd, bb, UnconstrainedSolver="LBFGS" if inner == "LBFGS" else "ConjugateGradient",
                                 Precision=1e-8, MaxIteration=60)
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    for k in ("f", "cnorm2", "iters", "outer", "status", "nf", "ng", "lambda"):
        assert torch.equal(ou[k], ob[k]), k
    assert int(ou["nf"].min()) > 5 * int(ou["ng"].max()) // 4


@pytest.mark.parametrize("inner,n,M", [("LBFGS", 512, 8), ("CG", 512, 8), ("LBFGS", 256, 4), ("LBFGS", 1024, 8), ("LBFGS", 300, 5)])
def test_constraints_given_as_source_equal_the_builtin_family_and_the_oracle(inner, n, M):
    """the reference's c, cd callbacks (NO.f90:1928-1934) as a constraints FUNCTOR in the caller's source: block spheres restated
    by the caller = fl_augmented_lagrangian_batched = the oracle, bit for bit (the caller's constraints go through one masked
    reduction, the built-in ones through lane-group sums and speculative trials: same bits by construction)."""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B = 5
    rng = np.random.default_rng(n * M)
    kappa = np.exp(rng.uniform(np.log(2), np.log(10), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    b = rng.uniform(-1, 1, (B, n))
    x0 = 0.05 + 0.1 * rng.random((B, n))
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    code = NLO.LBFGS_ if inner == "LBFGS" else NLO.CG
    obj = NLO.compile_objective(US.DIAGQUAD + US.BLOCK_SPHERES, "MyQuadratic", n, solver=code, tune_like=NLO.DIAGQUAD, constraints_class="MySpheres")
    xu = torch.tensor(x0, device=dev)
    ou = obj.solve_auglag(xu, M, dd, bb, None, Precision=1e-8, MaxIteration=60)
    xb = torch.tensor(x0, device=dev)
    ob = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xb, M, dd, bb, UnconstrainedSolver="LBFGS" if inner == "LBFGS" else "ConjugateGradient",
                                 Precision=1e-8, MaxIteration=60)
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    for k in ("f", "cnorm2", "iters", "outer", "status", "nf", "ng", "lambda"):
        assert torch.equal(ou[k], ob[k]), k
    T, E = NLO.reduction_geometry(n)
    o = O.auglag_batch(O.LBFGS if inner == "LBFGS" else O.CG, O.DIAGQUAD, x0, M, d=d, b=b,
                       opts=O.defaults(precision=1e-8, maxit=60, c2=0.45 if inner != "LBFGS" else 0.9), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(xu.cpu().numpy().view(np.uint64), o["x"].view(np.uint64)) and np.array_equal(ou["nf"].cpu().numpy(), o["nf"])


@pytest.mark.parametrize("n,M", [(512, 8), (1000, 3), (130, 1)])
def test_linear_constraints_given_as_source_reach_the_kkt_point(n, M):
    """a family the library does not have: c_j = sum_{i = j mod M} x_i - 1.  For f = 1/2 sum d x^2 - sum b x the constrained
    minimiser is x_i = (b_i + nu_j) / d_i with nu_j = (1 - sum_{S_j} b/d) / sum_{S_j} 1/d: the augmented Lagrangian around the
    compiled objective + constraints lands on it (||c|| < Precision, x within 1e-8, multipliers = -nu)."""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B = 7
    rng = np.random.default_rng(n + M)
    d = 1.0 + 4.0 * rng.random((B, n))
    b = rng.uniform(-1, 1, (B, n))
    obj = NLO.compile_objective(US.DIAGQUAD + US.STRIDE_SUMS, "MyQuadratic", n, solver=NLO.LBFGS_, tune_like=NLO.DIAGQUAD, constraints_class="MyStrideSums")
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    out = obj.solve_auglag(x, M, torch.tensor(d, device=dev), torch.tensor(b, device=dev), None, Precision=1e-10, MaxIteration=200)
    torch.cuda.synchronize()
    xs = np.empty((B, n))
    nu = np.empty((B, M))
    for j in range(M):
        S = np.arange(j, n, M)
        nu[:, j] = (1.0 - (b[:, S] / d[:, S]).sum(1)) / (1.0 / d[:, S]).sum(1)
        xs[:, S] = (b[:, S] + nu[:, j][:, None]) / d[:, S]
    assert bool((out["status"] == 0).all()) and float(out["cnorm2"].max()) < 1e-20
    assert np.abs(x.cpu().numpy() - xs).max() < 1e-8
    # L = f - lambda.c: stationarity d x - b - lambda_j = 0 on S_j  ->  lambda_j = nu_j
    assert np.abs(out["lambda"].cpu().numpy() - nu).max() < 1e-7


# ---------------------------------------------------------------------------------------------------------------------------
# n > 4096: the STREAMING functor in the vectors-in-HBM kernel (csrc/fl_big.hpp; VERDICT r03 next #4c)
@pytest.mark.parametrize("solver,n,kw", [("LBFGS", 5000, {}), ("LBFGS", 10001, {"Memory": 5}), ("CG", 6000, {}), ("CG", 4097, {"Method": "PR"}),
                                         ("SD", 5000, {"MaxIteration": 60}), ("BFGS", 4500, {"MaxIteration": 40}), ("LBFGS", 70001, {"Strong": False})])
def test_streaming_quadratic_given_as_source_equals_the_builtin_beyond_n_4096(solver, n, kw):
    """the caller's objective asked one element pair at a time, the solver doing the loads, stores and fixed-order sums
    around it: a functor restating the diagonal quadratic gives the built-in kernel's bits (odd n: a padding element)"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B = 5
    d, b = _quads(B, n, n)
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    code = {"SD": NLO.SD, "CG": NLO.CG, "LBFGS": NLO.LBFGS_, "BFGS": NLO.BFGS_}[solver]
    obj = NLO.compile_objective(US.STREAM_DIAGQUAD, "MyBigQuadratic", n, solver=code)
    assert obj.geometry == NLO.reduction_geometry(n, code) and obj.geometry[0] == 1024
    kw = dict({"Precision": 1e-8, "MaxIteration": 200}, **kw)
    xu = torch.zeros(B, n, dtype=torch.float64, device=dev)
    ou = obj.solve(xu, dd, bb, **kw)
    xb = torch.zeros_like(xu)
    fn = {"SD": NLO.SteepestDescent, "CG": NLO.ConjugateGradient, "LBFGS": NLO.LBFGS, "BFGS": NLO.BFGS}[solver]
    ob = fn(NLO.DIAGQUAD, xb, dd, bb, **(dict(kw, ExactStep=0) if solver == "BFGS" else kw))
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    _same(ou, ob)
    assert int(ou["iters"].min()) > 3 and bool(torch.isfinite(ou["f"]).all())


@pytest.mark.parametrize("solver,n", [("LBFGS", 6000), ("CG", 4099), ("LBFGS", 20000)])
def test_streaming_neighbour_coupled_rosenbrock_equals_the_builtin_and_the_oracle(solver, n):
    """NEIGHBOURS = true: pair() reads x[e-1], x[e+2] from the row (the trial point stored in a pass of its own, barriers
    around the evaluation); the built-in kernel forms its neighbours from x0 + a p in one pass -- the same bits"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, (3, n))
    code = {"CG": NLO.CG, "LBFGS": NLO.LBFGS_}[solver]
    obj = NLO.compile_objective(US.STREAM_ROSENBROCK, "MyBigRosenbrock", n, solver=code)
    kw = {"Precision": 1e-9, "MaxIteration": 60}
    xu = torch.tensor(x0, device=dev)
    ou = obj.solve(xu, **kw)
    xb = torch.tensor(x0, device=dev)
    fn = {"CG": NLO.ConjugateGradient, "LBFGS": NLO.LBFGS}[solver]
    ob = fn(NLO.ROSENBROCK, xb, **kw)
    torch.cuda.synchronize()
    assert torch.equal(xu, xb)
    _same(ou, ob)
    if n <= 6000:
        T, E = obj.geometry
        o = O.solve_batch(O.CG if solver == "CG" else O.LBFGS, O.ROSENBROCK, x0,
                          opts=O.defaults(precision=1e-9, maxit=60, c2=0.45 if solver == "CG" else 0.9), sum_mode=O.TREE, threads=T, ept=E)
        assert np.array_equal(xu.cpu().numpy().view(np.uint64), o["x"].view(np.uint64))
        assert np.array_equal(ou["nf"].cpu().numpy(), o["nf"])


def test_streaming_functor_refusals():
    NLO = _nlo()
    with pytest.raises(NLO.FLError):  # the register path's functor is not the streaming one
        NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", 5000)
    with pytest.raises(NLO.FLError):  # not inside the augmented Lagrangian (register path only)
        NLO.compile_objective(US.STREAM_DIAGQUAD, "MyBigQuadratic", 5000, constrained=True)
