"""GPU parity of the HELPER-WAVE kernels (round 4; csrc/fl_solver_launch.hpp: fl_solve_rep_kernel, Solver::fast_forward_wide_cs).

For batches that under-fill the device (one GPU's share of BASELINE config 5 on eight GPUs: 1024 problems) the fused
augmented-Lagrangian kernels of the one-wave geometry run a master wave plus 1 or 3 helper waves per problem; the helpers
take their share of the line search's objective-only shrink loop BY TRIAL (the reference walks a <- a / incrmt until
Armijo holds, NO.f90:1517-1521, 1636-1640: hundreds of steps after every restart of the inner solver).  Every trial is still
summed inside one wave in the throughput geometry's order, so the helpers must be INVISIBLE: every output bit for bit that
of the unhelped kernel and of the oracle (threads x elements per thread of fl_reduction_geometry).  The library picks the
number of waves by batch size; FL_FORCE_REPLICAS (read per call) forces it here.
Reference: AugmentedLagrangian NO.f90:2005-2241, StrongWolfe(_fdwithf) 1462-1698."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _quads(B, n, klo, khi, seed):
    rng = np.random.default_rng(seed)
    kappa = np.exp(rng.uniform(np.log(klo), np.log(khi), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / max(n - 1, 1))[None, :]
    return d, rng.uniform(-1, 1, (B, n))


@pytest.mark.parametrize("n,M", [(512, 8), (512, 4), (512, 16), (256, 8), (256, 4), (384, 6), (200, 5)])
@pytest.mark.parametrize("inner", ["LBFGS", "ConjugateGradient"])
@pytest.mark.parametrize("kind", ["DIAGQUAD", "QUARTIC"])
def test_helper_waves_are_invisible_in_the_results(monkeypatch, n, M, inner, kind):
    """1, 2 and 4 waves per problem: every output -- minimiser, multipliers, objective, c.c, counts -- has the same bits, and
    they are the oracle's in the throughput geometry.  Block widths 32 / 64 / 128 (lane-group constraints: helped) and one
    shape without them (n = 200, M = 5: the general path, never helped -- FL_FORCE_REPLICAS must be ignored there).  Both
    register layouts of the line search's x0 (LDS row: 1 x 8 diagonal quadratic; registers: the others, published per loop)."""
    NLO = _nlo()
    B = 5
    kobj = getattr(NLO, kind)
    d = b = None
    rng = np.random.default_rng(n + M)
    if kind == "DIAGQUAD":
        d, b = _quads(B, n, 2, 10, n + M)
        x0 = 0.05 + 0.1 * rng.random((B, n))
    else:
        x0 = rng.random((B, n))
    dev = torch.device("cuda:0")
    res = {}
    for rep in ("1", "2", "4"):
        monkeypatch.setenv("FL_FORCE_REPLICAS", rep)
        x = torch.tensor(x0, device=dev)
        out = NLO.AugmentedLagrangian(kobj, x, M, torch.tensor(d, device=dev) if d is not None else None,
                                      torch.tensor(b, device=dev) if b is not None else None, UnconstrainedSolver=inner,
                                      Precision=1e-8, MaxIteration=40)
        torch.cuda.synchronize()
        res[rep] = dict({k: v.cpu().numpy() for k, v in out.items() if k != "workspace"}, x=x.cpu().numpy())
    for rep in ("2", "4"):
        for k, v in res["1"].items():
            same = np.array_equal(v.view(np.uint64), res[rep][k].view(np.uint64)) if v.dtype == np.float64 else np.array_equal(v, res[rep][k])
            assert same, (rep, k)
    T, E = NLO.reduction_geometry(n)
    okind = O.DIAGQUAD if kind == "DIAGQUAD" else O.QUARTIC
    o = O.auglag_batch(O.LBFGS if inner == "LBFGS" else O.CG, okind, x0, M, d=d, b=b,
                       opts=O.defaults(precision=1e-8, maxit=40, c2=0.45 if inner != "LBFGS" else 0.9), sum_mode=O.TREE, threads=T, ept=E)
    g = res["4"]
    assert np.array_equal(g["x"].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(g["nf"], o["nf"]) and np.array_equal(g["ng"], o["ng"]) and np.array_equal(g["outer"], o["outer"])
    assert np.array_equal(g["lambda"].view(np.uint64), o["lam"].view(np.uint64))
    assert np.array_equal(g["cnorm2"].view(np.uint64), o["cnorm2"].view(np.uint64))


@pytest.mark.parametrize("strong", [True, False])
def test_helper_waves_through_long_shrink_loops(monkeypatch, strong):
    """BASELINE config 5's family solved to the end (Precision 1e-10, ~150 outer rounds): after every restart of the inner
    solver the first searches shrink a through hundreds of trials -- many passes of the shared loop, exits at every position
    of a pass; strong Wolfe (SW_V_F, NO.f90:1517-1521 / 1636-1640) and Wolfe (W_SHRINK, NO.f90:1325-1329)"""
    NLO = _nlo()
    n, M, B = 512, 8, 6
    d, b = _quads(B, n, 2, 10, 77)
    rng = np.random.default_rng(3)
    x0 = 0.05 + 0.1 * rng.random((B, n))
    dev = torch.device("cuda:0")
    res = {}
    for rep in ("1", "4"):
        monkeypatch.setenv("FL_FORCE_REPLICAS", rep)
        x = torch.tensor(x0, device=dev)
        out = NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, torch.tensor(d, device=dev), torch.tensor(b, device=dev),
                                      UnconstrainedSolver="LBFGS", Precision=1e-10, Strong=strong)
        torch.cuda.synchronize()
        res[rep] = dict({k: v.cpu().numpy() for k, v in out.items() if k != "workspace"}, x=x.cpu().numpy())
    assert np.array_equal(res["1"]["x"].view(np.uint64), res["4"]["x"].view(np.uint64))
    assert np.array_equal(res["1"]["nf"], res["4"]["nf"]) and np.array_equal(res["1"]["ng"], res["4"]["ng"])
    assert res["1"]["nf"].sum() > 5 * res["1"]["ng"].sum()  # (objective-only trials dominate: the loop was exercised)
    T, E = NLO.reduction_geometry(n)
    oo = O.defaults(precision=1e-10)
    oo.strong = int(strong)
    o = O.auglag_batch(O.LBFGS, O.DIAGQUAD, x0, M, d=d, b=b, opts=oo, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(res["4"]["x"].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(res["4"]["nf"], o["nf"]) and np.array_equal(res["4"]["outer"], o["outer"])


def test_a_share_of_config5_runs_helped_and_equals_the_full_batch_rows(monkeypatch):
    """one GPU's share of BASELINE config 5 at eight GPUs (1024 problems, n = 512, M = 8): the library helps it by itself
    (batch <= the break-even); its rows equal the same problems solved inside a batch of 8192 (unhelped) bit for bit, and the
    oracle on a subset -- sharding a batch over GPUs changes no bit although the kernels differ"""
    NLO = _nlo()
    monkeypatch.delenv("FL_FORCE_REPLICAS", raising=False)
    n, M, B = 512, 8, 8192
    dev = torch.device("cuda:0")
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty_like(d)
    x0 = torch.empty_like(d)
    NLO.synth_diag_spectrum(20261003, d, 2.0, 10.0)
    NLO.synth_uniform(20261003, b, -1.0, 1.0)
    NLO.synth_uniform(20261010, x0, 0.05, 0.15)
    xf = x0.clone()
    of = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xf, M, d, b, UnconstrainedSolver="LBFGS", Precision=1e-10)
    S = 1024
    xs = x0[:S].clone()
    os_ = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xs, M, d[:S].contiguous(), b[:S].contiguous(), UnconstrainedSolver="LBFGS", Precision=1e-10)
    torch.cuda.synchronize()
    assert torch.equal(xs, xf[:S])
    for k in ("f", "nf", "ng", "iters", "outer", "lambda", "cnorm2", "status"):
        assert torch.equal(os_[k], of[k][:S]), k
    assert int(os_["status"].sum()) == 0
    T, E = NLO.reduction_geometry(n)
    Q = 16
    o = O.auglag_batch(O.LBFGS, O.DIAGQUAD, x0[:Q].cpu().numpy(), M, d=d[:Q].cpu().numpy(), b=b[:Q].cpu().numpy(),
                       opts=O.defaults(precision=1e-10), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(xs[:Q].cpu().numpy().view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(os_["nf"][:Q].cpu().numpy(), o["nf"])


def test_staged_launches_equal_the_single_launch(monkeypatch):
    """BASELINE config 5 at full size (8192 x n = 512, M = 8): the library runs it STAGED -- the plain kernel until at most
    2560 problems are unfinished, those pause at their next outer iteration's boundary and continue with one helper wave, the
    last 1280 with three (csrc/fl_solver_kernels.hip).  Pausing there changes no bit (the reference starts a fresh inner solve
    from (x, lambda, miu) at that point anyway, NO.f90:2155-2157): every output equals the single launch's (FL_AUG_STAGED=0),
    no problem is left paused, and with nothing to pause (one outer iteration) the empty stages do no harm."""
    NLO = _nlo()
    monkeypatch.delenv("FL_FORCE_REPLICAS", raising=False)
    n, M, B = 512, 8, 8192
    dev = torch.device("cuda:0")
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty_like(d)
    x0 = torch.empty_like(d)
    NLO.synth_diag_spectrum(20261003, d, 2.0, 10.0)
    NLO.synth_uniform(20261003, b, -1.0, 1.0)
    NLO.synth_uniform(20261010, x0, 0.05, 0.15)
    for kw in ({"Precision": 1e-10}, {"Precision": 1e-10, "MaxIteration": 1}, {"Precision": 1e-10, "MaxIteration": 7}):
        res = {}
        for staged in ("1", "0"):
            monkeypatch.setenv("FL_AUG_STAGED", staged)
            x = x0.clone()
            out = NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, d, b, UnconstrainedSolver="LBFGS", **kw)
            torch.cuda.synchronize()
            res[staged] = (x, out)
        assert torch.equal(res["1"][0], res["0"][0]), kw
        for k in ("f", "nf", "ng", "iters", "outer", "lambda", "cnorm2", "status"):
            assert torch.equal(res["1"][1][k], res["0"][1][k]), (kw, k)
        assert int((res["1"][1]["status"] == 3).sum()) == 0
    monkeypatch.delenv("FL_AUG_STAGED", raising=False)


@pytest.mark.parametrize("n,M,inner,kind,B", [(256, 8, "ConjugateGradient", "DIAGQUAD", 6000), (256, 4, "LBFGS", "QUARTIC", 3000), (384, 3, "LBFGS", "DIAGQUAD", 2000)])
def test_staged_launches_on_the_other_helper_kernels(monkeypatch, n, M, inner, kind, B):
    """the one-wave x 4 geometry (n <= 256), ConjugateGradient inside, the quartic objective (x0 in registers: published to the
    helpers per loop), block width 128: staged = single launch, bit for bit, at batches that take two and three stages"""
    NLO = _nlo()
    monkeypatch.delenv("FL_FORCE_REPLICAS", raising=False)
    dev = torch.device("cuda:0")
    kobj = getattr(NLO, kind)
    x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
    d = b = None
    if kind == "DIAGQUAD":
        d = torch.empty_like(x0)
        b = torch.empty_like(x0)
        NLO.synth_diag_spectrum(7, d, 2.0, 10.0)
        NLO.synth_uniform(7, b, -1.0, 1.0)
        NLO.synth_uniform(8, x0, 0.05, 0.15)
    else:
        NLO.synth_uniform(8, x0, 0.0, 1.0)
    solver = NLO.LBFGS_ if inner == "LBFGS" else NLO.CG
    assert len(NLO.augmented_lagrangian_launch_plan(solver, kobj, B, n, M)) >= 2
    res = {}
    for staged in ("1", "0"):
        monkeypatch.setenv("FL_AUG_STAGED", staged)
        x = x0.clone()
        out = NLO.AugmentedLagrangian(kobj, x, M, d, b, UnconstrainedSolver=inner, Precision=1e-9, MaxIteration=40)
        torch.cuda.synchronize()
        res[staged] = (x, out)
    assert torch.equal(res["1"][0], res["0"][0])
    for k in ("f", "nf", "ng", "iters", "outer", "lambda", "cnorm2", "status"):
        assert torch.equal(res["1"][1][k], res["0"][1][k]), k
    assert int((res["1"][1]["status"] == 3).sum()) == 0


def test_a_zoom_that_never_narrows_ends_the_problem_instead_of_the_kernel_running_for_ever(monkeypatch):
    """FL_STATUS_STALLED (include/fl_nlopt.h).  Found in round 4: problem 1724 of the quartic + 4 block spheres family (n = 256,
    x0 uniform in (0, 1), Philox seed 8) converges to x = 1/8, and in its tenth outer round the reference's zoom
    (NO.f90:1557-1579, no iteration limit) returns to the same two points for ever, both end slopes positive at the rounding
    level -- the oracle, restating the reference, never returns on it either (which is why it is skipped below), and the
    round-3 kernel ran until it was killed.  Now 65 536 consecutive zoom trials end the problem; its neighbours are solved as
    ever (bit for bit the oracle), with and without helper waves."""
    import philox_ref as P
    NLO = _nlo()
    dev = torch.device("cuda:0")
    n, M, lo, hi, bad = 256, 4, 1700, 1764, 1724
    x0 = P.philox_uniform(8, 3000, n, 0.0, 1.0)[lo:hi]
    res = {}
    for rep in ("1", "4"):
        monkeypatch.setenv("FL_FORCE_REPLICAS", rep)
        x = torch.tensor(x0, device=dev)
        out = NLO.AugmentedLagrangian(NLO.QUARTIC, x, M, UnconstrainedSolver="LBFGS", Precision=1e-9, MaxIteration=40)
        torch.cuda.synchronize()
        res[rep] = dict({k: v.cpu().numpy() for k, v in out.items() if k != "workspace"}, x=x.cpu().numpy())
    for k, v in res["1"].items():
        assert np.array_equal(v, res["4"][k]), k
    st = res["1"]["status"]
    assert st[bad - lo] == 5 and np.all(np.delete(st, bad - lo) != 5), st
    assert np.all(np.isfinite(res["1"]["x"])) and abs(np.abs(res["1"]["x"][bad - lo]).mean() - 0.125) < 1e-3  # (x -> +-1/8, it stopped next to it)
    keep = np.array([k for k in range(hi - lo) if k != bad - lo])
    T, E = NLO.reduction_geometry(n)
    o = O.auglag_batch(O.LBFGS, O.QUARTIC, x0[keep], M, opts=O.defaults(precision=1e-9, maxit=40), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(res["1"]["x"][keep].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(res["1"]["nf"][keep], o["nf"]) and np.array_equal(res["1"]["outer"][keep], o["outer"])
