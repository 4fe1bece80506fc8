"""GPU parity of the LATENCY geometries (round 4): the fused SD / CG / L-BFGS kernels and the augmented Lagrangian around
CG / L-BFGS on more waves x fewer elements per thread -- what FL_GEOMETRY_AUTO takes for a batch that under-fills the
device (include/fl_nlopt.h, csrc/fl_solver_kernels.hip: select_fused_geometry).  Same bar as tests/test_gpu_parity.py:
BIT-EXACT against the oracle summing in the kernel's order (threads x elements per thread of the geometry in use), through
the C ABI.  Every candidate geometry of every n range is forced once (FL_FORCE_GEOMETRY, read by the library per call);
the policy itself (option, process policy, by batch) is tested below through fl_reduction_geometry_for_batch.
Reference: NO.f90:55-625 (solvers), 1286-1698 (line searchers), 2005-2241 (AugmentedLagrangian)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

CANDIDATES = {256: ["2x2"], 512: ["4x2", "2x4"], 1024: ["8x2", "4x4"], 2048: ["8x4"]}


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _te(geo):
    w, e = geo.split("x")
    return int(w) * 64, int(e)


def _quads(B, n, klo, khi, seed):
    rng = np.random.default_rng(seed)
    kappa = np.exp(rng.uniform(np.log(klo), np.log(khi), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / max(n - 1, 1))[None, :]
    return d, rng.uniform(-1, 1, (B, n))


def _gpu(fn, kind, x0, d, b, **kw):
    dev = torch.device("cuda:0")
    x = torch.tensor(x0, dtype=torch.float64, device=dev).contiguous()
    dd = torch.tensor(d, device=dev) if d is not None else None
    bb = torch.tensor(b, device=dev) if b is not None else None
    out = fn(kind, x, dd, bb, **kw)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if k != "workspace"}
    res["x"] = x.cpu().numpy()
    return res


def _assert_bitexact(g, o, keys=("iters", "status", "nf", "ng")):
    for k in keys:
        assert np.array_equal(g[k], o[k]), (k, g[k], o[k])
    assert np.array_equal(g["x"].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(g["f"].view(np.uint64), o["f"].view(np.uint64))


GEOS = [(n, geo) for n, gs in CANDIDATES.items() for geo in gs]


@pytest.mark.parametrize("n,geo", GEOS)
@pytest.mark.parametrize("solver,kw", [(O.LBFGS, {}), (O.LBFGS, {"f_fd": True, "Memory": 3}), (O.CG, {}), (O.CG, {"Method": "PR"}),
                                       (O.SD, {"MaxIteration": 100}), (O.LBFGS, {"Strong": False})])
def test_latency_geometries_bitexact_on_the_quadratics(monkeypatch, n, geo, solver, kw):
    NLO = _nlo()
    monkeypatch.setenv("FL_FORCE_GEOMETRY", geo)
    B = 6
    nn = n - 3 if solver == O.CG else n  # (odd n: unaligned rows, padded tail)
    d, b = _quads(B, nn, 10, 1000, 3 * n + solver)
    x0 = np.zeros((B, nn))
    fn = {O.SD: NLO.SteepestDescent, O.CG: NLO.ConjugateGradient, O.LBFGS: NLO.LBFGS}[solver]
    kw = dict({"Precision": 1e-7, "MaxIteration": 400}, **kw)
    g = _gpu(fn, NLO.DIAGQUAD, x0, d, b, **kw)
    T, E = _te(geo)
    oo = O.defaults(c2=0.45 if solver == O.CG else 0.9, precision=kw["Precision"], maxit=kw["MaxIteration"])
    if "Memory" in kw:
        oo.memory = kw["Memory"]
    if "Strong" in kw:
        oo.strong = int(kw["Strong"])
    if kw.get("Method") == "PR":
        oo.method = 1
    o = O.solve_batch(solver, O.DIAGQUAD, x0, d=d, b=b, opts=oo, use_ffd=bool(kw.get("f_fd", False)), sum_mode=O.TREE, threads=T, ept=E)
    _assert_bitexact(g, o)
    assert np.array_equal(g["gg"].view(np.uint64), o["gg"].view(np.uint64))


@pytest.mark.parametrize("n,geo", GEOS)
def test_latency_geometries_bitexact_on_rosenbrock(monkeypatch, n, geo):
    """the neighbour-coupled objective: x staged through LDS across several waves"""
    NLO = _nlo()
    monkeypatch.setenv("FL_FORCE_GEOMETRY", geo)
    rng = np.random.default_rng(n)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, (4, n - 1))
    g = _gpu(NLO.LBFGS, NLO.ROSENBROCK, x0, None, None, MaxIteration=80, Precision=1e-9)
    T, E = _te(geo)
    o = O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, opts=O.defaults(precision=1e-9, maxit=80), sum_mode=O.TREE, threads=T, ept=E)
    _assert_bitexact(g, o)


@pytest.mark.parametrize("n,geo", GEOS)
@pytest.mark.parametrize("inner", ["LBFGS", "ConjugateGradient"])
def test_latency_geometries_bitexact_augmented_lagrangian(monkeypatch, n, geo, inner):
    """BASELINE config 5's shape (n = 512, 8 block spheres) and the other n ranges: constraints by lane group (block widths
    32 / 64 / 128) with the speculative objective-only trials, across waves"""
    NLO = _nlo()
    monkeypatch.setenv("FL_FORCE_GEOMETRY", geo)
    M, B = 8, 4
    d, b = _quads(B, n, 2, 10, n + 5)
    rng = np.random.default_rng(n)
    x0 = 0.05 + 0.1 * rng.random((B, n))
    dev = torch.device("cuda:0")
    x = torch.tensor(x0, device=dev)
    out = NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, torch.tensor(d, device=dev), torch.tensor(b, device=dev),
                                  UnconstrainedSolver=inner, Precision=1e-8, MaxIteration=60)
    torch.cuda.synchronize()
    T, E = _te(geo)
    o = O.auglag_batch(O.LBFGS if inner == "LBFGS" else O.CG, O.DIAGQUAD, x0, M, d=d, b=b,
                       opts=O.defaults(precision=1e-8, maxit=60, c2=0.45 if inner != "LBFGS" else 0.9), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x.cpu().numpy().view(np.uint64), o["x"].view(np.uint64))
    for k, ok in (("nf", "nf"), ("ng", "ng"), ("iters", "iters"), ("outer", "outer")):
        assert np.array_equal(out[k].cpu().numpy(), o[ok]), k
    assert np.array_equal(out["lambda"].cpu().numpy().view(np.uint64), o["lam"].view(np.uint64))
    assert np.array_equal(out["cnorm2"].cpu().numpy().view(np.uint64), o["cnorm2"].view(np.uint64))


def test_geometry_policy_by_option_process_and_batch():
    """fl_options.geometry beats the process policy; AUTO answers by batch; THROUGHPUT is the n-only geometry; a solve under
    each policy equals the oracle in the geometry fl_reduction_geometry_for_batch reports for it"""
    NLO = _nlo()
    n = 512
    thr = NLO.reduction_geometry(n, NLO.LBFGS_)
    assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=8, Geometry="throughput") == thr == (64, 8)
    lat = NLO.reduction_geometry(n, NLO.LBFGS_, batch=1 << 20, Geometry="latency")
    assert lat[0] > thr[0] and lat[0] * lat[1] == thr[0] * thr[1]  # more threads, same padded row
    assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=1 << 20, Geometry="auto") == thr  # a full chip: throughput
    assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=8, Geometry="auto") != thr  # eight problems: latency
    old = NLO.set_geometry_policy("throughput")
    try:
        assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=8) == thr
        assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=8, Geometry="latency") == lat
        assert NLO.set_geometry_policy("auto") == "throughput"
        assert NLO.reduction_geometry(n, NLO.LBFGS_, batch=8) != thr
    finally:
        NLO.set_geometry_policy(old)
    # dense solvers and sizes without candidates: always the n-only geometry
    assert NLO.reduction_geometry(n, NLO.BFGS_, batch=8, Geometry="latency") == NLO.reduction_geometry(n, NLO.BFGS_)
    assert NLO.reduction_geometry(64, NLO.LBFGS_, batch=8, Geometry="latency") == NLO.reduction_geometry(64, NLO.LBFGS_)
    assert NLO.reduction_geometry(4096, NLO.LBFGS_, batch=8, Geometry="latency") == NLO.reduction_geometry(4096, NLO.LBFGS_)
    d, b = _quads(5, n, 10, 100, 1)
    x0 = np.zeros((5, n))
    seen = set()
    for pol in ("throughput", "latency", "auto"):
        T, E = NLO.reduction_geometry(n, NLO.LBFGS_, batch=5, Geometry=pol)
        seen.add((T, E))
        g = _gpu(NLO.LBFGS, NLO.DIAGQUAD, x0, d, b, Precision=1e-7, MaxIteration=300, Geometry=pol)
        o = O.solve_batch(O.LBFGS, O.DIAGQUAD, x0, d=d, b=b, opts=O.defaults(precision=1e-7, maxit=300), sum_mode=O.TREE, threads=T, ept=E)
        _assert_bitexact(g, o)
    assert len(seen) >= 2


def test_auto_geometry_on_a_share_of_config5(monkeypatch):
    """one GPU's share of BASELINE config 5 at eight GPUs (1024 problems, n = 512, M = 8): AUTO takes a latency geometry, the
    result equals the oracle in that geometry on a subset, and agrees with the throughput geometry's result to the
    north-star tolerance (same minimiser, other summation order)"""
    NLO = _nlo()
    monkeypatch.delenv("FL_FORCE_GEOMETRY", raising=False)
    n, M, B = 512, 8, 1024
    dev = torch.device("cuda:0")
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty_like(d)
    x0 = torch.empty_like(d)
    NLO.synth_diag_spectrum(20261003, d, 2.0, 10.0)
    NLO.synth_uniform(20261003, b, -1.0, 1.0)
    NLO.synth_uniform(20261010, x0, 0.05, 0.15)
    res = {}
    for pol in ("auto", "throughput"):
        x = x0.clone()
        out = NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, d, b, UnconstrainedSolver="LBFGS", Precision=1e-10, Geometry=pol)
        torch.cuda.synchronize()
        res[pol] = (x.cpu().numpy(), {k: v.cpu().numpy() for k, v in out.items() if k != "workspace"})
    T, E = NLO.reduction_geometry(n, NLO.LBFGS_, batch=B, constrained=True, Geometry="auto")
    assert (T, E) != NLO.reduction_geometry(n, NLO.LBFGS_), "a 1024-problem share must not run the full-chip geometry"
    S = 24
    o = O.auglag_batch(O.LBFGS, O.DIAGQUAD, x0[:S].cpu().numpy(), M, d=d[:S].cpu().numpy(), b=b[:S].cpu().numpy(),
                       opts=O.defaults(precision=1e-10), sum_mode=O.TREE, threads=T, ept=E)
    xa, oa = res["auto"]
    assert np.array_equal(xa[:S].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(oa["nf"][:S], o["nf"]) and np.array_equal(oa["outer"][:S], o["outer"])
    xt, ot = res["throughput"]
    assert np.all(oa["status"] == 0) and np.all(ot["status"] == 0)
    fa = 0.5 * (d.cpu().numpy() * xa * xa).sum(1) - (b.cpu().numpy() * xa).sum(1)
    ft = 0.5 * (d.cpu().numpy() * xt * xt).sum(1) - (b.cpu().numpy() * xt).sum(1)
    assert np.max(np.abs(fa - ft) / np.abs(ft)) < 1e-10
    assert np.max(np.abs(xa - xt)) < 1e-8 * max(1.0, float(np.abs(xt).max()) * np.sqrt(n))


@pytest.mark.parametrize("n,M", [(512, 8), (512, 4), (512, 16), (256, 8), (256, 4), (384, 6), (200, 5)])
@pytest.mark.parametrize("inner", ["LBFGS", "ConjugateGradient"])
@pytest.mark.parametrize("kind", ["DIAGQUAD", "QUARTIC"])
def test_replicated_groups_are_invisible_in_the_results(monkeypatch, n, M, inner, kind):
    """fl_solve_rep_kernel: 2 / 4 complete copies of the machine per problem share the objective-only shrink loop by trial
    (Solver::fast_forward_wide).  Whatever the number of copies, every output -- minimiser, multipliers, objective, c.c,
    counts -- has the bits of the unreplicated kernel and of the oracle in the throughput geometry.  Block widths 32 / 64 /
    128 (lane-group constraints, replicated) and one shape without them (n = 200, M = 5: the general path, never
    replicated -- FL_FORCE_REPLICAS must be ignored there)."""
    NLO = _nlo()
    monkeypatch.delenv("FL_FORCE_GEOMETRY", raising=False)
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
                                      Precision=1e-8, MaxIteration=40, Geometry="throughput")
        torch.cuda.synchronize()
        res[rep] = dict({k: v.cpu().numpy() for k, v in out.items() if k != "workspace"}, x=x.cpu().numpy())
    for rep in ("2", "4"):
        for k, v in res["1"].items():
            assert np.array_equal(v.view(np.uint64) if v.dtype == np.float64 else v, res[rep][k].view(np.uint64) if v.dtype == np.float64 else res[rep][k]), (rep, k)
    T, E = NLO.reduction_geometry(n)
    okind = O.DIAGQUAD if kind == "DIAGQUAD" else O.QUARTIC
    o = O.auglag_batch(O.LBFGS if inner == "LBFGS" else O.CG, okind, x0, M, d=d, b=b,
                       opts=O.defaults(precision=1e-8, maxit=40, c2=0.45 if inner != "LBFGS" else 0.9), sum_mode=O.TREE, threads=T, ept=E)
    g = res["4"]
    assert np.array_equal(g["x"].view(np.uint64), o["x"].view(np.uint64))
    assert np.array_equal(g["nf"], o["nf"]) and np.array_equal(g["ng"], o["ng"]) and np.array_equal(g["outer"], o["outer"])
    assert np.array_equal(g["lambda"].view(np.uint64), o["lam"].view(np.uint64))
    assert np.array_equal(g["cnorm2"].view(np.uint64), o["cnorm2"].view(np.uint64))
