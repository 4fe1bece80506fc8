"""-m gpu: the COOPERATIVE form of the fused solve beyond n = 4096 (csrc/fl_big.hpp: fl_big_solve_kernel with groups > 1).  Few
problems of very large n -- the reference's callers typically solve ONE problem of any dim (NO.f90:398-625 takes one x) -- share
each problem among several workgroups: the whole solve stays one launch and a line-search trial runs at the chip's bandwidth
instead of one CU's.  Every sum is the workgroups' sums added left to right: the oracle's tree order with `groups`
(flo_set_sum_groups), so the results are held to the oracle bit for bit for the group count the library reports."""
import time

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
    kappa = np.exp(rng.uniform(np.log(10), np.log(200), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / max(n - 1, 1))[None, :]
    return d, rng.uniform(-1, 1, (B, n))


@pytest.mark.parametrize("solver_name,kind,n,B,force", [
    ("LBFGS", "DIAGQUAD", 20001, 2, None), ("CG", "DIAGQUAD", 40000, 3, None), ("SD", "QUARTIC", 30000, 1, None),
    ("LBFGS", "QUARTIC", 16385, 2, None), ("LBFGS", "DIAGQUAD", 9000, 2, 2), ("CG", "DIAGQUAD", 20001, 2, 7)])
def test_fused_cooperative_solve_is_the_oracle_with_groups_bit_for_bit(solver_name, kind, n, B, force, monkeypatch):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    if force is not None:
        monkeypatch.setenv("FL_COOP_GROUPS", str(force))
    else:
        monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
    solver = {"LBFGS": NLO.LBFGS_, "CG": NLO.CG, "SD": NLO.SD}[solver_name]
    osolver = {"LBFGS": O.LBFGS, "CG": O.CG, "SD": O.SD}[solver_name]
    okind = O.DIAGQUAD if kind == "DIAGQUAD" else O.QUARTIC
    G = NLO.cooperative_groups(solver, okind, B, n)
    assert G > 1, G
    rng = np.random.default_rng(n)
    d = b = dd = bb = None
    if kind == "DIAGQUAD":
        d, b = _quads(B, n, n)
        dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
        x0 = np.zeros((B, n))
    else:
        x0 = rng.uniform(0.2, 1.0, (B, n))
    kw = dict(Precision=1e-6, MaxIteration=30 if solver_name == "SD" else 200)
    fn = {"SD": NLO.SteepestDescent, "CG": NLO.ConjugateGradient, "LBFGS": NLO.LBFGS}[solver_name]
    x = torch.tensor(x0, device=dev)
    out = fn(okind, x, dd, bb, **kw)
    torch.cuda.synchronize()
    assert np.all(out["status"].cpu().numpy() >= 0)
    T, E = NLO.reduction_geometry(n, solver)
    oo = O.defaults(precision=1e-6, maxit=kw["MaxIteration"], c2=0.45 if solver_name == "CG" else 0.9)
    lib = O.lib()
    lib.flo_set_sum_groups(G)
    try:
        o = O.solve_batch(osolver, okind, x0, d=d, b=b, opts=oo, sum_mode=O.TREE, threads=T, ept=E, nthreads=1)
    finally:
        lib.flo_set_sum_groups(1)
    assert np.array_equal(x.cpu().numpy().view(np.uint64), o["x"].view(np.uint64)), G
    assert np.array_equal(out["iters"].cpu().numpy(), o["iters"]) and np.array_equal(out["nf"].cpu().numpy(), o["nf"])
    assert np.array_equal(out["ng"].cpu().numpy(), o["ng"]) and int(o["iters"].min()) > 3
    # one workgroup per problem (FL_COOP_GROUPS=1): the plain path's bits -- another, equally reproducible, order of the same sums
    monkeypatch.setenv("FL_COOP_GROUPS", "1")
    assert NLO.cooperative_groups(solver, okind, B, n) == 1
    x1 = torch.tensor(x0, device=dev)
    fn(okind, x1, dd, bb, **kw)
    torch.cuda.synchronize()
    o1 = O.solve_batch(osolver, okind, x0, d=d, b=b, opts=oo, sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x1.cpu().numpy().view(np.uint64), o1["x"].view(np.uint64))
    assert np.abs(x1.cpu().numpy() - x.cpu().numpy()).max() < 1e-5


def test_which_batches_get_the_cooperative_form(monkeypatch):
    NLO = _nlo()
    monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
    cg = NLO.cooperative_groups
    assert cg(NLO.LBFGS_, O.DIAGQUAD, 1, 1 << 20) == 64          # one huge problem: 64 workgroups (more costs more in barriers than it gains)
    assert cg(NLO.LBFGS_, O.DIAGQUAD, 1, 10000) == 1             # too few slots to share out (n <= 14336)
    assert cg(NLO.LBFGS_, O.DIAGQUAD, 200, 1 << 20) == 1         # enough problems to fill the chip by themselves
    assert cg(NLO.BFGS_, O.DIAGQUAD, 1, 16000) == 1              # dense H: no
    assert cg(NLO.LBFGS_, O.ROSENBROCK, 1, 1 << 20) == 1         # neighbour reads across workgroups: no
    assert cg(NLO.CG, O.QUARTIC, 4, 1 << 18) > 1
    assert cg(NLO.LBFGS_, O.DIAGQUAD, 1, 4096) == 1


def test_streaming_functor_under_the_cooperative_form_equals_the_builtin(monkeypatch):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
    B, n = 2, 50001
    d, b = _quads(B, n, 5)
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    assert NLO.cooperative_groups(NLO.LBFGS_, O.DIAGQUAD, B, n) > 1
    obj = NLO.compile_objective(US.STREAM_DIAGQUAD, "MyBigQuadratic", n, solver=NLO.LBFGS_)
    xu = torch.zeros(B, n, dtype=torch.float64, device=dev)
    ou = obj.solve(xu, dd, bb, Precision=1e-7, MaxIteration=200)
    xb = torch.zeros_like(xu)
    ob = NLO.LBFGS(NLO.DIAGQUAD, xb, dd, bb, Precision=1e-7, MaxIteration=200)
    torch.cuda.synchronize()
    assert torch.equal(xu, xb) and torch.equal(ou["nf"], ob["nf"]) and torch.equal(ou["f"], ob["f"])
    # a functor that reads its neighbours stays with one workgroup per problem (and with the built-in's bits)
    x0 = 1.0 + 0.1 * np.random.default_rng(1).uniform(-1, 1, (1, n))
    objr = NLO.compile_objective(US.STREAM_ROSENBROCK, "MyBigRosenbrock", n, solver=NLO.LBFGS_)
    xr = torch.tensor(x0, device=dev)
    orr = objr.solve(xr, Precision=1e-8, MaxIteration=40)
    xq = torch.tensor(x0, device=dev)
    oq = NLO.LBFGS(NLO.ROSENBROCK, xq, Precision=1e-8, MaxIteration=40)
    torch.cuda.synchronize()
    assert torch.equal(xr, xq) and torch.equal(orr["nf"], oq["nf"])


def test_one_problem_of_a_million_unknowns_fills_the_chip(monkeypatch, capsys):
    """the reference's typical call -- ONE problem -- at n = 2^20: same minimiser with and without the cooperative form (the
    sums' orders differ), and the launch is many times shorter"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    n = 1 << 20
    d, b = _quads(1, n, 3)
    dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
    res, ms = {}, {}
    monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
    G = NLO.cooperative_groups(NLO.LBFGS_, O.DIAGQUAD, 1, n)
    for tag, env in (("coop", None), ("one", "1")):
        if env is None:
            monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
        else:
            monkeypatch.setenv("FL_COOP_GROUPS", env)
        ws = None
        for rep in range(2):  # (the first call pays the allocations)
            x = torch.zeros(1, n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = NLO.LBFGS(NLO.DIAGQUAD, x, dd, bb, Precision=1e-6, MaxIteration=60, workspace_=ws)
            torch.cuda.synchronize()
            ms[tag] = (time.perf_counter() - t) * 1e3
            ws = out["workspace"]
        res[tag] = (x.cpu().numpy(), int(out["iters"][0]), int(out["nf"][0]), int(out["status"][0]))
    with capsys.disabled():
        print(f"\n[cooperative] n = 2^20, one problem: {G} workgroups {ms['coop']:.1f} ms ({res['coop'][1]} iterations, {res['coop'][2]} evaluations), "
              f"one workgroup {ms['one']:.1f} ms ({res['one'][1]} iterations)")
    assert res["coop"][3] >= 0 and res["one"][3] >= 0
    assert np.abs(res["coop"][0] - res["one"][0]).max() < 1e-5
    assert ms["coop"] * 5 < ms["one"]


def test_shards_running_side_by_side_on_one_device_size_their_groups_for_the_devices_load(monkeypatch):
    """fl_multi_solve runs up to four shards per device at once: every shard's cooperative launch must leave room for the others'
    workgroups (their barriers spin) -- the groups are chosen from the device's load, not the shard's, which also makes the
    sharded result the one-call result bit for bit"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    monkeypatch.delenv("FL_COOP_GROUPS", raising=False)
    B, n = 4, 30001
    d, b = _quads(B, n, 11)
    assert NLO.cooperative_groups(NLO.LBFGS_, O.DIAGQUAD, B, n) > 1
    xh = np.zeros((B, n))
    om = NLO.multi_solve(NLO.LBFGS_, NLO.DIAGQUAD, xh, d, b, nshards=4, Precision=1e-6, MaxIteration=100)
    assert np.all(om["status"] >= 0), om["status"]
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    o1 = NLO.LBFGS(NLO.DIAGQUAD, x, torch.tensor(d, device=dev), torch.tensor(b, device=dev), Precision=1e-6, MaxIteration=100)
    torch.cuda.synchronize()
    assert np.array_equal(xh.view(np.uint64), x.cpu().numpy().view(np.uint64))
    assert np.array_equal(om["nf"], o1["nf"].cpu().numpy())
