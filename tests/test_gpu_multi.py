"""-m gpu: fl_multi_solve -- a batch over all the GPUs of the node from ONE process, one host thread per shard (the C-ABI /
Fortran-level multi-device entry).  A one-GPU box rehearses it with several shards on the one device:
whatever the sharding -- one shard, three ragged contiguous blocks, five interleaved ones -- every problem is solved by
the same kernel, so all outputs equal the one-device batched entry's bit for bit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _quads(B, n, seed):
    rng = np.random.default_rng(seed)
    kappa = np.exp(rng.uniform(np.log(2), np.log(300), B))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    return np.ascontiguousarray(d), rng.uniform(-1, 1, (B, n))


@pytest.mark.parametrize("solver_name", ["LBFGS", "CG", "BFGS"])
def test_multi_solve_equals_the_one_device_entry_for_every_sharding(solver_name):
    NLO = _nlo()
    B, n = 61, 300
    d, b = _quads(B, n, 1)
    solver = {"LBFGS": NLO.LBFGS_, "CG": NLO.CG, "BFGS": NLO.BFGS_}[solver_name]
    kw = dict(Precision=1e-7, MaxIteration=40 if solver_name == "BFGS" else 1000)
    if solver_name == "BFGS":
        kw["ExactStep"] = 0
    dev = torch.device("cuda:0")
    xt = torch.zeros(B, n, dtype=torch.float64, device=dev)
    fused = {"LBFGS": NLO.LBFGS, "CG": NLO.ConjugateGradient, "BFGS": NLO.BFGS}[solver_name]
    ref = fused(NLO.DIAGQUAD, xt, torch.tensor(d, device=dev), torch.tensor(b, device=dev), **kw)
    assert NLO.FL.fl_multi_device_count() >= 1
    for nshards, inter in ((0, False), (1, False), (3, False), (5, True), (64, True)):
        x = np.zeros((B, n))
        out = NLO.multi_solve(solver, NLO.DIAGQUAD, x, d, b, nshards=nshards, interleaved=inter, **kw)
        assert np.array_equal(x, xt.cpu().numpy()), (nshards, inter)
        for k in ("f", "gg", "iters", "status", "nf", "ng"):
            assert np.array_equal(out[k], ref[k].cpu().numpy()), (k, nshards, inter)
    assert int(ref["iters"].min()) > 0


def test_multi_solve_augmented_lagrangian_config5_shape_sharded():
    """BASELINE config 5's shape (n = 512, 8 block spheres, L-BFGS inside), 24 problems over 4 interleaved shards"""
    NLO = _nlo()
    B, n, M = 24, 512, 8
    d, b = _quads(B, n, 2)
    d = 1.0 + (d - 1.0) * (9.0 / np.maximum(d[:, -1:] - 1.0, 1e-300))  # kappa = 10
    rng = np.random.default_rng(3)
    x0 = 0.05 + 0.1 * rng.random((B, n))
    dev = torch.device("cuda:0")
    xt = torch.tensor(x0, device=dev)
    ref = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xt, M, torch.tensor(d, device=dev), torch.tensor(b, device=dev),
                                  UnconstrainedSolver="LBFGS", Precision=1e-9)
    x = x0.copy()
    out = NLO.multi_solve(NLO.LBFGS_, NLO.DIAGQUAD, x, d, b, M=M, nshards=4, interleaved=True, Precision=1e-9)
    assert np.array_equal(x, xt.cpu().numpy())
    for k in ("f", "iters", "outer", "status", "nf", "ng", "cnorm2", "lambda"):
        assert np.array_equal(out[k], ref[k].cpu().numpy()), k
    assert np.all(out["status"] == 0) and float(np.sqrt(out["cnorm2"].max())) < 1e-9


def test_multi_solve_argument_errors():
    NLO = _nlo()
    x = np.zeros((4, 16))
    with pytest.raises(NLO.FLError):
        NLO.multi_solve(NLO.LBFGS_, NLO.DIAGQUAD, x)  # the diagonal quadratic needs d, b
