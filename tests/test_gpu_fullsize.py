"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle would need
minutes for the whole batch): every problem meets the reference's convergence test, the returned gradient
norm is the true one, a second solve from the minimiser is a no-op (idempotence), counts are consistent, and a
sampled subset equals the oracle bit for bit.  Plus the argument edge cases (empty / oversized inputs)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _inputs(B, n, dev):
    import FortranLibrary.NonlinearOptimization as NLO
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty(B, n, dtype=torch.float64, device=dev)
    NLO.synth_diag_spectrum(20261003, d, 10.0, 1000.0)
    NLO.synth_uniform(20261003, b, -1.0, 1.0)
    return d, b


def test_headline_batch_65536_n1024_lbfgs_properties():
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n, m, prec = 65536, 1024, 10, 1e-6
    d, b = _inputs(B, n, dev)
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    ws = NLO.workspace(B, n, m, dev)
    out = NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=prec, MaxIteration=3000)
    torch.cuda.synchronize()
    assert bool((out["status"] == NLO.CONVERGED).all())
    g = d * x - b
    gg = (g * g).sum(1)
    assert bool((gg < prec * prec * (1 + 1e-9)).all())                      # the reference's test, recomputed
    assert torch.allclose(out["gg"], gg, rtol=1e-9, atol=0)                 # reported g.g is the true one
    f = 0.5 * (d * x * x).sum(1) - (b * x).sum(1)
    assert torch.allclose(out["f"], f, rtol=1e-12, atol=0)
    fstar = -0.5 * (b * b / d).sum(1)
    assert bool(((out["f"] - fstar) <= 1e-10 * fstar.abs()).all()) and bool((out["f"] >= fstar - 1e-12 * fstar.abs()).all())
    assert bool((out["nf"] >= out["iters"]).all()) and bool((out["ng"] >= out["iters"]).all())
    assert bool((out["iters"] >= 1).all()) and int(out["iters"].max()) < 3000 + m
    # idempotence: solving again from the minimiser performs no line search and leaves x untouched
    x2 = x.clone()
    out2 = NLO.LBFGS(NLO.DIAGQUAD, x2, d, b, workspace_=ws, Precision=prec, MaxIteration=3000)
    torch.cuda.synchronize()
    assert bool((out2["iters"] == 0).all()) and torch.equal(x2, x)
    # sampled subset against the oracle, bit for bit
    S = 48
    T, E = NLO.reduction_geometry(n)
    ref = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((S, n)), d=d[:S].cpu().numpy(), b=b[:S].cpu().numpy(),
                        opts=O.defaults(precision=prec, maxit=3000), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(x[:S].cpu().numpy(), ref["x"])
    assert np.array_equal(out["iters"][:S].cpu().numpy(), ref["iters"])


def test_config3_cg_batch_65536_properties():
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n, prec = 65536, 1024, 1e-5
    d, b = _inputs(B, n, dev)
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    out = NLO.ConjugateGradient(NLO.DIAGQUAD, x, d, b, Precision=prec, MaxIteration=3000)
    torch.cuda.synchronize()
    conv = out["status"] == NLO.CONVERGED
    assert float(conv.double().mean()) > 0.99
    gg = ((d * x - b) ** 2).sum(1)
    assert bool((gg[conv] < prec * prec * (1 + 1e-9)).all())
    xs = b / d
    assert float(((x - xs).norm(dim=1) / xs.norm(dim=1)).max()) < 1e-4


def test_edge_cases_empty_oversized_and_limits():
    import ctypes as C
    import FortranLibrary.NonlinearOptimization as NLO
    FL = NLO.FL
    dev = torch.device("cuda:0")
    o = NLO.default_options(NLO.LBFGS_)
    x = torch.zeros(1, 8, dtype=torch.float64, device=dev)
    # empty batch / n: refused, nothing launched
    assert FL.fl_lbfgs_batched(NLO.QUARTIC, 0, 8, x.data_ptr(), None, None, C.byref(o), None, 0, *([None] * 7)) == -1
    assert FL.fl_conjugate_gradient_batched(NLO.QUARTIC, 1, 0, x.data_ptr(), None, None, C.byref(o), *([None] * 7)) == -1
    # workspace too small
    ws = torch.empty(4, dtype=torch.float64, device=dev)
    assert FL.fl_lbfgs_batched(NLO.QUARTIC, 1, 8, x.data_ptr(), None, None, C.byref(o), ws.data_ptr(), 32, *([None] * 7)) == -3
    # Memory beyond FL_MAX_MEMORY = 64; Memory = 64 works and equals the oracle
    with pytest.raises(NLO.FLError):
        NLO.LBFGS(NLO.QUARTIC, x, Memory=65)
    rng = np.random.default_rng(2)
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, (2, 40))
    xg = torch.tensor(x0, device=dev)
    out = NLO.LBFGS(NLO.ROSENBROCK, xg, Memory=64, Precision=1e-10)
    torch.cuda.synchronize()
    T, E = NLO.reduction_geometry(40)
    ref = O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, opts=O.defaults(memory=64, precision=1e-10), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(xg.cpu().numpy(), ref["x"]) and np.array_equal(out["iters"].cpu().numpy(), ref["iters"])
    # Memory <= 0 is clamped to 1 like the reference (mem=max(1,Memory), NO.f90:419)
    xg = torch.tensor(x0, device=dev)
    out = NLO.LBFGS(NLO.ROSENBROCK, xg, Memory=0, MaxIteration=50)
    torch.cuda.synchronize()
    ref = O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, opts=O.defaults(memory=0, maxit=50), sum_mode=O.TREE, threads=T, ept=E)
    assert np.array_equal(xg.cpu().numpy(), ref["x"])
