"""-m gpu: TrustRegion (NO.f90:1728-1906) for a BATCH on the device by reverse communication (fl_trust_region_*):
own Levenberg-Marquardt behind the reference's interface, parity unpinned by construction (MKL's dtrnlsp is closed), so
the tests pin solutions: known roots, agreement with the one-problem legacy symbol (the same iteration on the host),
bounds honoured, stopping reasons."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _rosen_res_jac(x):
    """chained Rosenbrock residuals r = (10 (x_{i+1} - x_i^2), 1 - x_i), M = 2 (N - 1); J [B, N, M] column-major M x N"""
    B, n = x.shape
    m = 2 * (n - 1)
    r = torch.empty(B, m, dtype=torch.float64, device=x.device)
    r[:, 0::2] = 10.0 * (x[:, 1:] - x[:, :-1] ** 2)
    r[:, 1::2] = 1.0 - x[:, :-1]
    J = torch.zeros(B, n, m, dtype=torch.float64, device=x.device)
    i = torch.arange(n - 1, device=x.device)
    J[:, i, 2 * i] = -20.0 * x[:, :-1]
    J[:, i + 1, 2 * i] = 10.0
    J[:, i, 2 * i + 1] = -1.0
    return r, J


def test_batched_trust_region_solves_chained_rosenbrock_systems():
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B, n = 64, 12
    rng = np.random.default_rng(5)
    x0 = np.tile(np.where(np.arange(n) % 2 == 0, -1.2, 1.0), (B, 1)) + 0.05 * rng.standard_normal((B, n))
    x = torch.tensor(x0, device=dev)
    out = NLO.TrustRegion(x, lambda xx, rq: _rosen_res_jac(xx), 2 * (n - 1), MaxIteration=200, MaxStepIteration=50,
                          Precision=1e-10)
    assert np.all(out["reason"].cpu().numpy() == 3)  # ||f'(x)|| < Precision
    assert float((x - 1.0).abs().max()) < 1e-8
    assert float(out["resnorm"].max()) < 1e-10
    assert int(out["iters"].max()) < 200


def test_batched_trust_region_agrees_with_the_one_problem_legacy_symbol():
    """problem 0 of the batch through __nonlinearoptimization_MOD_trustregion_basic with host callbacks: the same
    Levenberg-Marquardt iteration (host loop, same DGEMM / Cholesky kernels) must land on the same point"""
    NLO = _nlo()
    from FortranLibrary.basic import FL
    dev = torch.device("cuda:0")
    n = 10
    m = 2 * (n - 1)
    x0 = np.where(np.arange(n) % 2 == 0, -1.2, 1.0)
    xb = torch.tensor(np.tile(x0, (3, 1)), device=dev)
    out = NLO.TrustRegion(xb, lambda xx, rq: _rosen_res_jac(xx), m, MaxIteration=200, MaxStepIteration=50, Precision=1e-10)
    RES = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))
    JAC = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int))

    def res(r, xx, M, N):
        for i in range(n - 1):
            r[2 * i] = 10.0 * (xx[i + 1] - xx[i] * xx[i])
            r[2 * i + 1] = 1.0 - xx[i]

    def jac(J, xx, M, N):
        for k in range(m * n):
            J[k] = 0.0
        for i in range(n - 1):
            J[i * m + 2 * i] = -20.0 * xx[i]
            J[(i + 1) * m + 2 * i] = 10.0
            J[i * m + 2 * i + 1] = -1.0
        return 0
    xl = x0.copy()
    dp = C.POINTER(C.c_double)
    w, mi, ms = C.c_int32(0), C.c_int(200), C.c_int(50)
    pr, mn = C.c_double(1e-10), C.c_double(1e-15)
    FL.__nonlinearoptimization_MOD_trustregion_basic(RES(res), JAC(jac), xl.ctypes.data_as(dp), C.byref(C.c_int(m)),
                                                     C.byref(C.c_int(n)), C.byref(w), C.byref(mi), C.byref(ms), C.byref(pr),
                                                     C.byref(mn))
    got = xb.cpu().numpy()
    assert np.abs(got[0] - xl).max() < 1e-9 and np.abs(got - got[0]).max() == 0.0  # identical problems, identical answers


def test_batched_trust_region_honours_bounds_and_reports_reasons():
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B, n = 8, 12
    x0 = np.tile(np.where(np.arange(n) % 2 == 0, -1.2, 0.3), (B, 1))
    x = torch.tensor(x0, device=dev)
    low = torch.full((n,), -2.0, dtype=torch.float64, device=dev)
    up = torch.full((n,), 0.5, dtype=torch.float64, device=dev)
    r0 = _rosen_res_jac(x)[0].pow(2).sum(1)
    out = NLO.TrustRegion(x, lambda xx, rq: _rosen_res_jac(xx), 2 * (n - 1), low=low, up=up, MaxIteration=200,
                          MaxStepIteration=50, Precision=1e-10)
    xs = x.cpu().numpy()
    assert np.all(xs <= 0.5 + 1e-15) and np.all(xs >= -2.0)  # the root x = 1 is outside the box
    assert np.all(out["reason"].cpu().numpy() != 3)
    assert float((_rosen_res_jac(x)[0].pow(2).sum(1) / r0).max()) < 0.5
    # a budget of one iteration: reason 1 (MaxIteration) and exactly one accepted step
    x = torch.tensor(x0, device=dev)
    out = NLO.TrustRegion(x, lambda xx, rq: _rosen_res_jac(xx), 2 * (n - 1), MaxIteration=1, Precision=1e-10)
    assert np.all(out["reason"].cpu().numpy() == 1) and np.all(out["iters"].cpu().numpy() == 1)


def test_batched_trust_region_larger_overdetermined_systems():
    """N = 600 (blocked Cholesky path of fl_dposv_batched), M = 900: r(x) = C x - t + 0.05 x_head^3, a smooth system with
    a unique least-squares solution near the linear one; gradient J^T r must vanish at the answer"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B, n, m = 3, 600, 900
    g = torch.Generator(device="cpu").manual_seed(3)
    Cm = (torch.randn(B, m, n, generator=g, dtype=torch.float64) / np.sqrt(n)).to(dev)
    Cm[:, :n, :] += torch.eye(n, dtype=torch.float64, device=dev)
    t = torch.randn(B, m, generator=g, dtype=torch.float64).to(dev)

    def fun(xx, rq):
        r = torch.einsum("bmn,bn->bm", Cm, xx) - t
        r[:, :n] += 0.05 * xx ** 3
        J = Cm.clone()                                   # [B, m, n] = dr_i/dx_j
        J[:, torch.arange(n), torch.arange(n)] += 0.15 * xx ** 2
        return r, J.transpose(1, 2).contiguous()          # [B, n, m]: column-major M x N
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    out = NLO.TrustRegion(x, fun, m, MaxIteration=100, Precision=1e-12, MinStepLength=1e-13)
    r, J = fun(x, None)
    grad = torch.einsum("bnm,bm->bn", J, r)
    # a non-zero-residual fit: first-order stationarity.  The bar is the dependency's own: on this family the real MKL
    # dtrnlsp stops (criterion 6) at |J^T r| = 1.3e-7 |r| (tests/golden/mkl_trnlsp.npz: fit_n20, fit_n40); the
    # library's iteration is held to better than that here and to MKL's end points in tests/test_mkl_pins.py
    assert float(grad.abs().max()) < 1e-7 * float(r.norm(dim=1).max())
    assert np.all(np.isin(out["reason"].cpu().numpy(), (4, 5, 3, 2)))


def test_batched_trust_region_ignores_arrays_it_did_not_ask_for():
    """A rejected or retried step must be rebuilt from J^T J, J^T r of the problem's CURRENT point, whatever the
    caller's arrays hold meanwhile: a caller that poisons r / J of every problem whose request bits did not ask for them
    gets bit for bit the answer of one that evaluates everything everywhere -- with rejections on the way (far start,
    tiny first damping)."""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    B, n = 16, 12
    rng = np.random.default_rng(9)
    x0 = np.tile(np.where(np.arange(n) % 2 == 0, -3.0, 4.0), (B, 1)) + 0.3 * rng.standard_normal((B, n))
    seen = {"again": 0, "r_only": 0}

    def poisoned(xx, rq):
        r, J = _rosen_res_jac(xx)
        r[(rq & 1) == 0] = float("nan")
        J[(rq & 2) == 0] = float("nan")
        seen["again"] += int(((rq & 4) != 0).sum())
        seen["r_only"] += int(((rq & 3) == 1).sum())
        return r, J
    xa = torch.tensor(x0, device=dev)
    oa = NLO.TrustRegion(xa, lambda xx, rq: _rosen_res_jac(xx), 2 * (n - 1), MaxIteration=300, MaxStepIteration=60, Precision=1e-10,
                         check_every=1)
    xb = torch.tensor(x0, device=dev)
    ob = NLO.TrustRegion(xb, poisoned, 2 * (n - 1), MaxIteration=300, MaxStepIteration=60, Precision=1e-10, check_every=1)
    assert seen["r_only"] > int(ob["iters"].sum())  # more trial points than accepted steps: some trials were rejected
    assert torch.equal(xa, xb) and torch.equal(oa["iters"], ob["iters"]) and torch.equal(oa["reason"], ob["reason"])
    assert np.all(ob["reason"].cpu().numpy() == 3) and float((xb - 1.0).abs().max()) < 1e-8
