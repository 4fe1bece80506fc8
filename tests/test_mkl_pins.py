"""The numerical-Hessian and TrustRegion branches of the reference go through Intel MKL (djacobi, dtrnlsp_*:
NO.f90:676, 981, 1067, 1258, 1782-1888).  MKL is closed, but its runtime ships in the build image, so
tools/make_mkl_golden.py drove the REAL routines (ctypes, no stand-in source) and committed what they returned:

  * tests/golden/mkl_djacobi.npz -- Jacobians of this repo's gradients by the real djacobi, plus (checked while
    generating) the step rule the real routine uses.  libFL.so's replacement fl_djacobi -- the routine behind every
    numerical-Hessian branch of the legacy entry points -- must return THE SAME BITS.  Host code: runs without a GPU.
  * tests/golden/mkl_trnlsp.npz  -- end points of the real dtrnlsp on the systems of tests/test_gpu_trust_region.py:
    held against the library's own Levenberg-Marquardt iteration in the -m gpu tests below.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_mkl_golden as G  # noqa: E402  (the generator's own test problems, pure numpy; MKL is only touched by its main())

dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
FCN = C.CFUNCTYPE(None, ip, ip, dp, dp)


def _lib():
    return C.CDLL(os.path.join(ROOT, "fortran-library_amd", "lib", "libFL.so"))


def _fl_djacobi(fun, x, m, eps=1e-8):
    n = len(x)
    calls = []

    def cb(pm, pn, px, pf):
        xx = np.ctypeslib.as_array(px, (n,)).copy()
        calls.append(xx)
        f = fun(xx)
        for i in range(m):
            pf[i] = f[i]
    J = np.zeros((m, n), order="F")
    xc = np.array(x, dtype=float)
    lib = _lib()
    lib.fl_djacobi.restype = C.c_int
    rc = lib.fl_djacobi(FCN(cb), C.byref(C.c_int(n)), C.byref(C.c_int(m)), J.ctypes.data_as(dp), xc.ctypes.data_as(dp),
                        C.byref(C.c_double(eps)))
    assert rc == 1501 and np.array_equal(xc, x)
    return J, calls


CASES = [("rosenbrock_std", G.grad_rosenbrock), ("rosenbrock_near", G.grad_rosenbrock), ("quartic", G.grad_quartic),
         ("quadratic", None), ("quartic_mixed_scales", G.grad_quartic)]


@pytest.mark.parametrize("n", [10, 64])
@pytest.mark.parametrize("name,fun", CASES)
def test_fl_djacobi_returns_the_bits_of_mkl_djacobi(name, fun, n):
    z = np.load(os.path.join(GOLD, "mkl_djacobi.npz"))
    x, Jmkl = z[f"{name}_n{n}_x"], z[f"{name}_n{n}_J"]
    fun = fun or G.make_grad_quadratic(n)
    J, calls = _fl_djacobi(fun, x, n)
    assert len(calls) == 2 * n  # 2n gradient evaluations per Hessian, like djacobi
    assert np.array_equal(np.ascontiguousarray(J).view(np.uint64), np.ascontiguousarray(Jmkl).view(np.uint64)), np.abs(J - Jmkl).max()


@pytest.mark.parametrize("n", [10, 64])
@pytest.mark.parametrize("name,fun", CASES)
def test_oracle_central_hessian_returns_the_bits_of_mkl_djacobi(name, fun, n):
    """the oracle's restatement of the djacobi call sites (oracle/fl_oracle.c: flo_central_hessian)"""
    import oracle_lib as O
    z = np.load(os.path.join(GOLD, "mkl_djacobi.npz"))
    x, Jmkl = z[f"{name}_n{n}_x"], z[f"{name}_n{n}_J"]
    fun = fun or G.make_grad_quadratic(n)
    FD = C.CFUNCTYPE(None, dp, dp, C.c_int, C.c_void_p)

    def cb(pg, px, nn, ctx):
        g = fun(np.ctypeslib.as_array(px, (n,)).copy())
        for i in range(n):
            pg[i] = g[i]
    H = np.zeros((n, n), order="F")
    xc = x.copy()
    O.lib().flo_central_hessian(FD(cb), H.ctypes.data_as(dp), xc.ctypes.data_as(dp), C.c_int(n), None, None)
    assert np.array_equal(xc, x)
    assert np.array_equal(np.ascontiguousarray(H).view(np.uint64), np.ascontiguousarray(Jmkl).view(np.uint64))


def test_fl_djacobi_rejects_bad_arguments():
    lib = _lib()
    lib.fl_djacobi.restype = C.c_int
    x = np.ones(3)
    J = np.zeros((3, 3), order="F")
    one = C.c_int(3)
    eps = C.c_double(1e-8)
    assert lib.fl_djacobi(None, C.byref(one), C.byref(one), J.ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(eps)) == 1502
    cb = FCN(lambda *a: None)
    assert lib.fl_djacobi(cb, C.byref(C.c_int(0)), C.byref(one), J.ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(eps)) == 1502
    assert lib.fl_djacobi(cb, C.byref(one), C.byref(one), J.ctypes.data_as(dp), x.ctypes.data_as(dp),
                          C.byref(C.c_double(0.0))) == 1502


# ------------------------------------------------------------------------------------------ TrustRegion vs dtrnlsp
def _res_jac_rosen(torch):
    def fun(x, rq):
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
    return fun


@pytest.mark.gpu
@pytest.mark.parametrize("n", [10, 12])
def test_trust_region_reaches_the_root_mkl_dtrnlsp_reaches(n):
    torch = pytest.importorskip("torch")
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
    import FortranLibrary.NonlinearOptimization as NLO
    z = np.load(os.path.join(GOLD, "mkl_trnlsp.npz"))
    xm = z[f"rosen_n{n}_x"]
    assert int(z[f"rosen_n{n}_stop"]) == 3 and float(z[f"rosen_n{n}_r_final"]) < 1e-10  # MKL: ||F|| < Precision
    x = torch.tensor(np.tile(z[f"rosen_n{n}_x0"], (4, 1)), device="cuda:0")
    out = NLO.TrustRegion(x, _res_jac_rosen(torch), 2 * (n - 1), Precision=1e-10)
    assert np.all(out["reason"].cpu().numpy() == 3)  # the same stopping criterion
    assert np.abs(x.cpu().numpy() - xm[None, :]).max() <= 1e-8
    assert float(out["resnorm"].max()) < 1e-10
    # the legacy one-problem symbols with host callbacks
    from FortranLibrary.basic import FL
    RES = C.CFUNCTYPE(None, dp, dp, ip, ip)
    JAC = C.CFUNCTYPE(C.c_int, dp, dp, ip, ip)
    m = 2 * (n - 1)

    def res(r, xx, M, N):
        v = G.rosen_residual(np.ctypeslib.as_array(xx, (n,)))
        for i in range(m):
            r[i] = v[i]

    def jac(J, xx, M, N):
        v = G.rosen_jacobian(np.ctypeslib.as_array(xx, (n,))).ravel(order="F")
        for i in range(m * n):
            J[i] = v[i]
        return 0
    for with_jac in (True, False):  # False: the Jacobian by central differences with djacobi's step rule (NO.f90:1779)
        xl = z[f"rosen_n{n}_x0"].copy()
        w, mi, ms = C.c_int32(0), C.c_int(1000), C.c_int(100)
        pr, mn = C.c_double(1e-10), C.c_double(1e-15)
        if with_jac:
            FL.__nonlinearoptimization_MOD_trustregion_basic(RES(res), JAC(jac), xl.ctypes.data_as(dp), C.byref(C.c_int(m)),
                                                             C.byref(C.c_int(n)), C.byref(w), C.byref(mi), C.byref(ms),
                                                             C.byref(pr), C.byref(mn))
        else:
            FL.__nonlinearoptimization_MOD_trustregion(RES(res), xl.ctypes.data_as(dp), C.byref(C.c_int(m)),
                                                       C.byref(C.c_int(n)), None, None, None, C.byref(w), C.byref(mi),
                                                       C.byref(ms), C.byref(pr), C.byref(mn))
        assert np.abs(xl - xm).max() <= 1e-8, with_jac


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(20, 30), (40, 60)])
def test_trust_region_reaches_the_least_squares_fit_mkl_dtrnlsp_reaches(n, m):
    """non-zero-residual fits: MKL stops on its criterion 6 with |J^T r| ~ 1e-7 |r| (fixture), i.e. its own minimiser is
    stationary to ~1e-7 only; the library's iteration is held to the same residual norm (second order in the distance
    of the minimisers: 1e-12 relative) and to MKL's minimiser within what MKL's own stationarity defines"""
    torch = pytest.importorskip("torch")
    sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
    import FortranLibrary.NonlinearOptimization as NLO
    z = np.load(os.path.join(GOLD, "mkl_trnlsp.npz"))
    Cm, t, xm = z[f"fit_n{n}_C"], z[f"fit_n{n}_t"], z[f"fit_n{n}_x"]
    dev = torch.device("cuda:0")
    Ct, tt = torch.tensor(Cm, device=dev), torch.tensor(t, device=dev)

    def fun(xx, rq):
        r = xx @ Ct.T - tt
        r[:, :n] += 0.05 * xx ** 3
        J = Ct.unsqueeze(0).repeat(xx.shape[0], 1, 1)
        J[:, torch.arange(n), torch.arange(n)] += 0.15 * xx ** 2
        return r, J.transpose(1, 2).contiguous()
    x = torch.zeros(3, n, dtype=torch.float64, device=dev)
    out = NLO.TrustRegion(x, fun, m, MaxIteration=100, Precision=1e-12, MinStepLength=1e-13)
    rn = out["resnorm"].cpu().numpy()
    assert np.all(np.abs(rn - float(z[f"fit_n{n}_r_final"])) <= 1e-12 * rn)
    # MKL's own stationarity at its answer, and the curvature of |r|^2 there, bound the distance of the two minimisers
    res, jac, _, _ = G.make_fit(n, m, {20: 3, 40: 4}[n])
    Jm = jac(xm)
    gm = Jm.T @ res(xm)
    lam_min = np.linalg.eigvalsh(Jm.T @ Jm).min()
    bound = 2.0 * np.linalg.norm(gm) / lam_min + 1e-9
    assert np.abs(x.cpu().numpy() - xm[None, :]).max() <= bound
    r, J = fun(x, None)
    grad = torch.einsum("bnm,bm->bn", J, r)
    assert float(grad.abs().max()) <= float(np.abs(gm).max())  # at least as stationary as MKL's answer
