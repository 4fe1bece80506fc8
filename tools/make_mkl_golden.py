#!/usr/bin/env python3
"""Golden vectors from the REAL third-party dependency of the reference's numerical-Hessian and TrustRegion branches:
Intel MKL's `djacobi` and `dtrnlsp_*` (NO.f90:676, 981, 1067, 1258; 1782-1888), called through ctypes on the MKL
runtime that ships in the build image (/opt/conda/lib/libmkl_rt.so, MKL 2021.4).  Build container only -- the GPU
box has no MKL and needs none: the outputs are committed as tests/golden/mkl_djacobi.npz and mkl_trnlsp.npz.  Nothing
of /root/reference is read, no stand-in source is written: the callbacks below are this repo's own test problems.

What it pins:
  * djacobi(fcn, n, m, fjac, x, eps = 1e-8) -- the central-difference Jacobian the reference takes of f' when no fdd is
    passed.  The script also RECORDS THE POINTS at which MKL calls fcn and checks the rule they reveal, bit for bit:
        |x_j| >  eps:  fcn at x_j (1 + eps) and x_j (1 - eps),   h = eps * x_j   (signed)
        |x_j| <= eps:  fcn at x_j + eps     and x_j - eps,       h = eps
        fjac(:, j) = (f_plus - f_minus) * (0.5 / h)
    libFL.so's host central differences (csrc/fl_rci.hip: central_difference_hessian) restate exactly that, so the
    numerical-Hessian branches are pinned to the dependency the reference calls (tests/test_mkl_pins.py).
  * dtrnlsp_init / _check / _solve / _get / _delete with the reference's settings (tol = 1e-15 except tol(2) =
    Precision, StepBound = 100, analytic Jacobian): end points, iteration counts, stop criteria and residual norms on
    the systems of tests/test_gpu_trust_region.py -- the end-point oracle for the library's own Levenberg-Marquardt
    iteration behind the TrustRegion interface.

usage: python tools/make_mkl_golden.py            (writes both files; prints a summary)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MKL = os.environ.get("FL_MKL_RT", "/opt/conda/lib/libmkl_rt.so")
TR_SUCCESS = 1501
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
FCN = C.CFUNCTYPE(None, ip, ip, dp, dp)  # subroutine fcn(m, n, x, f), everything by reference


# ---------------------------------------------------------------- the repo's test objectives (gradients)
def grad_rosenbrock(x):  # chained Rosenbrock, the objective of BASELINE configs 1 and 2
    n = len(x)
    g = np.zeros(n)
    for i in range(n - 1):
        u = x[i + 1] - x[i] * x[i]
        g[i] += -400.0 * x[i] * u - 2.0 * (1.0 - x[i])
        g[i + 1] += 200.0 * u
    return g


def grad_quartic(x):  # sum x^4, the objective of the reference's own tests
    return 4.0 * x ** 3


def make_grad_quadratic(n):  # diagonal quadratic of BASELINE config 3: d_i = 1 + 99 (i-1)/(n-1), b = sin(i)
    i = np.arange(1, n + 1, dtype=float)
    d = 1.0 + 99.0 * (i - 1.0) / max(1, n - 1)
    b = np.sin(i)
    return lambda x: d * x - b


def djacobi(mkl, fun, x, m, eps=1e-8):
    """(fjac [m, n] column-major, list of (point, value) of every fcn call)"""
    n = len(x)
    calls = []

    def cb(pm, pn, px, pf):
        xx = np.ctypeslib.as_array(px, (n,)).copy()
        f = np.asarray(fun(xx), dtype=float)
        calls.append((xx, f.copy()))
        for i in range(m):
            pf[i] = f[i]
    J = np.zeros((m, n), order="F")
    xc = np.array(x, dtype=float)
    rc = mkl.djacobi(FCN(cb), C.byref(C.c_int(n)), C.byref(C.c_int(m)), J.ctypes.data_as(dp), xc.ctypes.data_as(dp),
                     C.byref(C.c_double(eps)))
    assert rc == TR_SUCCESS, rc
    assert np.array_equal(xc, x), "djacobi restores x"
    return J, calls


def check_rule(x, J, calls, eps=1e-8):
    """the step rule stated in the module docstring, bit for bit"""
    n = len(x)
    for j in range(n):
        cj = [(xx, f) for xx, f in calls if xx[j] != x[j]]
        assert len(cj) == 2 and all(np.array_equal(np.delete(xx, j), np.delete(x, j)) for xx, _ in cj), j
        (xa, fa), (xb, fb) = cj
        if abs(x[j]) > eps:
            h, pa, pb = eps * x[j], x[j] * (1.0 + eps), x[j] * (1.0 - eps)
        else:
            h, pa, pb = eps, x[j] + eps, x[j] - eps
        assert xa[j] == pa and xb[j] == pb, (j, x[j], xa[j], xb[j])
        assert np.array_equal((fa - fb) * (0.5 / h), J[:, j]), j


def make_djacobi(mkl):
    out = {}
    rng = np.random.default_rng(20261004)
    for n in (10, 64):
        i = np.arange(1, n + 1, dtype=float)
        cases = {
            "rosenbrock_std": (grad_rosenbrock, np.where(np.arange(n) % 2 == 0, -1.2, 1.0)),
            "rosenbrock_near": (grad_rosenbrock, 1.0 + 0.1 * np.sin(i)),
            "quartic": (grad_quartic, 0.1 * i),
            "quadratic": (make_grad_quadratic(n), rng.standard_normal(n)),
            # tiny, zero and huge coordinates: both branches of the step rule
            "quartic_mixed_scales": (grad_quartic, rng.standard_normal(n) * 10.0 ** rng.integers(-12, 4, n) * (rng.random(n) > 0.15)),
        }
        for name, (fun, x) in cases.items():
            x = np.ascontiguousarray(x, dtype=float)
            J, calls = djacobi(mkl, fun, x, n)
            check_rule(x, J, calls)
            out[f"{name}_n{n}_x"] = x
            out[f"{name}_n{n}_J"] = np.ascontiguousarray(J)  # [m, n]: J[i, j] = d f'_i / d x_j
    out["eps"] = np.array(1e-8)
    return out


# ---------------------------------------------------------------- dtrnlsp
def rosen_residual(x):
    n = len(x)
    r = np.empty(2 * (n - 1))
    r[0::2] = 10.0 * (x[1:] - x[:-1] ** 2)
    r[1::2] = 1.0 - x[:-1]
    return r


def rosen_jacobian(x):
    n = len(x)
    m = 2 * (n - 1)
    J = np.zeros((m, n), order="F")
    for i in range(n - 1):
        J[2 * i, i] = -20.0 * x[i]
        J[2 * i, i + 1] = 10.0
        J[2 * i + 1, i] = -1.0
    return J


def make_fit(n, m, seed):
    """r(x) = C x - t + 0.05 x_head^3 (the over-determined family of tests/test_gpu_trust_region.py, smaller)"""
    rng = np.random.default_rng(seed)
    Cm = rng.standard_normal((m, n)) / np.sqrt(n)
    Cm[:n, :] += np.eye(n)
    t = rng.standard_normal(m)

    def res(x):
        r = Cm @ x - t
        r[:n] += 0.05 * x ** 3
        return r

    def jac(x):
        J = np.array(Cm, order="F")
        J[np.arange(n), np.arange(n)] += 0.15 * x ** 2
        return J
    return res, jac, Cm, t


def trnlsp(mkl, res, jac, x0, m, precision=1e-10, minstep=None, maxit=1000, maxstepit=100):
    """MKL's RCI trust-region solver with the reference's settings (NO.f90:1764-1775, 1784, 1838-1888)"""
    n = len(x0)
    x = np.array(x0, dtype=float)
    tol = np.full(6, 1e-15)
    tol[1] = precision
    if minstep is not None:
        tol[0] = tol[3] = tol[4] = minstep
    handle = C.c_void_p()
    ni, mi = C.c_int(n), C.c_int(m)
    fvec = np.ascontiguousarray(res(x))
    fjac = np.asfortranarray(jac(x))
    rc = mkl.dtrnlsp_init(C.byref(handle), C.byref(ni), C.byref(mi), x.ctypes.data_as(dp), tol.ctypes.data_as(dp),
                          C.byref(C.c_int(maxit)), C.byref(C.c_int(maxstepit)), C.byref(C.c_double(100.0)))
    assert rc == TR_SUCCESS, rc
    info = (C.c_int * 6)()
    rc = mkl.dtrnlsp_check(C.byref(handle), C.byref(ni), C.byref(mi), fjac.ctypes.data_as(dp), fvec.ctypes.data_as(dp),
                           tol.ctypes.data_as(dp), info)
    assert rc == TR_SUCCESS and not any(info[:4]), (rc, list(info))
    rq = C.c_int(0)
    nres = njac = 0
    while True:
        rc = mkl.dtrnlsp_solve(C.byref(handle), fvec.ctypes.data_as(dp), fjac.ctypes.data_as(dp), C.byref(rq))
        assert rc == TR_SUCCESS, rc
        if rq.value in (-1, -2, -3, -4, -5, -6):
            break
        if rq.value == 1:
            fvec[:] = res(x)
            nres += 1
        elif rq.value == 2:
            fjac[:, :] = jac(x)
            njac += 1
    it, st = C.c_int(), C.c_int()
    r1, r2 = C.c_double(), C.c_double()
    rc = mkl.dtrnlsp_get(C.byref(handle), C.byref(it), C.byref(st), C.byref(r1), C.byref(r2))
    assert rc == TR_SUCCESS
    mkl.dtrnlsp_delete(C.byref(handle))
    return dict(x=x, iterations=it.value, stop=st.value, r_initial=r1.value, r_final=r2.value, nres=nres, njac=njac,
                rci_exit=rq.value)


def make_trnlsp(mkl):
    out = {}
    for n in (10, 12):
        x0 = np.where(np.arange(n) % 2 == 0, -1.2, 1.0)
        r = trnlsp(mkl, rosen_residual, rosen_jacobian, x0, 2 * (n - 1))
        for k, v in r.items():
            out[f"rosen_n{n}_{k}"] = np.asarray(v)
        out[f"rosen_n{n}_x0"] = x0
        print(f"dtrnlsp chained Rosenbrock n={n}: stop {r['stop']} after {r['iterations']} iterations, "
              f"|r| {r['r_initial']:.3e} -> {r['r_final']:.3e}, max|x-1| {np.abs(r['x'] - 1).max():.2e}")
    for n, m, seed in ((20, 30, 3), (40, 60, 4)):
        res, jac, Cm, t = make_fit(n, m, seed)
        r = trnlsp(mkl, res, jac, np.zeros(n), m, precision=1e-12, minstep=1e-13, maxit=100)
        g = jac(r["x"]).T @ res(r["x"])
        for k, v in r.items():
            out[f"fit_n{n}_{k}"] = np.asarray(v)
        out[f"fit_n{n}_C"], out[f"fit_n{n}_t"] = Cm, t
        print(f"dtrnlsp fit n={n} m={m}: stop {r['stop']} after {r['iterations']} iterations, |r| {r['r_initial']:.3e} -> "
              f"{r['r_final']:.6e}, |J^T r| {np.abs(g).max():.2e}")
    return out


def main():
    if not os.path.exists(MKL):
        sys.exit(f"{MKL} not found: this script runs in the build image only")
    mkl = C.CDLL(MKL)
    for f in ("djacobi", "dtrnlsp_init", "dtrnlsp_check", "dtrnlsp_solve", "dtrnlsp_get", "dtrnlsp_delete"):
        getattr(mkl, f).restype = C.c_int
    gold = os.path.join(ROOT, "tests", "golden")
    dj = make_djacobi(mkl)
    np.savez_compressed(os.path.join(gold, "mkl_djacobi.npz"), **dj)
    print("mkl_djacobi.npz:", len([k for k in dj if k.endswith("_J")]), "Jacobians; step rule verified bit for bit on every column")
    tr = make_trnlsp(mkl)
    np.savez_compressed(os.path.join(gold, "mkl_trnlsp.npz"), **tr)
    print("mkl_trnlsp.npz:", sorted({k.rsplit('_', 1)[0] for k in tr})[:8], "...")


if __name__ == "__main__":
    main()
