"""ctypes view of the CPU oracle (oracle/libfl_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")

SEQ, TREE = 0, 1
QUARTIC, ROSENBROCK, DIAGQUAD = 0, 1, 2
SD, CG, LBFGS, BFGS = 0, 1, 2, 3
CONVERGED, STEP_CONVERGED, MAXIT = 0, 1, 2


class Opts(C.Structure):
    _fields_ = [("strong", C.c_int), ("maxit", C.c_int), ("precision", C.c_double), ("minstep", C.c_double),
                ("c1", C.c_double), ("c2", C.c_double), ("increment", C.c_double), ("memory", C.c_int),
                ("exact_step", C.c_int), ("method", C.c_int), ("clamp", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ODIR, "libfl_oracle.so")
        srcs = [os.path.join(ODIR, f) for f in ("fl_oracle.c", "fl_oracle_problems.c", "fl_oracle.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
            subprocess.check_call(["make", "-C", ODIR, "-s"])
        _lib = C.CDLL(so)
        _lib.flo_defaults.argtypes = [C.POINTER(Opts)]
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        _lib.flo_solve_batch.restype = C.c_int
        _lib.flo_solve_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, C.POINTER(Opts), C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, ip, ip, ip, dp]
        _lib.flo_auglag_batch.restype = C.c_int
        _lib.flo_auglag_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double,
                                          C.POINTER(Opts), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, ip, ip,
                                          ip, ip, dp]
        _lib.flo_dot.restype = C.c_double
        _lib.flo_dot.argtypes = [C.c_int, dp, dp]
        _lib.flo_set_sum_mode.argtypes = [C.c_int, C.c_int, C.c_int]
        _lib.flo_dpotri_lower.restype = C.c_int
        _lib.flo_dpotri_lower.argtypes = [dp, C.c_int]
    return _lib


def defaults(**kw):
    o = Opts()
    lib().flo_defaults(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def solve_batch(solver, kind, x0, d=None, b=None, opts=None, use_ffd=False, bfgs_form=0, sum_mode=SEQ, threads=64,
                ept=2, nthreads=0):
    """Run the oracle on a batch x0[B, n]; returns dict of numpy arrays."""
    x = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64).copy()
    B, n = x.shape
    if d is not None:
        d = np.ascontiguousarray(np.broadcast_to(d, (B, n)), dtype=np.float64)
        b = np.ascontiguousarray(np.broadcast_to(b, (B, n)), dtype=np.float64)
    o = opts if opts is not None else defaults()
    f = np.zeros(B)
    gg = np.zeros(B)
    it = np.zeros(B, dtype=np.int32)
    st = np.zeros(B, dtype=np.int32)
    nf = np.zeros(B, dtype=np.int32)
    ng = np.zeros(B, dtype=np.int32)
    used = lib().flo_solve_batch(solver, kind, B, n, _dp(x), _dp(d), _dp(b), C.byref(o), int(use_ffd), bfgs_form,
                                 sum_mode, threads, ept, nthreads, _dp(f), _ip(it), _ip(st), _ip(nf), _ip(ng),
                                 _dp(gg))
    return dict(x=x, f=f, iters=it, status=st, nf=nf, ng=ng, gg=gg, threads=used)


def auglag_batch(solver, kind, x0, m, d=None, b=None, lambda0=None, miu0=1.0, opts=None, use_ffd=False,
                 sum_mode=SEQ, threads=64, ept=2, nthreads=0):
    x = np.ascontiguousarray(np.atleast_2d(x0), dtype=np.float64).copy()
    B, n = x.shape
    if d is not None:
        d = np.ascontiguousarray(np.broadcast_to(d, (B, n)), dtype=np.float64)
        b = np.ascontiguousarray(np.broadcast_to(b, (B, n)), dtype=np.float64)
    lam = np.zeros((B, m)) if lambda0 is None else np.ascontiguousarray(np.broadcast_to(lambda0, (B, m)),
                                                                      dtype=np.float64).copy()
    o = opts if opts is not None else defaults()
    f = np.zeros(B)
    cc = np.zeros(B)
    it = np.zeros(B, dtype=np.int32)
    outer = np.zeros(B, dtype=np.int32)
    nf = np.zeros(B, dtype=np.int32)
    ng = np.zeros(B, dtype=np.int32)
    used = lib().flo_auglag_batch(solver, kind, B, n, m, _dp(x), _dp(d), _dp(b), _dp(lam), miu0, C.byref(o),
                                  int(use_ffd), sum_mode, threads, ept, nthreads, _dp(f), _ip(it), _ip(outer),
                                  _ip(nf), _ip(ng), _dp(cc))
    if used < 0:
        raise ValueError("flo_auglag_batch refused its arguments")
    return dict(x=x, f=f, iters=it, outer=outer, nf=nf, ng=ng, cnorm2=cc, lam=lam, threads=used)
