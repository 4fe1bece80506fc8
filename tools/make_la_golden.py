#!/usr/bin/env python3
"""Writes tests/golden/la_ref.npz: outputs of the REFERENCE's LinearAlgebra routines (the unmodified
/root/reference/source/LinearAlgebra.f90 compiled by oracle/build_ref.sh against the image's MKL runtime) on seeded
inputs.  The fixture pins the oracle's dense restatements and the GPU dense kernels to the reference itself:

    My_dpotri LA.f90:798-812   My_dposv 719-730   My_dsysv 695-703   vector_direct_product 105-114
    sycp 241-249   dsyL2U 260-265   My_dgemm 182-188   My_dgemm_T 190-196   My_dsyev 879-887

Inputs are functions of (case name, seed) -- tests/la_cases.py regenerates them -- so only outputs are stored, and
for n = 1024 only sampled rows plus products with seeded probe vectors (a digest of the n x n result that a wrong
result cannot match).  Run in the build container (needs /root/reference and oracle/_ref):

    bash oracle/build_ref.sh && python tools/make_la_golden.py
"""
import ctypes as C
import os
import sys

os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
os.environ.setdefault("MKL_INTERFACE_LAYER", "LP64")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import la_cases as LC  # noqa: E402


def ref_lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfl_ref_la.so"))
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.ref_my_dpotri.argtypes = [dp, C.c_int, ip]
    lib.ref_my_dposv.argtypes = [dp, dp, C.c_int, ip]
    lib.ref_my_dsysv.argtypes = [dp, dp, C.c_int]
    lib.ref_vector_direct_product.argtypes = [dp, dp, dp, C.c_int, C.c_int]
    lib.ref_sycp.argtypes = [dp, dp, C.c_int]
    lib.ref_dsyl2u.argtypes = [dp, C.c_int]
    lib.ref_my_dgemm.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int]
    lib.ref_my_dgemm_t.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int]
    lib.ref_my_dsyev.argtypes = [C.c_char, dp, dp, C.c_int]
    for f in (lib.ref_my_dpotri, lib.ref_my_dposv, lib.ref_my_dsysv, lib.ref_vector_direct_product, lib.ref_sycp,
              lib.ref_dsyl2u, lib.ref_my_dgemm, lib.ref_my_dgemm_t, lib.ref_my_dsyev):
        f.restype = None
    return lib


def P(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def F(a):  # a fresh column-major working copy
    return np.array(a, dtype=np.float64, order="F", copy=True)


def main():
    lib = ref_lib()
    out = {}

    def put(key, val):
        out[key] = np.asarray(val)

    # ---- SPD: My_dpotri, My_dposv
    for n in LC.SPD_SIZES:
        A, b = LC.spd_case(n)
        W = F(A)
        info = C.c_int(-7)
        lib.ref_my_dpotri(P(W), n, C.byref(info))
        put(f"dpotri_info_{n}", info.value)
        LC.store_matrix(put, f"dpotri_{n}", np.tril(W), n)  # only the lower triangle is defined by dpotri 'L'
        W = F(A)
        x = b.copy()
        lib.ref_my_dposv(P(W), P(x), n, C.byref(info))
        put(f"dposv_info_{n}", info.value)
        put(f"dposv_x_{n}", x)
        LC.store_matrix(put, f"dposv_L_{n}", np.tril(W), n)
    # not positive definite: info = index of the failing leading minor, b untouched, A partially factorised
    for n in LC.NONSPD_SIZES:
        A, b = LC.nonspd_case(n)
        W = F(A)
        info = C.c_int(-7)
        lib.ref_my_dpotri(P(W), n, C.byref(info))
        put(f"nonspd_dpotri_info_{n}", info.value)
        W = F(A)
        x = b.copy()
        lib.ref_my_dposv(P(W), P(x), n, C.byref(info))
        put(f"nonspd_dposv_info_{n}", info.value)
        put(f"nonspd_dposv_x_{n}", x)
    # ---- symmetric indefinite: My_dsysv
    for n in LC.SYM_SIZES:
        A, b = LC.indefinite_case(n)
        W = F(A)
        x = b.copy()
        lib.ref_my_dsysv(P(W), P(x), n)
        put(f"dsysv_x_{n}", x)
    # ---- outer product, triangle copy / mirror
    for (m, n) in LC.OUTER_SHAPES:
        a, b = LC.outer_case(m, n)
        Cm = np.zeros((m, n), order="F")
        lib.ref_vector_direct_product(P(a), P(b), P(Cm), m, n)
        put(f"outer_{m}x{n}", Cm)
    for n in LC.TRI_SIZES:
        B = LC.tri_case(n)
        A = F(np.full((n, n), -1.0))
        lib.ref_sycp(P(A), P(F(B)), n)
        put(f"sycp_{n}", A)
        W = F(B)
        lib.ref_dsyl2u(P(W), n)
        put(f"syl2u_{n}", W)
    # ---- My_dgemm, My_dgemm_T
    for (m, k, n) in LC.GEMM_SHAPES:
        A, B = LC.gemm_case(m, k, n)
        Cm = np.zeros((m, n), order="F")
        lib.ref_my_dgemm(P(F(A)), P(F(B)), P(Cm), m, k, n)
        LC.store_matrix(put, f"dgemm_{m}x{k}x{n}", Cm, max(m, n))
        At = F(A.T)  # K x M
        Cm = np.zeros((m, n), order="F")
        lib.ref_my_dgemm_t(P(At), P(F(B)), P(Cm), m, k, n)
        LC.store_matrix(put, f"dgemmT_{m}x{k}x{n}", Cm, max(m, n))
    # ---- My_dsyev
    for n in LC.EIG_SIZES:
        A = LC.eig_case(n)
        for job in (b"N", b"V"):
            W = F(np.tril(A))  # only the lower triangle is referenced
            ev = np.zeros(n)
            lib.ref_my_dsyev(job, P(W), P(ev), n)
            put(f"dsyev_{job.decode()}_{n}", ev)
            if job == b"V":
                res = np.abs(A @ W - W * ev[None, :]).max()
                orth = np.abs(W.T @ W - np.eye(n)).max()
                put(f"dsyev_V_resid_{n}", res)
                put(f"dsyev_V_orth_{n}", orth)
    put("input_digest", LC.input_digest())
    path = os.path.join(ROOT, "tests", "golden", "la_ref.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
