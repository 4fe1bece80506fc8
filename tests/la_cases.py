"""Seeded inputs of the LinearAlgebra fixtures (tests/golden/la_ref.npz, written by tools/make_la_golden.py from the
reference's own LinearAlgebra.f90 + MKL) and the digest used for the n = 1024 results.  Inputs are pure functions of
the case, so the fixture holds outputs only; `input_digest()` is stored with them and checked by the tests, so a
numpy whose generator drew different numbers would be noticed rather than compared against the wrong answers."""
import zlib

import numpy as np

SPD_SIZES = (5, 64, 200, 1024)
NONSPD_SIZES = (5, 64)
SYM_SIZES = (5, 64, 200, 1024)
OUTER_SHAPES = ((7, 5), (64, 200), (1, 9))
TRI_SIZES = (5, 64)
GEMM_SHAPES = ((5, 7, 3), (64, 200, 33), (129, 65, 257), (1024, 1024, 1024))
EIG_SIZES = (5, 64, 200, 1024)
FULL_UP_TO = 200  # results up to this order are stored whole, larger ones as sampled rows + probe products


def rng(name):
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode()) + 20261004))


def spd_case(n):
    """A = G G^T / n + I (condition number of order 10), b ~ U(-1,1)"""
    r = rng(f"spd{n}")
    G = r.standard_normal((n, n))
    A = G @ G.T / n + np.eye(n)
    A = 0.5 * (A + A.T)
    return np.asfortranarray(A), r.uniform(-1, 1, n)


def nonspd_case(n):
    """SPD except for one negative direction: the Cholesky factorisation fails at a known leading minor"""
    A, b = spd_case(n)
    A = A.copy()
    k = (2 * n) // 3
    A[k, k] = -1.0
    return np.asfortranarray(A), b


def indefinite_case(n):
    """symmetric, both signs in the spectrum, well conditioned: Q diag(+-(1..3)) Q^T-like via a shifted random matrix"""
    r = rng(f"sym{n}")
    G = r.standard_normal((n, n))
    S = 0.5 * (G + G.T) / np.sqrt(n)
    d = np.where(np.arange(n) % 2 == 0, 3.0, -3.0)  # the shift keeps |eigenvalues| away from 0
    A = S + np.diag(d)
    return np.asfortranarray(A), r.uniform(-1, 1, n)


def outer_case(m, n):
    r = rng(f"outer{m}x{n}")
    return r.standard_normal(m), r.standard_normal(n)


def tri_case(n):
    return np.asfortranarray(rng(f"tri{n}").standard_normal((n, n)))


def gemm_case(m, k, n):
    r = rng(f"gemm{m}x{k}x{n}")
    return np.asfortranarray(r.standard_normal((m, k))), np.asfortranarray(r.standard_normal((k, n)))


def eig_case(n):
    r = rng(f"eig{n}")
    G = r.standard_normal((n, n))
    return np.asfortranarray(0.5 * (G + G.T))


def input_digest():
    acc = []
    for n in SPD_SIZES:
        A, b = spd_case(n)
        acc += [A.sum(), b.sum()]
    for n in SYM_SIZES:
        A, b = indefinite_case(n)
        acc += [A.sum(), b.sum()]
    for s in GEMM_SHAPES:
        A, B = gemm_case(*s)
        acc += [A.sum(), B.sum()]
    for n in EIG_SIZES:
        acc.append(eig_case(n).sum())
    return np.array(acc)


# ---- digest of a large result: sampled rows and columns, and products with seeded probe vectors
def _samples(rows):
    idx = sorted({0, 1, 2, rows // 3, rows // 2, rows // 2 + 1, (2 * rows) // 3, rows - 2, rows - 1})
    return np.array([i for i in idx if 0 <= i < rows])


def _probes(cols):
    return rng(f"probe{cols}").standard_normal((cols, 4))


def store_matrix(put, key, M, order):
    if order <= FULL_UP_TO:
        put(key, M)
    else:
        put(key + "_rows", M[_samples(M.shape[0]), :])
        put(key + "_cols", M[:, _samples(M.shape[1])])
        put(key + "_probe", M @ _probes(M.shape[1]))


def compare_matrix(fix, key, M, order, rtol, scale=None):
    """max |M - reference| / scale over what the fixture holds for `key`; asserts <= rtol"""
    def err(ref, got):
        s = scale if scale is not None else max(1.0, float(np.abs(ref).max()))
        return float(np.abs(got - ref).max()) / s
    if order <= FULL_UP_TO:
        e = err(fix[key], M)
    else:
        e = max(err(fix[key + "_rows"], M[_samples(M.shape[0]), :]),
                err(fix[key + "_cols"], M[:, _samples(M.shape[1])]),
                err(fix[key + "_probe"], M @ _probes(M.shape[1])) / np.sqrt(M.shape[1]))
    assert e <= rtol, f"{key}: error {e:.3e} > {rtol:.1e}"
    return e
