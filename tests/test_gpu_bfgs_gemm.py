"""The as-written BFGS update (two dense matmuls, NO.f90:958-962) on the f64 matrix cores against the
oracle's restatement of the same two matmuls (sequential sums) and against the rank-2 form.
MFMA accumulates with fused multiply-adds and its own k order, so this is a tolerance test (relative to
the magnitude of the products), on ASYMMETRIC data so that a transposed fragment map cannot pass."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("n", [10, 64, 130, 257, 600])
def test_two_gemm_update_matches_reference_matmuls(n):
    import FortranLibrary.NonlinearOptimization as NLO
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_bfgs_update.argtypes = [C.c_int, dp, dp, dp, C.c_int]
    lib.flo_set_sum_mode(O.SEQ, 64, 2)
    rng = np.random.default_rng(n)
    B = 3
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    Hs = rng.standard_normal((B, n, n))  # asymmetric on purpose; column-major: Hs[b, col, row]
    s = rng.standard_normal((B, n))
    y = s * rng.uniform(0.5, 2.0, (B, n)) + 0.1 * rng.standard_normal((B, n))
    Hpad = np.zeros((B, n, ld))
    Hpad[:, :, :n] = Hs
    dev = torch.device("cuda:0")
    Hd = torch.tensor(Hpad, device=dev)
    NLO.bfgs_update_gemm(Hd, torch.tensor(s, device=dev), torch.tensor(y, device=dev), chunk=2)
    torch.cuda.synchronize()
    got = Hd.cpu().numpy()[:, :, :n]
    assert np.all(Hd.cpu().numpy()[:, :, n:] == 0.0)  # padding untouched
    for k in range(B):
        ref0 = np.ascontiguousarray(Hs[k]).copy()
        lib.flo_bfgs_update(n, ref0.ctypes.data_as(dp), s[k].ctypes.data_as(dp), y[k].ctypes.data_as(dp), 0)
        ref1 = np.ascontiguousarray(Hs[k]).copy()
        # the rank-2 form assumes a symmetric H: compare it on the symmetrised matrix below instead
        scale = np.abs(ref0).max()
        assert np.max(np.abs(got[k] - ref0)) <= 1e-12 * scale * n, (n, k)
    # symmetric H: both forms of the oracle and the GPU agree
    A = rng.standard_normal((n, n))
    Hsym = A @ A.T / n + np.eye(n)
    Hpad = np.zeros((1, n, ld))
    Hpad[0, :, :n] = Hsym
    Hd = torch.tensor(Hpad, device=dev)
    NLO.bfgs_update_gemm(Hd, torch.tensor(s[:1], device=dev), torch.tensor(y[:1], device=dev))
    torch.cuda.synchronize()
    r1 = Hsym.copy()
    lib.flo_bfgs_update(n, r1.ctypes.data_as(dp), s[0].ctypes.data_as(dp), y[0].ctypes.data_as(dp), 1)
    got = Hd.cpu().numpy()[0, :, :n]
    assert np.max(np.abs(got - r1)) <= 1e-11 * np.abs(r1).max() * n


def test_two_gemm_update_n1024_against_the_oracle_as_written():
    """n = 1024 (one full 8 x 8 grid of 128-tiles), asymmetric H, against the oracle's two sequential matmuls"""
    import FortranLibrary.NonlinearOptimization as NLO
    import la_cases as LC
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_bfgs_update.argtypes = [C.c_int, dp, dp, dp, C.c_int]
    lib.flo_set_sum_mode(O.SEQ, 64, 2)
    n = 1024
    r = LC.rng("bfgs_gemm1024")
    H = r.standard_normal((n, n))
    s = r.standard_normal(n)
    y = s * r.uniform(0.5, 2.0, n) + 0.1 * r.standard_normal(n)
    T, E = NLO.reduction_geometry(n)
    assert T * E == n
    dev = torch.device("cuda:0")
    Hd = torch.tensor(H[None].copy(), device=dev)
    NLO.bfgs_update_gemm(Hd, torch.tensor(s[None].copy(), device=dev), torch.tensor(y[None].copy(), device=dev))
    torch.cuda.synchronize()
    ref = np.ascontiguousarray(H).copy()
    lib.flo_bfgs_update(n, ref.ctypes.data_as(dp), s.ctypes.data_as(dp), y.ctypes.data_as(dp), 0)
    got = Hd.cpu().numpy()[0]
    # entries are sums of n^2 products of O(1) numbers scaled by rho^2 ...: compare to the largest entry; two
    # summation orders of n-term sums of size `scale` differ by ~ sqrt(n) eps scale
    assert np.abs(got - ref).max() <= 1e-13 * np.sqrt(n) * np.abs(ref).max()


def test_two_gemm_update_n4096_against_the_committed_oracle_digest():
    """the BASELINE config-4 size.  Reference: oracle update_form 0 run once on the CPU (tools/make_bfgs_gemm_golden.py,
    ~3 minutes), kept as sampled rows / columns / probe products in tests/golden/bfgs_gemm_4096.npz; plus the
    algebraically equal O(n^2) expansion U^T H U + rho s s^T = H - rho s (y^T H) - rho (H y) s^T + rho^2 (y^T H y) s s^T
    + rho s s^T evaluated here with numpy for the whole matrix."""
    import os
    import sys
    import FortranLibrary.NonlinearOptimization as NLO
    import la_cases as LC
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_bfgs_gemm_golden as MG
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bfgs_gemm_4096.npz"))
    n = 4096
    H, s, y = MG.inputs(n)
    assert np.array_equal(np.array([H.sum(), s.sum(), y.sum()]), fix["input_sums"])
    dev = torch.device("cuda:0")
    Hd = torch.tensor(H[None].copy(), device=dev)
    NLO.bfgs_update_gemm(Hd, torch.tensor(s[None].copy(), device=dev), torch.tensor(y[None].copy(), device=dev))
    torch.cuda.synchronize()
    got = Hd.cpu().numpy()[0]
    scale = float(fix["scale"])
    LC.compare_matrix(fix, "Hnew", got, n, rtol=1e-13 * np.sqrt(n), scale=scale)
    # whole matrix against the O(n^2) expansion.  Array layout is [col][row]: Hm = H.T is the matrix.
    Hm = H.T
    rho = 1.0 / (y @ s)
    Hy, yH = Hm @ y, y @ Hm
    full = Hm - rho * np.outer(s, yH) - rho * np.outer(Hy, s) + (rho * rho * (y @ Hy) + rho) * np.outer(s, s)
    assert np.abs(got.T - full).max() <= 1e-12 * np.sqrt(n) * scale
