"""-m gpu: the HIP dense kernels against the REFERENCE's LinearAlgebra routines (tests/golden/la_ref.npz, written by
tools/make_la_golden.py from the unmodified reference module + MKL; no oracle and no /root/reference at run time).

  fl_dposv_batched   <- My_dposv   LA.f90:719-730      fl_dpotri_batched <- My_dpotri 798-812 + dsyL2U 260-265
  fl_dsysv_batched   <- My_dsysv   695-703
  __linearalgebra_MOD_my_dgemm / _my_dgemm_t <- My_dgemm / My_dgemm_T 182-196 (cpp/FortranLibrary.hpp:48-63)
  __linearalgebra_MOD_my_dsyev <- My_dsyev 879-887
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import la_cases as LC

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(ROOT, "tests", "golden", "la_ref.npz"))
dp = C.POINTER(C.c_double)


def _nlo():
    import FortranLibrary.NonlinearOptimization as NLO
    return NLO


def _fl():
    from FortranLibrary.basic import FL
    return FL


def _padded(A, ld):
    n = A.shape[0]
    P = np.zeros((1, n, ld))
    P[0, :, :n] = A.T  # column-major [n][ld]: row j of the array is column j of the matrix (symmetric here anyway)
    return P


def test_inputs():
    assert np.array_equal(LC.input_digest(), FIX["input_digest"])


@pytest.mark.parametrize("n", LC.SPD_SIZES)
def test_dposv_and_dpotri_kernels_match_the_reference(n):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    A, b = LC.spd_case(n)
    Ad, bd = torch.tensor(_padded(A, ld), device=dev), torch.tensor(b[None, :].copy(), device=dev)
    info = NLO.dposv(Ad, bd).cpu().numpy()
    assert list(info) == [int(FIX[f"dposv_info_{n}"])] == [0]
    x, ref = bd.cpu().numpy()[0], FIX[f"dposv_x_{n}"]
    tol = 2e-13 if n < 512 else 5e-13  # from n = 512 on: the blocked path (MFMA summation order)
    assert np.abs(x - ref).max() <= tol * max(1.0, np.abs(ref).max())
    L = np.tril(Ad.cpu().numpy()[0, :, :n].T)  # A harvests the Cholesky factor (LA.f90:717)
    LC.compare_matrix(FIX, f"dposv_L_{n}", L, n, rtol=tol)
    Ad = torch.tensor(_padded(A, ld), device=dev)
    info = NLO.dpotri(Ad).cpu().numpy()
    assert list(info) == [int(FIX[f"dpotri_info_{n}"])] == [0]
    inv = Ad.cpu().numpy()[0, :, :n].T
    assert np.abs(inv - inv.T).max() <= (0.0 if n < 512 else 1e-15 * n * np.abs(inv).max())  # dsyL2U: both triangles
    LC.compare_matrix(FIX, f"dpotri_{n}", np.tril(inv), n, rtol=2e-13 if n < 512 else 5e-13)


@pytest.mark.parametrize("n", LC.NONSPD_SIZES)
def test_kernels_report_the_reference_info_when_not_positive_definite(n):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    T, E = NLO.reduction_geometry(n)
    A, b = LC.nonspd_case(n)
    Ad, bd = torch.tensor(_padded(A, T * E), device=dev), torch.tensor(b[None, :].copy(), device=dev)
    assert list(NLO.dposv(Ad, bd).cpu().numpy()) == [int(FIX[f"nonspd_dposv_info_{n}"])]
    assert np.array_equal(bd.cpu().numpy()[0], FIX[f"nonspd_dposv_x_{n}"])  # b untouched
    Ad = torch.tensor(_padded(A, T * E), device=dev)
    assert list(NLO.dpotri(Ad).cpu().numpy()) == [int(FIX[f"nonspd_dpotri_info_{n}"])]


@pytest.mark.parametrize("n", LC.SYM_SIZES)
def test_dsysv_kernel_matches_the_reference(n):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    T, E = NLO.reduction_geometry(n)
    A, b = LC.indefinite_case(n)
    Ad, bd = torch.tensor(_padded(A, T * E), device=dev), torch.tensor(b[None, :].copy(), device=dev)
    assert list(NLO.dsysv(Ad, bd).cpu().numpy()) == [0]
    x, ref = bd.cpu().numpy()[0], FIX[f"dsysv_x_{n}"]
    assert np.abs(x - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("shape", LC.GEMM_SHAPES)
def test_my_dgemm_symbols_match_the_reference(shape):
    FL = _fl()
    m, k, n = shape
    A, B = LC.gemm_case(m, k, n)
    At = np.asfortranarray(A.T.copy())
    iM, iK, iN = C.c_int(m), C.c_int(k), C.c_int(n)
    # |C_ij| ~ sqrt(k); two orders of summation of k products differ by ~ sqrt(k) eps |a||b|
    tol = 1e-13 * np.sqrt(k)
    Cm = np.asfortranarray(np.zeros((m, n)))
    FL.__linearalgebra_MOD_my_dgemm(A.ctypes.data_as(dp), B.ctypes.data_as(dp), Cm.ctypes.data_as(dp), C.byref(iM),
                                    C.byref(iK), C.byref(iN))
    LC.compare_matrix(FIX, f"dgemm_{m}x{k}x{n}", Cm, max(m, n), rtol=tol, scale=np.sqrt(k))
    Cm = np.asfortranarray(np.zeros((m, n)))
    FL.__linearalgebra_MOD_my_dgemm_t(At.ctypes.data_as(dp), B.ctypes.data_as(dp), Cm.ctypes.data_as(dp), C.byref(iM),
                                      C.byref(iK), C.byref(iN))
    LC.compare_matrix(FIX, f"dgemmT_{m}x{k}x{n}", Cm, max(m, n), rtol=tol, scale=np.sqrt(k))


@pytest.mark.parametrize("n", LC.EIG_SIZES)
def test_my_dsyev_symbol_matches_the_reference(n):
    FL = _fl()
    A = LC.eig_case(n)
    norm = np.abs(A).sum(axis=1).max()
    for job in (b"N", b"V"):
        S = np.asfortranarray(np.tril(A))  # only the lower triangle is referenced ('L', LA.f90:886)
        w = np.zeros(n)
        FL.__linearalgebra_MOD_my_dsyev(job, S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
        ref = FIX[f"dsyev_{job.decode()}_{n}"]
        assert np.all(np.diff(w) >= 0)  # ascending
        assert np.abs(w - ref).max() <= 1e-12 * norm
        if job == b"V":  # eigenvectors are defined up to sign (and rotation in near-degenerate pairs): residuals
            assert np.abs(A @ S - S * w[None, :]).max() <= max(50 * float(FIX[f"dsyev_V_resid_{n}"]), 1e-12 * norm)
            assert np.abs(S.T @ S - np.eye(n)).max() <= max(50 * float(FIX[f"dsyev_V_orth_{n}"]), 1e-12)


# ---- the blocked multi-workgroup Cholesky (csrc/fl_chol_blocked.hip): default from n = 512, forced here for smaller n
@pytest.fixture
def blocked_from_64():
    FL = _fl()
    old = FL.fl_set_chol_blocked_min_n(64)
    yield
    FL.fl_set_chol_blocked_min_n(old)


@pytest.mark.parametrize("n", [64, 200, 1024])
def test_blocked_cholesky_matches_the_reference(n, blocked_from_64):
    """the same fixtures as the one-workgroup kernels above, through the blocked path (MFMA summation order: LAPACK
    rounding, not bits)"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    A, b = LC.spd_case(n)
    Ad, bd = torch.tensor(_padded(A, ld), device=dev), torch.tensor(b[None, :].copy(), device=dev)
    assert list(NLO.dposv(Ad, bd).cpu().numpy()) == [0]
    x, ref = bd.cpu().numpy()[0], FIX[f"dposv_x_{n}"]
    assert np.abs(x - ref).max() <= 5e-13 * max(1.0, np.abs(ref).max())
    LC.compare_matrix(FIX, f"dposv_L_{n}", np.tril(Ad.cpu().numpy()[0, :, :n].T), n, rtol=5e-13)
    Ad = torch.tensor(_padded(A, ld), device=dev)
    assert list(NLO.dpotri(Ad).cpu().numpy()) == [0]
    inv = Ad.cpu().numpy()[0, :, :n].T
    assert np.abs(inv - inv.T).max() <= 1e-15 * np.abs(inv).max() * n
    LC.compare_matrix(FIX, f"dpotri_{n}", np.tril(inv), n, rtol=5e-13)


def test_blocked_cholesky_reports_info_and_leaves_b_alone(blocked_from_64):
    NLO = _nlo()
    dev = torch.device("cuda:0")
    n = 64
    T, E = NLO.reduction_geometry(n)
    A, b = LC.nonspd_case(n)
    A2, b2 = LC.spd_case(n)
    Ap = np.concatenate([_padded(A, T * E), _padded(A2, T * E)])  # a batch: one failing, one fine
    Ad, bd = torch.tensor(Ap, device=dev), torch.tensor(np.stack([b, b2]), device=dev)
    info = NLO.dposv(Ad, bd).cpu().numpy()
    assert list(info) == [int(FIX[f"nonspd_dposv_info_{n}"]), 0]
    got = bd.cpu().numpy()
    assert np.array_equal(got[0], b)  # untouched (LA.f90:718)
    assert np.abs(got[1] - FIX[f"dposv_x_{n}"]).max() <= 5e-13


@pytest.mark.parametrize("n,B", [(2048, 2), (5000, 1)])
def test_blocked_cholesky_large_matrices_against_lapack(n, B):
    """the default path from n = 512 on, and beyond the 4096 of the register geometries; against numpy's LAPACK"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    rng = np.random.default_rng(n)
    Ap = np.zeros((B, n, ld))
    As, bs = [], rng.uniform(-1, 1, (B, n))
    for k in range(B):
        G = rng.standard_normal((n, n))
        A = G @ G.T / n + np.eye(n)
        As.append(A)
        Ap[k, :, :n] = A
    Ad, bd = torch.tensor(Ap, device=dev), torch.tensor(bs, device=dev)
    assert list(NLO.dposv(Ad, bd).cpu().numpy()) == [0] * B
    for k in range(B):
        ref = np.linalg.solve(As[k], bs[k])
        assert np.abs(bd.cpu().numpy()[k] - ref).max() <= 1e-11 * np.abs(ref).max()
    Ad = torch.tensor(Ap, device=dev)
    assert list(NLO.dpotri(Ad).cpu().numpy()) == [0] * B
    for k in range(B):
        inv = Ad.cpu().numpy()[k, :, :n].T
        assert np.abs(inv @ As[k] - np.eye(n)).max() <= 1e-10


@pytest.mark.parametrize("n", [1, 2, 3, 7, 65, 130, 513])
def test_my_dsyev_values_only_edge_cases(n):
    """jobz = 'N' runs Householder tridiagonalisation + multisection (fl_dsyev_values): tiny and odd sizes, spectra with
    multiplicities (identity, a projector), zero sub-columns (a matrix that is already tridiagonal / diagonal: reflectors
    with tau = 0), graded entries.  Against numpy's LAPACK, to n eps ||A||."""
    FL = _fl()
    rng = np.random.default_rng(100 + n)
    G = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(G)
    cases = {
        "random": 0.5 * (G + G.T),
        "identity": np.eye(n),
        "diagonal": np.diag(np.arange(n, 0, -1.0)),
        "tridiagonal": np.diag(np.full(n, 2.0)) + np.diag(np.full(max(n - 1, 0), -1.0), 1) + np.diag(np.full(max(n - 1, 0), -1.0), -1),
        "projector": Q[:, : max(1, n // 3)] @ Q[:, : max(1, n // 3)].T,
        "graded": (Q * np.logspace(0, -12, n)[None, :]) @ Q.T,
        "zero": np.zeros((n, n)),
    }
    for name, A in cases.items():
        A = 0.5 * (A + A.T)
        S = np.asfortranarray(np.tril(A))
        w = np.full(n, np.nan)
        FL.__linearalgebra_MOD_my_dsyev(b"N", S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
        ref = np.linalg.eigvalsh(A)
        norm = max(np.abs(A).sum(axis=1).max(), 1e-300)
        assert np.all(np.diff(w) >= 0), name
        assert np.abs(w - ref).max() <= 4 * max(n, 4) * 2.3e-16 * norm + 1e-300, (name, np.abs(w - ref).max(), norm)


def _eig_cases(n, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(G)
    rep = np.repeat(np.arange(1, n // 8 + 2), 8)[:n].astype(float)
    off = np.ones(max(n - 1, 0))
    cases = {
        "random": 0.5 * (G + G.T),
        "identity": np.eye(n),
        "diagonal": np.diag(np.arange(n, 0, -1.0)),
        "tridiagonal": np.diag(np.full(n, 2.0)) - np.diag(off, 1) - np.diag(off, -1),
        "projector": Q[:, : max(1, n // 3)] @ Q[:, : max(1, n // 3)].T,
        "graded": (Q * np.logspace(0, -12, n)[None, :]) @ Q.T,
        "zero": np.zeros((n, n)),
        "wilkinson": np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(off, 1) + np.diag(off, -1),
        "clusters": (Q * rep[None, :]) @ Q.T,                                                 # eigenvalues of multiplicity 8
        "near_clusters": (Q * (rep + 1e-13 * rng.standard_normal(n))[None, :]) @ Q.T,        # ... split at the 1e-13 level
        "tiny": 1e-200 * 0.5 * (G + G.T),                                                     # squares underflow / overflow
        "huge": 1e+200 * 0.5 * (G + G.T),
        "subnormal": 1e-310 * 0.5 * (G + G.T),                                                # max |a_ij| < 2^-1022: the scaling factor itself must not overflow
        # numerically rank deficient: a null cluster of ~n - 20 eigenvalues right below genuine ones at 1e-14, 1e-13, ...
        # (shifts chained upwards walked into those and hundreds of vectors collapsed: invit_shift_kernel)
        "hilbert": 1.0 / (np.arange(n)[:, None] + np.arange(n)[None, :] + 1.0),
        "low_rank": G[:, : max(1, n // 5)] @ G[:, : max(1, n // 5)].T,
    }
    ar = np.diag(rng.standard_normal(n))
    ar[-1, :] = ar[:, -1] = rng.standard_normal(n)
    cases["arrow"] = ar
    return {k: 0.5 * (v + v.T) for k, v in cases.items()}


def _check_basis(name, A, S, w, n):
    norm = np.abs(A).sum(axis=1).max()
    e = -int(np.floor(np.log2(norm))) if norm > 0 else 0
    An, wn = np.ldexp(A, e), np.ldexp(w, e)  # (exact; the extreme scalings would under / overflow in the products below)
    ref = np.linalg.eigvalsh(An)
    assert np.all(np.diff(w) >= 0), name
    # (eigenvalues that are subnormal numbers themselves carry fewer bits: 1e-310 has 44, i.e. 3e-14 relative)
    wtol = 1e-13 if name == "subnormal" else 4 * max(n, 4) * 2.3e-16
    assert np.abs(wn - ref).max() <= wtol, (name, np.abs(wn - ref).max())
    # the bar of the library's own device-side check (512 eps on the tridiagonal's vectors) plus the two orthogonal
    # transformations around it; typical figures are ~1e-16 (residual) and ~2e-15 (orthogonality), see DESIGN.md 8
    bar = max(600, 4 * n) * 2.3e-16
    assert np.abs(An @ S - S * wn[None, :]).max() <= bar, (name, np.abs(An @ S - S * wn[None, :]).max())
    assert np.abs(S.T @ S - np.eye(n)).max() <= bar, (name, np.abs(S.T @ S - np.eye(n)).max())


@pytest.mark.parametrize("n", [1, 2, 3, 7, 65, 130, 513])
def test_my_dsyev_with_vectors_edge_cases(n):
    """jobz = 'V' runs tridiagonalisation + inverse iteration + Cholesky-QR + back-transformation (fl_dsyev_vectors):
    multiple and nearly multiple eigenvalues (identity, a projector, clusters of 8, Wilkinson's pairs), matrices that are
    already tridiagonal / diagonal, graded and extremely scaled entries.  Residuals and orthogonality to n eps."""
    FL = _fl()
    for name, A in _eig_cases(n, 300 + n).items():
        S = np.asfortranarray(np.tril(A))
        w = np.full(n, np.nan)
        FL.__linearalgebra_MOD_my_dsyev(b"V", S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
        _check_basis(name, A, S, w, n)


@pytest.mark.parametrize("n", [5, 64, 200, 1024, 1100, 2500, 3072, 4096])
def test_fl_dsyev_vectors_passes_its_own_check_and_reports_it(n):
    """the device entry itself: FL_OK (not the Jacobi fallback) on every case but the extreme scalings, which the legacy
    symbol rescales first (dsyev's dlascl) and the device entry reports as 1 = 'check failed'; its quality figures are
    those of the basis it returns"""
    FL = _fl()
    FL.fl_dsyev_vectors_workspace_bytes.restype = C.c_size_t
    dev = torch.device("cuda:0")
    wsb = FL.fl_dsyev_vectors_workspace_bytes(n)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
    # (n >= 3072: the 1024-thread tridiagonalisation kernel with its own kept reflectors -- the device-side check looks at the
    # tridiagonal's vectors only, so the final basis is held against A here: ADVICE r03)
    names = (("random", "projector") if n >= 3072 else ("random", "projector", "near_clusters", "tiny", "hilbert")) if n > 1024 else None
    for name, A in _eig_cases(n, 500 + n).items():
        if names and name not in names:
            continue
        Ad = torch.tensor(np.tril(A).T.copy(), device=dev)  # column-major, lower triangle only
        w = torch.zeros(n, dtype=torch.float64, device=dev)
        q = (C.c_double * 3)()
        rc = FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(Ad.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()),
                                 C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), q, None)
        torch.cuda.synchronize()
        if name in ("tiny", "huge", "subnormal"):
            assert rc in (0, 1), (name, rc)
            continue
        if n >= 3072 and name == "projector" and rc == 1:
            continue  # (~2000-fold eigenvalues at this size: the fast path may give up -- it says so, the legacy symbol then runs Jacobi)
        assert rc == 0, (name, rc, list(q))
        assert q[0] <= 512 * 2.3e-16 and q[1] <= 512 * 2.3e-16 and q[2] in (1.0, 2.0, 3.0), (name, list(q))
        _check_basis(name, A, Ad.cpu().numpy().T, w.cpu().numpy(), n)
    # argument checks
    assert FL.fl_dsyev_vectors(C.c_int(n), None, C.c_int(n), C.c_void_p(ws.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), None, None) == -1
    assert FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(ws.data_ptr()), C.c_int(n), C.c_void_p(ws.data_ptr()), C.c_void_p(ws.data_ptr()),
                               C.c_size_t(8), None, None) == -3


def test_fl_dsyev_vectors_structured_families_stay_on_the_fast_path():
    """tools/dsyev_stress.py's ten families (prescribed multiplicities, low rank, Hilbert, arrow, weakly coupled blocks, graded,
    glued Wilkinson, Toeplitz, clusters at several scales), 80 matrices of n = 2 ... 400: the device-side check never sends
    one to the Jacobi fallback, residuals and orthogonality at LAPACK's level"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dsyev_stress as DS
    FL = _fl()
    FL.fl_dsyev_vectors_workspace_bytes.restype = C.c_size_t
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2024)
    for t in range(80):
        n = int(rng.choice([2, 3, 5, 8, 17, 33, 64, 65, 100, 129, 200, 257, 400]))
        A = DS.family(rng, t % 10, n)
        A = 0.5 * (A + A.T)
        norm = max(np.abs(A).sum(axis=1).max(), 1e-300)
        Ad = torch.tensor(np.tril(A).T.copy(), device=dev)
        w = torch.zeros(n, dtype=torch.float64, device=dev)
        wsb = FL.fl_dsyev_vectors_workspace_bytes(n)
        ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
        q = (C.c_double * 3)()
        rc = FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(Ad.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()), C.c_void_p(ws.data_ptr()),
                                 C.c_size_t(wsb), q, None)
        torch.cuda.synchronize()
        assert rc == 0, (t, n, list(q))
        V, wh = Ad.cpu().numpy().T, w.cpu().numpy()
        assert np.abs(A @ V - V * wh[None, :]).max() <= 1e-13 * norm, (t, n)
        assert np.abs(V.T @ V - np.eye(n)).max() <= 1e-13, (t, n)
        assert np.abs(wh - np.linalg.eigvalsh(A)).max() <= 1e-13 * norm, (t, n)


def test_python_dsyev_on_device_tensors():
    """NonlinearOptimization.dsyev: the device entries behind My_dsyev for a torch tensor (rows of V = eigenvectors)"""
    NLO = _nlo()
    dev = torch.device("cuda:0")
    for n in (7, 300):
        A = LC.eig_case(n) if n in (5, 64, 200, 1024) else _eig_cases(n, 9)["random"]
        At = torch.tensor(np.tril(A).T.copy(), device=dev)  # row j of the tensor = column j of the lower triangle
        w, V = NLO.dsyev(At)
        torch.cuda.synchronize()
        wh, Vh = w.cpu().numpy(), V.cpu().numpy()
        norm = np.abs(A).sum(axis=1).max()
        assert np.abs(wh - np.linalg.eigvalsh(A)).max() <= 1e-13 * norm
        assert np.abs(A @ Vh.T - Vh.T * wh[None, :]).max() <= 1e-13 * norm and np.abs(Vh @ Vh.T - np.eye(n)).max() <= 1e-13
        w2, none = NLO.dsyev(torch.tensor(np.tril(A).T.copy(), device=dev), vectors=False)
        assert none is None and np.abs(w2.cpu().numpy() - wh).max() <= 1e-13 * norm


def test_my_dsyev_jacobi_on_request(monkeypatch):
    """FL_DSYEV_JACOBI=1: the cyclic Jacobi path (the fallback of the 'V' job) stays reachable and correct"""
    FL = _fl()
    monkeypatch.setenv("FL_DSYEV_JACOBI", "1")
    n = 96
    for name, A in _eig_cases(n, 7).items():
        if name in ("tiny", "huge", "graded"):
            continue
        S = np.asfortranarray(np.tril(A))
        w = np.full(n, np.nan)
        FL.__linearalgebra_MOD_my_dsyev(b"V", S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
        norm = max(np.abs(A).sum(axis=1).max(), 1e-300)
        assert np.abs(A @ S - S * w[None, :]).max() <= 1e-12 * norm and np.abs(S.T @ S - np.eye(n)).max() <= 1e-12, name


def test_blocked_cholesky_a_failing_matrix_in_the_middle_of_a_batch_disturbs_nothing():
    """n = 600 (blocked multi-workgroup path): matrix 2 of 5 is indefinite from its 3rd 64-column block on.  The others
    are solved as if it were not there; the failing one reports LAPACK's info, keeps its right-hand side, has its first
    two block columns factorised and, behind the failing block, the Schur complement of those two steps -- no NaN anywhere."""
    import torch
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n = 5, 600
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    g = torch.Generator(device="cpu").manual_seed(11)
    G = torch.randn(B, n, n, generator=g, dtype=torch.float64)
    spd = G @ G.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64)
    bad_at = 150  # inside the third block (columns 128..191)
    spd[2, bad_at, bad_at] = -5.0
    A0 = torch.zeros(B, n, ld, dtype=torch.float64)
    A0[:, :, :n] = spd  # symmetric: row / column major agree
    rhs0 = torch.randn(B, n, generator=g, dtype=torch.float64)
    A, rhs = A0.to(dev), rhs0.to(dev)
    info = NLO.dposv(A, rhs).cpu().numpy()
    Ah, xh = A.cpu(), rhs.cpu()
    assert info[2] == bad_at + 1 and all(info[k] == 0 for k in (0, 1, 3, 4))
    assert bool(torch.isfinite(Ah).all()) and bool(torch.isfinite(xh).all())
    for k in (0, 1, 3, 4):
        assert float((spd[k] @ xh[k] - rhs0[k]).abs().max()) < 1e-9
    assert torch.equal(xh[2], rhs0[2])  # b untouched (My_dposv, LA.f90:718)
    # column-major A[k][col][row]: block columns 0..127 hold the factor (L L^T reproduces that part of the matrix) ...
    L = torch.tril(Ah[2, :128, :n].T)[:, :128]  # rows x first 128 columns
    assert float((L[:128] @ L[:128].T - spd[2, :128, :128]).abs().max()) < 1e-10
    assert float((L[128:] @ L[:128].T - spd[2, 128:, :128]).abs().max()) < 1e-10
    # ... and behind the failing block column the lower triangle holds what a right-looking factorisation has there after
    # two completed block steps -- the Schur complement A22 - L21 L21^T -- untouched by the failing step and the ones after
    S = spd[2, 192:, 192:] - L[192:] @ L[192:].T
    got = Ah[2, 192:n, 192:n].T  # [row, col]
    assert float((torch.tril(got) - torch.tril(S)).abs().max()) < 1e-10
    # the inverse: the healthy matrices are inverted, the failing one only reports
    A2 = A0.to(dev)
    info2 = NLO.dpotri(A2).cpu().numpy()
    assert info2[2] == bad_at + 1 and all(info2[k] == 0 for k in (0, 1, 3, 4))
    for k in (0, 1, 3, 4):
        assert float((A2[k, :, :n].cpu() @ spd[k] - torch.eye(n, dtype=torch.float64)).abs().max()) < 1e-8


def test_batches_beyond_65535_matrices_go_in_chunks():
    """the strided DGEMM (matrix index in gridDim.y) under the batched TrustRegion: 70 000 tiny systems at once"""
    import torch
    import FortranLibrary.NonlinearOptimization as NLO
    dev = torch.device("cuda:0")
    B, n, m = 70000, 3, 4
    t = torch.linspace(0.5, 2.0, B, dtype=torch.float64, device=dev)

    def fun(x, rq):  # r = (x0 - t, x1 - 2 t, x2 + t, x0 x1 - 2 t^2): root (t, 2t, -t)
        r = torch.stack([x[:, 0] - t, x[:, 1] - 2 * t, x[:, 2] + t, x[:, 0] * x[:, 1] - 2 * t * t], dim=1)
        J = torch.zeros(x.shape[0], n, m, dtype=torch.float64, device=dev)
        J[:, 0, 0] = 1.0
        J[:, 1, 1] = 1.0
        J[:, 2, 2] = 1.0
        J[:, 0, 3] = x[:, 1]
        J[:, 1, 3] = x[:, 0]
        return r, J
    x = torch.ones(B, n, dtype=torch.float64, device=dev)
    out = NLO.TrustRegion(x, fun, m, Precision=1e-10)
    sol = torch.stack([t, 2 * t, -t], dim=1)
    assert float((x - sol).abs().max()) < 1e-8 and bool((out["reason"] == 3).all())
