"""The dense restatements of oracle/ pinned to the REFERENCE's own LinearAlgebra routines.

tests/golden/la_ref.npz holds outputs of /root/reference/source/LinearAlgebra.f90 -- compiled unmodified by
oracle/build_ref.sh against the image's MKL runtime -- on the seeded inputs of tests/la_cases.py
(tools/make_la_golden.py wrote it).  Checked here, on the CPU:
  * the fixture really is what the reference produces (only where oracle/_ref was built: the build container);
  * flo_dpotri_lower (+flo_syL2U), flo_dposv_lower, flo_dsysv agree with the reference to LAPACK rounding
    (My_dpotri LA.f90:798-812, dsyL2U 260-265, My_dposv 719-730, My_dsysv 695-703), including `info` of a matrix
    that is not positive definite and "b untouched on failure".
The -m gpu twin, tests/test_gpu_la_reference.py, holds the HIP kernels to the same fixture.
"""
import ctypes as C
import os

import numpy as np
import pytest

import la_cases as LC
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = np.load(os.path.join(ROOT, "tests", "golden", "la_ref.npz"))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfl_ref_la.so")
dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(dp)


def test_fixture_inputs_are_the_ones_the_reference_saw():
    assert np.array_equal(LC.input_digest(), FIX["input_digest"])


@pytest.mark.skipif(not (os.path.exists(REF_SO) and os.path.exists("/opt/conda/lib/libmkl_rt.so")),
                    reason="oracle/_ref not built here (needs /root/reference + MKL: bash oracle/build_ref.sh)")
def test_fixture_is_what_the_reference_library_returns():
    """re-run the reference (oracle/_ref) on a few cases: the committed fixture must be its output bit for bit"""
    os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
    ref = C.CDLL(REF_SO)
    for n in (5, 64, 200):
        A, b = LC.spd_case(n)
        W = np.array(A, order="F", copy=True)
        info = C.c_int(-1)
        ref.ref_my_dpotri(_p(W), C.c_int(n), C.byref(info))
        assert info.value == 0 and np.array_equal(np.tril(W), FIX[f"dpotri_{n}"])
        W = np.array(A, order="F", copy=True)
        x = b.copy()
        ref.ref_my_dposv(_p(W), _p(x), C.c_int(n), C.byref(info))
        assert np.array_equal(x, FIX[f"dposv_x_{n}"])
        A, b = LC.indefinite_case(n)
        W = np.array(A, order="F", copy=True)
        x = b.copy()
        ref.ref_my_dsysv(_p(W), _p(x), C.c_int(n))
        assert np.array_equal(x, FIX[f"dsysv_x_{n}"])


@pytest.mark.parametrize("n", LC.SPD_SIZES)
def test_oracle_dpotri_and_syl2u_match_the_reference(n):
    lib = O.lib()
    lib.flo_syL2U.argtypes = [dp, C.c_int]
    A, _ = LC.spd_case(n)
    W = np.array(A, order="F", copy=True)
    assert lib.flo_dpotri_lower(_p(W), n) == int(FIX[f"dpotri_info_{n}"]) == 0
    # cond(A) ~ 10: the inverse is defined to a few n eps; LAPACK's blocked sums and the oracle's sequential ones
    # differ by rounding only
    LC.compare_matrix(FIX, f"dpotri_{n}", np.tril(W), n, rtol=2e-13)
    lib.flo_syL2U(_p(W), n)
    assert np.array_equal(W, W.T)
    assert np.abs(W @ A - np.eye(n)).max() < 1e-12


@pytest.mark.parametrize("n", LC.SPD_SIZES)
def test_oracle_dposv_matches_the_reference(n):
    lib = O.lib()
    lib.flo_dposv_lower.restype = C.c_int
    lib.flo_dposv_lower.argtypes = [dp, dp, C.c_int]
    A, b = LC.spd_case(n)
    W = np.array(A, order="F", copy=True)
    x = b.copy()
    assert lib.flo_dposv_lower(_p(W), _p(x), n) == int(FIX[f"dposv_info_{n}"]) == 0
    ref = FIX[f"dposv_x_{n}"]
    assert np.abs(x - ref).max() <= 2e-13 * max(1.0, np.abs(ref).max())
    LC.compare_matrix(FIX, f"dposv_L_{n}", np.tril(W), n, rtol=2e-13)  # A harvests the Cholesky factor (LA.f90:717)


@pytest.mark.parametrize("n", LC.NONSPD_SIZES)
def test_oracle_reports_the_reference_info_for_a_matrix_that_is_not_positive_definite(n):
    lib = O.lib()
    lib.flo_dposv_lower.restype = C.c_int
    lib.flo_dposv_lower.argtypes = [dp, dp, C.c_int]
    A, b = LC.nonspd_case(n)
    W = np.array(A, order="F", copy=True)
    assert lib.flo_dpotri_lower(_p(W), n) == int(FIX[f"nonspd_dpotri_info_{n}"]) > 0
    W = np.array(A, order="F", copy=True)
    x = b.copy()
    assert lib.flo_dposv_lower(_p(W), _p(x), n) == int(FIX[f"nonspd_dposv_info_{n}"])
    assert np.array_equal(x, FIX[f"nonspd_dposv_x_{n}"]) and np.array_equal(x, b)  # b untouched (LA.f90:718)


@pytest.mark.parametrize("n", LC.SYM_SIZES)
def test_oracle_dsysv_matches_the_reference(n):
    """MKL's dsysv is Bunch-Kaufman, the restatement eliminates with partial pivoting: same solution to rounding"""
    lib = O.lib()
    lib.flo_dsysv.restype = C.c_int
    lib.flo_dsysv.argtypes = [dp, dp, C.c_int]
    A, b = LC.indefinite_case(n)
    W = np.array(A, order="F", copy=True)
    x = b.copy()
    assert lib.flo_dsysv(_p(W), _p(x), n) == 0
    ref = FIX[f"dsysv_x_{n}"]
    assert np.abs(x - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    assert np.abs(A @ ref - b).max() < 1e-12  # (and the reference's answer does solve the system)


def test_reference_outer_product_and_triangle_helpers_are_what_the_kernels_fuse():
    """vector_direct_product (LA.f90:105-114), sycp (241-249), dsyL2U (260-265) are exact operations: the fixture
    equals their definitions bit for bit, which is what the fused BFGS update / fl_dpotri_batched implement"""
    for (m, n) in LC.OUTER_SHAPES:
        a, b = LC.outer_case(m, n)
        assert np.array_equal(FIX[f"outer_{m}x{n}"], a[:, None] * b[None, :])
    for n in LC.TRI_SIZES:
        B = LC.tri_case(n)
        want = np.full((n, n), -1.0)
        il = np.tril_indices(n)
        want[il] = B[il]
        assert np.array_equal(FIX[f"sycp_{n}"], want)
        assert np.array_equal(FIX[f"syl2u_{n}"], np.tril(B) + np.tril(B, -1).T)
