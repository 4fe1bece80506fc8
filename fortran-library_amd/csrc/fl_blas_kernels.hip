// fl_blas_kernels.hip -- the two dense primitives of LinearAlgebra.f90 that the reference's C++ header exports next to
// the optimisers (cpp/FortranLibrary.hpp:48-63), as kernels of this library instead of vendor calls:
//
//   fl_dgemm   <- My_dgemm / My_dgemm_T  LinearAlgebra.f90:182-196  (dgemm 'N','N' / 'T','N', alpha = 1, beta = 0)
//   fl_dsyev   <- My_dsyev               LinearAlgebra.f90:879-887  (dsyev jobtype,'L': ascending eigenvalues,
//                                                                      normalised eigenvectors)
//
// DGEMM: C(M,N) = op(A) B on the f64 matrix cores (v_mfma_f64_16x16x4_f64), 128 x 128 tile per workgroup of 2 x 4 waves
// (8 accumulator tiles each, two waves per SIMD -- the layout bfgs_gemm_kernel measured best), BK = 16, BOTH operands
// streamed through a double-buffered LDS image stored k-major (X[kk][index]), so that either fragment is the same
// conflict-free read X[4 ks + (lane >> 4)][base + (lane & 15)].  The MFMA's row index is given to the N side and its
// column index to the M side: accumulator register r of lane l is then C(m = base + (l & 15), n = base + (l >> 4) + 4 r)
// and the 16 lanes of a row write 128 contiguous bytes of the column-major C.
//
// DSYEV, three routes (the legacy symbol My_dsyev chooses, csrc/fl_linalg.cpp):
//   'N': Householder tridiagonalisation, one launch per reflector (tridiag_step_kernel), + multisection (fl_dsyev_values);
//   'V': the same tridiagonalisation with the reflectors kept, inverse iteration, Cholesky-QR / Newton-Schulz on the
//        matrix cores, back-transformation (csrc/fl_eig_vectors.hip: fl_dsyev_vectors), checked on the device;
//   the fallback of 'V' when that check fails, and beyond n = 6144: cyclic two-sided Jacobi (fl_dsyev_jacobi), below.
// Jacobi: cyclic two-sided with the round-robin (tournament) ordering: a step applies n/2 disjoint rotations at
// once, A <- J^T (A J), V <- V J; a sweep is n - 1 steps.  ONE launch per step (jacobi_step_kernel): the workgroup of a
// pair stages its two columns in LDS, applies the previous step's pending row rotations to them, forms its own rotation
// from the finished 2x2 block and writes the rotated columns (and V's) in place; jacobi_rows applies the last pending
// row rotations once per sweep, for the convergence test.
// Converged when the off-diagonal norm is below eps * the Frobenius norm (checked once per sweep).  Jacobi's
// eigenvalues are at least as accurate as the QR iteration's behind LAPACK's dsyev; eigenvectors are defined up to
// sign, like LAPACK's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"

namespace fl {

using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int DBK = 16, DPAD = 16; // DPAD: rows of the LDS image 32 banks apart

struct GemmArgs {
    int transA, transB, M, K, N, lda, ldb, ldc, lower;
    double alpha, beta;
    const double *A, *B;
    double *C;
    size_t strideA, strideB, strideC;
};

// X is "index-contiguous" (element (idx, k) at idx + k*ld): A for 'N'.  Thread t stages 8 doubles of row kk.
// X is "k-contiguous"     (element (idx, k) at k + idx*ld): A for 'T', and B.  Thread t stages k = 8*(t&1).. of idx t>>1.
// BT x BT tile per workgroup of WGM x WGN waves = BT*4 threads: 128 (2 x 4 waves, 8 accumulator tiles per wave) for
// large problems, 64 (2 x 2 waves, 4 tiles) while 128-tiles would leave CUs idle (n = 1024: 64 tiles for 256 CUs).
template <int BT, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void dgemm_kernel(GemmArgs g)
{
    // C = alpha op(A) op(B) + beta C for matrix blockIdx.y of a strided batch.  transB: B is given N x K (element
    // (k, n) at n + k*ldb) -- the layout of L21 in "A22 -= L21 L21^T".  lower: only the tiles that touch the lower
    // triangle (m >= n) are computed (symmetric rank-k updates of a matrix whose upper triangle is not referenced).
    const int transA = g.transA, transB = g.transB, M = g.M, K = g.K, N = g.N, lda = g.lda, ldb = g.ldb, ldc = g.ldc;
    const double *A = g.A + (size_t)blockIdx.y * g.strideA, *B = g.B + (size_t)blockIdx.y * g.strideB;
    double *C = g.C + (size_t)blockIdx.y * g.strideC;
    constexpr int DBM = BT, DBN = BT;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TI = DBN / 16 / WGM, TJ = DBM / 16 / WGN; // accumulator tiles per wave: rows (N side) x columns (M side)
    constexpr int TPR = BT / 4;                             // threads per staged row of BT doubles (4 doubles each)
    static_assert(NT == 4 * BT, "staging: 16 x BT doubles per operand and k-block, 4 per thread");
    __shared__ __attribute__((aligned(16))) double As[2][DBK][DBM + DPAD]; // op(A)(m, k) at [k][m]
    __shared__ __attribute__((aligned(16))) double Bs[2][DBK][DBN + DPAD]; // B(k, n)     at [k][n]
    // tile -> (m block, n block); consecutive workgroups share the B panel (same n block) -- they run close in time
    const int tiles_m = (M + DBM - 1) / DBM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * DBM, n0 = tn * DBN;
    if (g.lower && m0 + DBM <= n0) return; // the whole tile lies above the diagonal
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WGN, wj = wave % WGN; // wave's block of the tile: rows (n) wi, columns (m) wj
    const int lr = lane & 15, lq = lane >> 4;

    f64x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    // staging registers: 4 doubles of A and 4 of B per thread and k-block (16 x 128 doubles / 512 threads)
    double a_st[4], b_st[4];
    auto gload = [&](int k0) {
        if (!transA) { // row kk = tid / TPR, columns 4*(tid % TPR)..+3 of the tile
            const int kk = tid / TPR, mm = (tid % TPR) * 4;
            const int k = k0 + kk;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = m0 + mm + u;
                a_st[u] = (k < K && m < M) ? A[(size_t)k * lda + m] : 0.0;
            }
        } else { // index m = tid / 4, k = 4*(tid % 4)..+3
            const int mm = tid >> 2, kk = (tid & 3) * 4;
            const int m = m0 + mm;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + kk + u;
                a_st[u] = (k < K && m < M) ? A[(size_t)m * lda + k] : 0.0;
            }
        }
        if (transB) { // row kk = tid / TPR, columns 4*(tid % TPR)..+3: (k, n) at n + k*ldb
            const int kk = tid / TPR, nn = (tid % TPR) * 4;
            const int k = k0 + kk;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = n0 + nn + u;
                b_st[u] = (k < K && n < N) ? B[(size_t)k * ldb + n] : 0.0;
            }
        } else {
            const int nn = tid >> 2, kk = (tid & 3) * 4;
            const int n = n0 + nn;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + kk + u;
                b_st[u] = (k < K && n < N) ? B[(size_t)n * ldb + k] : 0.0;
            }
        }
    };
    auto lstore = [&](int buf) {
        if (!transA) {
            const int kk = tid / TPR, mm = (tid % TPR) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[buf][kk][mm + u] = a_st[u];
        } else {
            const int mm = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[buf][kk + u][mm] = a_st[u];
        }
        if (transB) {
            const int kk = tid / TPR, nn = (tid % TPR) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[buf][kk][nn + u] = b_st[u];
        } else {
            const int nn = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[buf][kk + u][nn] = b_st[u];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += DBK) {
        const bool more = k0 + DBK < K;
        if (more) gload(k0 + DBK);
#pragma unroll
        for (int ks = 0; ks < DBK / 4; ++ks) {
            const int kk = 4 * ks + lq;
            double fi[TI], fj[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) fi[i] = Bs[buf][kk][16 * TI * wi + 16 * i + lr]; // MFMA "A" fragment: (row n, k)
#pragma unroll
            for (int j = 0; j < TJ; ++j) fj[j] = As[buf][kk][16 * TJ * wj + 16 * j + lr]; // MFMA "B" fragment: (k, col m)
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fi[i], fj[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * TI * wi + 16 * i + lq + 4 * r;
            if (n < N) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int m = m0 + 16 * TJ * wj + 16 * j + lr;
                    if (m < M) {
                        double *cp = C + (size_t)n * ldc + m;
                        *cp = (g.beta == 0.0) ? g.alpha * acc[i][j][r] : g.alpha * acc[i][j][r] + g.beta * *cp;
                    }
                }
            }
        }
    }
}

// The same product for SMALL problems (fewer than two 64-tiles per CU: n <= 1024 for one matrix).  There the kernel above
// is bound by latency, not by the matrix cores: one k-block of MFMAs (0.1-0.2 us) per global-load round trip (~1.3 us), the
// next block's loads being issued only one block ahead -- n = 1024: 97 us = 64 k-blocks x 1.5 us, whatever the tile.
// Here the operand loads run ST k-blocks ahead through a ring of staging registers (unrolled, so the ring's slots are
// registers), unconditionally (beyond K: the last block again, never used; edge tiles: clamped addresses x a 0 / 1 mask
// -- a load under a test, or a select on its result, is followed by s_waitcnt vmcnt(0) and would undo the distance),
// and the tiles are 32 x 32 for two waves (1024 workgroups at n = 1024: eight waves per CU instead of four).  Measured
// (profiles/r03/dgemm_small.txt): n = 256: 23.4 -> 14.0 us, 512: 43.7 -> 25.1, 768: 73.4 -> 51.0, 1024: 96.0 -> 70.1 us
// (22.4 -> 30.6 TFLOP/s); ST = 2 / 4: 83 / 92 us; the same ring on 64-tiles: 103 us.  What is left per k-block (1.1 us) is
// the two dependent MFMA chains of a wave and the barrier, not the loads.
template <int BT, int WGM, int WGN, int ST>
__global__ __launch_bounds__(WGM * WGN * 64) void dgemm_small_kernel(GemmArgs g)
{
    const int transA = g.transA, transB = g.transB, M = g.M, K = g.K, N = g.N, lda = g.lda, ldb = g.ldb, ldc = g.ldc;
    const double *A = g.A + (size_t)blockIdx.y * g.strideA, *B = g.B + (size_t)blockIdx.y * g.strideB;
    double *C = g.C + (size_t)blockIdx.y * g.strideC;
    constexpr int DBM = BT, DBN = BT;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TI = DBN / 16 / WGM, TJ = DBM / 16 / WGN;
    constexpr int TPR = BT / 4;
    static_assert(NT == 4 * BT, "staging: 16 x BT doubles per operand and k-block, 4 per thread");
    __shared__ __attribute__((aligned(16))) double As[2][DBK][DBM + DPAD];
    __shared__ __attribute__((aligned(16))) double Bs[2][DBK][DBN + DPAD];
    const int tiles_m = (M + DBM - 1) / DBM;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * DBM, n0 = tn * DBN;
    if (g.lower && m0 + DBM <= n0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WGN, wj = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    f64x4 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int nkb = (K + DBK - 1) / DBK;
    const bool plain = (m0 + DBM <= M) && (n0 + DBN <= N) && (K % DBK == 0); // nothing of this tile's operands is out of range
    double a_st[ST][4], b_st[ST][4];
    // operand X: idx_contiguous: element (idx, k) at idx + k*ld, else at k + idx*ld; the thread's four consecutive doubles
    auto gload_one = [&](const double *X, int ld, bool idx_contiguous, int idx0, int IDX, int k0, double(&st)[4]) {
        if (idx_contiguous) {
            const int k = k0 + tid / TPR, i0 = idx0 + (tid % TPR) * 4;
            if (plain) {
                const double *q = X + (size_t)k * ld + i0;
#pragma unroll
                for (int u = 0; u < 4; ++u) st[u] = q[u];
            } else {
                const int kc = k < K ? k : K - 1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u, ic = i < IDX ? i : IDX - 1;
                    st[u] = X[(size_t)kc * ld + ic] * ((k < K && i < IDX) ? 1.0 : 0.0);
                }
            }
        } else {
            const int i = idx0 + (tid >> 2), kb = k0 + (tid & 3) * 4;
            if (plain) {
                const double *q = X + (size_t)i * ld + kb;
#pragma unroll
                for (int u = 0; u < 4; ++u) st[u] = q[u];
            } else {
                const int ic = i < IDX ? i : IDX - 1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u, kc = k < K ? k : K - 1;
                    st[u] = X[(size_t)ic * ld + kc] * ((k < K && i < IDX) ? 1.0 : 0.0);
                }
            }
        }
    };
    auto gload = [&](int kb, double(&sa)[4], double(&sb)[4]) { // k-block kb (beyond the last: the last again, never used)
        const int k0 = (kb < nkb ? kb : nkb - 1) * DBK;
        gload_one(A, lda, !transA, m0, M, k0, sa);
        gload_one(B, ldb, transB != 0, n0, N, k0, sb);
    };
    auto lstore = [&](int buf, const double(&sa)[4], const double(&sb)[4]) {
        if (!transA) {
            const int kk = tid / TPR, mm = (tid % TPR) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[buf][kk][mm + u] = sa[u];
        } else {
            const int mm = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[buf][kk + u][mm] = sa[u];
        }
        if (transB) {
            const int kk = tid / TPR, nn = (tid % TPR) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[buf][kk][nn + u] = sb[u];
        } else {
            const int nn = tid >> 2, kk = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[buf][kk + u][nn] = sb[u];
        }
    };
#pragma unroll
    for (int s = 0; s < ST; ++s) gload(s, a_st[s], b_st[s]);
    lstore(0, a_st[0], b_st[0]);
    __syncthreads();
    int buf = 0;
    for (int kb = 0; kb < nkb; kb += ST) {
#pragma unroll
        for (int s = 0; s < ST; ++s) { // block kb + s is in LDS[buf]; slot s is free, slots s+1 ... hold the blocks after it
            if (kb + s < nkb) {
                gload(kb + s + ST, a_st[s], b_st[s]);
#pragma unroll
                for (int ks = 0; ks < DBK / 4; ++ks) {
                    const int kk = 4 * ks + lq;
                    double fi[TI], fj[TJ];
#pragma unroll
                    for (int i = 0; i < TI; ++i) fi[i] = Bs[buf][kk][16 * TI * wi + 16 * i + lr];
#pragma unroll
                    for (int j = 0; j < TJ; ++j) fj[j] = As[buf][kk][16 * TJ * wj + 16 * j + lr];
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fi[i], fj[j], acc[i][j], 0, 0, 0);
                }
                if (kb + s + 1 < nkb) lstore(buf ^ 1, a_st[(s + 1) % ST], b_st[(s + 1) % ST]);
                __syncthreads();
                buf ^= 1;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * TI * wi + 16 * i + lq + 4 * r;
            if (n < N) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int m = m0 + 16 * TJ * wj + 16 * j + lr;
                    if (m < M) {
                        double *cp = C + (size_t)n * ldc + m;
                        *cp = (g.beta == 0.0) ? g.alpha * acc[i][j][r] : g.alpha * acc[i][j][r] + g.beta * *cp;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ Jacobi eigensolver
// pair k of step t among n2 = even number of players (circle method; player n2-1 stays put)
__device__ __forceinline__ void jacobi_pair(int n2, int t, int k, int &p, int &q)
{
    const int r = n2 - 1;
    int a, b;
    if (k == 0) {
        a = r;
        b = t % r;
    } else {
        a = (t + k) % r;
        b = (t - k + r) % r;
    }
    p = a < b ? a : b;
    q = a < b ? b : a;
}
// symmetric 2x2 Schur decomposition: J = [c s; -s c] with (J^T [app apq; apq aqq] J) diagonal (Golub & Van Loan 8.4)
__device__ __forceinline__ void jacobi_rotation(double app, double aqq, double apq, double &c, double &s)
{
    if (apq == 0.0 || !(fabs(apq) > 1e-300)) {
        c = 1.0;
        s = 0.0;
        return;
    }
    const double tau = (aqq - app) / (2.0 * apq);
    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    c = 1.0 / sqrt(1.0 + t * t);
    s = t * c;
}

// A = J^T B: column j staged in LDS (dynamic: n doubles), one thread per pair.  grid n, block 256
__global__ __launch_bounds__(256) void jacobi_rows_kernel(int n, int n2, int step, const double *Bm, double *A, int ld,
                                                          const double *cs)
{
    extern __shared__ double col[];
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) col[i] = Bm[(size_t)j * ld + i];
    __syncthreads();
    for (int k = threadIdx.x; k < n2 / 2; k += 256) {
        int p, q;
        jacobi_pair(n2, step, k, p, q);
        if (q >= n) continue;
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double bp = col[p], bq = col[q];
        col[p] = c * bp - s * bq;
        col[q] = s * bp + c * bq;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) A[(size_t)j * ld + i] = col[i];
}
// One launch per step (round 2): with B_t = A_{t-1} J_t and A_t = J_t^T B_t, the next step's column pass needs of A_t
// only the two columns of its own pair -- B_{t+1}[:, {p,q}] = (J_t^T B_t)[:, {p,q}] J_{t+1} -- so the workgroup of pair
// (p, q) stages B_t's columns p, q in LDS, applies the PREVIOUS step's row rotations to them (rotations from cs_prev),
// forms its own rotation from the finished 2x2 block and writes the rotated columns back in place (the pairs of a
// step are disjoint).  A itself is materialised only where the convergence test wants it (jacobi_rows_kernel).  Same
// arithmetic per element as the two-pass form; half the launches and half the traffic.
// grid n2/2, block 256, dynamic LDS 2 n doubles.  prev_step < 0: no rotation pending (the first step).
__global__ __launch_bounds__(256) void jacobi_step_kernel(int n, int n2, int prev_step, int step, double *Bm, double *V, int ld,
                                                          const double *cs_prev, double *cs_next, int want_vectors)
{
    extern __shared__ double col[];
    double *cp = col, *cq = col + n;
    const int k = blockIdx.x;
    int p, q;
    jacobi_pair(n2, step, k, p, q);
    const bool has_q = q < n; // (q = n: the padding player of an odd n -- column p only takes the pending row rotations)
    double *gp = Bm + (size_t)p * ld, *gq = Bm + (size_t)(has_q ? q : p) * ld;
    for (int i = threadIdx.x; i < n; i += 256) {
        cp[i] = gp[i];
        if (has_q) cq[i] = gq[i];
    }
    __syncthreads();
    if (prev_step >= 0) {
        for (int kk = threadIdx.x; kk < n2 / 2; kk += 256) {
            int pp, qq;
            jacobi_pair(n2, prev_step, kk, pp, qq);
            if (qq >= n) continue;
            const double c = cs_prev[2 * kk], s = cs_prev[2 * kk + 1];
            const double bp = cp[pp], bq = cp[qq];
            cp[pp] = c * bp - s * bq;
            cp[qq] = s * bp + c * bq;
            if (has_q) {
                const double dp = cq[pp], dq = cq[qq];
                cq[pp] = c * dp - s * dq;
                cq[qq] = s * dp + c * dq;
            }
        }
        __syncthreads();
    }
    if (!has_q) {
        for (int i = threadIdx.x; i < n; i += 256) gp[i] = cp[i];
        if (threadIdx.x == 0) {
            cs_next[2 * k] = 1.0;
            cs_next[2 * k + 1] = 0.0;
        }
        return;
    }
    double c, s;
    jacobi_rotation(cp[p], cq[q], cq[p], c, s);
    if (threadIdx.x == 0) {
        cs_next[2 * k] = c;
        cs_next[2 * k + 1] = s;
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        const double ap = cp[i], aq = cq[i];
        gp[i] = c * ap - s * aq;
        gq[i] = s * ap + c * aq;
    }
    if (want_vectors) {
        double *vp = V + (size_t)p * ld, *vq = V + (size_t)q * ld;
        for (int i = threadIdx.x; i < n; i += 256) {
            const double a = vp[i], b = vq[i];
            vp[i] = c * a - s * b;
            vq[i] = s * a + c * b;
        }
    }
}
// out[0] = sum of squares of the strictly lower triangle, out[1] = of the diagonal (atomics: a convergence test only)
__global__ __launch_bounds__(256) void jacobi_norms_kernel(int n, const double *A, int ld, double *out)
{
    __shared__ double part[2][4];
    const int j = blockIdx.x;
    double off = 0.0, dia = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = A[(size_t)j * ld + i];
        if (i > j) off += v * v;
        else if (i == j) dia += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        off += __shfl_down(off, o);
        dia += __shfl_down(dia, o);
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = off;
        part[1][threadIdx.x >> 6] = dia;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        atomicAdd(&out[1], part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}
// A <- the symmetric matrix given by its lower triangle; V <- identity
__global__ __launch_bounds__(256) void jacobi_init_kernel(int n, double *A, double *V, int ld)
{
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (i < j) A[(size_t)j * ld + i] = A[(size_t)i * ld + j];
        V[(size_t)j * ld + i] = (i == j) ? 1.0 : 0.0;
    }
}
__global__ void jacobi_diag_kernel(int n, const double *A, int ld, double *w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = A[(size_t)i * ld + i];
}


// ------------------------------------------------------------------ eigenvalues only: Householder tridiagonalisation + bisection
// jobz = 'N' (LAPACK's dsyev does the same: dsytrd, then the tridiagonal's eigenvalues).  ONE launch per Householder
// step: launch k applies the rank-2 update of step k-1 to the trailing columns, forms step k's reflector from the
// updated column k and, in the same pass over each column, the product p_k = tau_k T_k v_k that step k+1's update needs
// -- the trailing matrix is read and written once per step (16 m^2 bytes instead of 24 m^2 in three passes).  What is
// O(m) -- w_{k-1} = p - (tau/2)(p.v) v, the updated column k, its norm, v_k -- every workgroup forms for itself in LDS
// (two workgroup reductions), so nothing but the kernel boundary orders the steps.  A wave owns whole columns: the
// product's dot is a wave reduction, no barrier in the column loop.  The full symmetric matrix is kept (both triangles
// updated) so that every access is a contiguous column.
//   vbuf / pbuf [2][n]: reflector and product of the previous / this step (double buffered), scal [2][2]: tau
//   dyn. LDS: 3 n + 16 doubles (w_{k-1}, v_{k-1}, v_k, reduction scratch)
template <int BS> __device__ __forceinline__ double wg_sum(double v, double *scratch)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads(); // scratch free again
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < BS / 64; q += 4) t += (scratch[q] + scratch[q + 1]) + (scratch[q + 2] + scratch[q + 3]);
    return t;
}
// BS = 256 threads per workgroup, or 1024 for large n: the three vectors of a step fill the LDS of a CU from n ~ 3000 on
// (one workgroup per CU), and what bounds a step then is the bytes its waves keep in flight -- sixteen waves instead of four.
template <int BS>
__global__ __launch_bounds__(BS) void tridiag_step_kernel(int n, int k, double *A, int ld, const double *vprev,
                                                           const double *pprev, const double *tauprev, double *vnext,
                                                           double *pnext, double *taunext, double *dvec, double *evec,
                                                           double *tauvec, double *Vkeep, int ldk)
{
    extern __shared__ double tl[];
    double *w = tl, *vv = tl + n, *vn = tl + 2 * n, *scr = tl + 3 * n;
    const int tid = threadIdx.x, mp = n - k; // mp = length of the previous step's vectors (rows k .. n-1)
    const bool first = (k == 0);
    const double *colk = A + (size_t)k * ld;
    const int m = n - k - 1; // rows / columns of the trailing matrix T_k
    const int lane = tid & 63, gw = blockIdx.x * (BS / 64) + (tid >> 6), nwv = gridDim.x * (BS / 64);
    // 0. Everything this workgroup reads first -- its share of v_{k-1}, p_{k-1}, of column k and the first rows of its
    //    first trailing column -- is requested NOW, before anything waits: the step is a chain of dependent O(m) phases
    //    and one memory round trip instead of four is most of what a step costs (n = 1024: 7.5 -> see DESIGN.md 8).
    //    (Unconditional loads at clamped indices, used under the guards below: a load the compiler can make conditional
    //    turns the waits after it into vmcnt(0).)
    constexpr int PF = 4, PFA = 8;
    double pf_v[PF], pf_p[PF], pf_c[PF], pf_a[PFA];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int r = tid + BS * j;
        pf_v[j] = vprev[r < mp ? r : mp - 1];
        pf_p[j] = pprev[r < mp ? r : mp - 1];
        pf_c[j] = colk[k + 1 + (r < m ? r : m - 1)];
    }
    const double dk_raw = colk[k]; // (the diagonal entry block 0 reports below: requested with the rest, not after phase 2)
    {
        const double *col0 = A + (size_t)(k + 1 + (gw < m ? gw : m - 1)) * ld + (k + 1);
#pragma unroll
        for (int j = 0; j < PFA; ++j) pf_a[j] = col0[lane + 64 * j < m ? lane + 64 * j : m - 1];
    }
    // 1. w_{k-1} = p + alpha v, alpha = -tau/2 (p.v)   (dsytd2)
    double w0 = 0.0, v0 = 0.0;
    if (!first) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int r = tid + BS * j;
            if (r < mp) {
                vv[r] = pf_v[j];
                w[r] = pf_p[j]; // (p for now)
                s += pf_p[j] * pf_v[j];
            }
        }
        for (int r = tid + BS * PF; r < mp; r += BS) {
            const double vr = vprev[r], pr = pprev[r];
            vv[r] = vr;
            w[r] = pr;
            s += pr * vr;
        }
        s = wg_sum<BS>(s, scr);
        const double alpha = -0.5 * tauprev[0] * s;
        for (int r = tid; r < mp; r += BS) w[r] = w[r] + alpha * vv[r];
        __syncthreads();
        w0 = w[0];
        v0 = vv[0];
        // with eigenvectors wanted: reflector k-1 is kept in column k-1 of Vkeep (ldk >= n, zero-initialised by the caller:
        // rows above the reflector and the padding below row n read as zero, so whoever applies it needs no mask), as
        // u = sqrt(tau) v (tau is 0 or in [1, 2]): H = I - u u^T needs no second array
        if (tauvec && blockIdx.x == 0) {
            double *keep = Vkeep + (size_t)(k - 1) * ldk + k;
            const double st = sqrt(tauprev[0]);
            for (int r = tid; r < mp; r += BS) keep[r] = st * vv[r];
            if (tid == 0) tauvec[k - 1] = tauprev[0];
        }
    }
    // 2. column k after the update: d_k on the diagonal, x_k = rows k+1 .. n-1 below it -> reflector (dlarfg)
    double nrm2 = 0.0;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int r = tid + BS * j;
        if (r < m) {
            double x = pf_c[j];
            if (!first) x = x - (vv[r + 1] * w0 + w[r + 1] * v0);
            vn[r] = x;
            if (r > 0) nrm2 += x * x;
        }
    }
    for (int r = tid + BS * PF; r < m; r += BS) {
        double x = colk[k + 1 + r];
        if (!first) x = x - (vv[r + 1] * w0 + w[r + 1] * v0);
        vn[r] = x;
        if (r > 0) nrm2 += x * x;
    }
    nrm2 = wg_sum<BS>(nrm2, scr);
    const double alpha0 = vn[0];
    double tau = 0.0, beta = alpha0, scale = 0.0;
    if (nrm2 > 0.0) {
        const double nr = sqrt(alpha0 * alpha0 + nrm2);
        beta = alpha0 >= 0.0 ? -nr : nr;
        tau = (beta - alpha0) / beta;
        scale = 1.0 / (alpha0 - beta);
    }
    __syncthreads(); // every thread has read vn[0]
    for (int r = tid; r < m; r += BS) vn[r] = (r == 0) ? 1.0 : vn[r] * scale;
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int r = tid; r < m; r += BS) vnext[r] = vn[r];
        if (tid == 0) {
            double dk = dk_raw;
            if (!first) dk = dk - 2.0 * (v0 * w0);
            dvec[k] = dk;
            evec[k] = beta;
            taunext[0] = tau;
        }
    }
    // 3. the trailing columns: update of step k-1, stored, and their share of p_k = tau_k T_k v_k
    for (int c = gw; c < m; c += nwv) {
        double *col = A + (size_t)(k + 1 + c) * ld + (k + 1);
        const double wj = first ? 0.0 : w[c + 1], vj = first ? 0.0 : vv[c + 1];
        double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
        int r = lane;
        if (c == gw) { // the rows requested at the top of the kernel
#pragma unroll
            for (int j = 0; j < PFA; ++j) {
                const int rr = lane + 64 * j;
                if (rr < m) {
                    double a = pf_a[j];
                    if (!first) {
                        a = a - (vv[rr + 1] * wj + w[rr + 1] * vj);
                        col[rr] = a;
                    }
                    if (j & 1) acc1 += a * vn[rr];
                    else acc += a * vn[rr];
                }
            }
            r = lane + 64 * PFA;
        }
        for (; r + 192 < m; r += 256) { // four independent rows in flight per lane
            double a0 = col[r], a1 = col[r + 64], a2 = col[r + 128], a3 = col[r + 192];
            if (!first) {
                a0 = a0 - (vv[r + 1] * wj + w[r + 1] * vj);
                a1 = a1 - (vv[r + 65] * wj + w[r + 65] * vj);
                a2 = a2 - (vv[r + 129] * wj + w[r + 129] * vj);
                a3 = a3 - (vv[r + 193] * wj + w[r + 193] * vj);
                col[r] = a0;
                col[r + 64] = a1;
                col[r + 128] = a2;
                col[r + 192] = a3;
            }
            acc += a0 * vn[r];
            acc1 += a1 * vn[r + 64];
            acc2 += a2 * vn[r + 128];
            acc3 += a3 * vn[r + 192];
        }
        for (; r < m; r += 64) {
            double a = col[r];
            if (!first) {
                a = a - (vv[r + 1] * wj + w[r + 1] * vj);
                col[r] = a;
            }
            acc += a * vn[r];
        }
        acc = (acc + acc1) + (acc2 + acc3);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) pnext[c] = tau * acc;
    }
}
// lower triangle -> full symmetric matrix
__global__ __launch_bounds__(256) void symmetrize_kernel(int n, double *A, int ld)
{
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < j; i += 256) A[(size_t)j * ld + i] = A[(size_t)i * ld + j];
}
// Eigenvalues of the symmetric tridiagonal (d, e) by multisection with Sturm counts (LAPACK's dstebz counts the same
// way): wave i finds eigenvalue i -- 64 sample points per pass, every lane one Sturm sequence, the bracket shrinks
// 65-fold per pass.  count(x) = number of eigenvalues < x.  grid ceil(n / 4), block 256, dyn. LDS 2 n doubles.
__global__ __launch_bounds__(256) void sturm_multisection_kernel(int n, const double *dvec, const double *evec,
                                                                 const double *last_diag, double *wout)
{
    extern __shared__ double tl[];
    double *d = tl, *e2 = tl + n;
    __shared__ double red[3][4];
    const int tid = threadIdx.x, lane = tid & 63;
    double gl = 1e308, gu = -1e308, emax = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double di = (i == n - 1) ? last_diag[0] : dvec[i];
        const double el = (i > 0) ? fabs(evec[i - 1]) : 0.0, er = (i < n - 1) ? fabs(evec[i]) : 0.0;
        d[i] = di;
        e2[i] = er * er; // e2[i] couples i and i+1
        gl = fmin(gl, di - el - er);
        gu = fmax(gu, di + el + er);
        emax = fmax(emax, er * er);
    }
    for (int o = 32; o > 0; o >>= 1) {
        gl = fmin(gl, __shfl_xor(gl, o));
        gu = fmax(gu, __shfl_xor(gu, o));
        emax = fmax(emax, __shfl_xor(emax, o));
    }
    if (lane == 0) {
        red[0][tid >> 6] = gl;
        red[1][tid >> 6] = gu;
        red[2][tid >> 6] = emax;
    }
    __syncthreads();
    gl = fmin(fmin(red[0][0], red[0][1]), fmin(red[0][2], red[0][3]));
    gu = fmax(fmax(red[1][0], red[1][1]), fmax(red[1][2], red[1][3]));
    emax = fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    const double tnorm = fmax(fabs(gl), fabs(gu));
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax);
    const int idx0 = blockIdx.x * 4 + (tid >> 6);
    if (gl == gu) { // a multiple of the identity (the zero matrix among them): exact
        if (lane == 0 && idx0 < n) wout[idx0] = gl;
        return;
    }
    // widen the Gershgorin interval a little so that both ends are strict bounds in floating point
    gl = gl - 2.0 * tnorm * 2.220446049250313e-16 * n - 2.0 * pivmin;
    gu = gu + 2.0 * tnorm * 2.220446049250313e-16 * n + 2.0 * pivmin;
    const int idx = blockIdx.x * 4 + (tid >> 6); // this wave's eigenvalue (ascending, 0-based)
    if (idx >= n) return;
    double lo = gl, hi = gu;
    for (int pass = 0; pass < 24; ++pass) {
        const double width = hi - lo;
        if (width <= 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + pivmin) break;
        const double x = lo + width * ((double)(lane + 1) / 65.0);
        double q = d[0] - x;
        if (fabs(q) < pivmin) q = -pivmin;
        int cnt = q < 0.0 ? 1 : 0;
        for (int i = 1; i < n; ++i) {
            q = d[i] - x - e2[i - 1] / q;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0 ? 1 : 0;
        }
        // lanes whose sample has at most idx eigenvalues below it lie left of eigenvalue idx (counts are monotone in
        // exact arithmetic; the popcount is used, which is robust to a non-monotone pair)
        const unsigned long long below = __ballot(cnt <= idx);
        const int t = __popcll(below);
        const double nlo = (t == 0) ? lo : __shfl(x, t - 1), nhi = (t == 64) ? hi : __shfl(x, t);
        lo = nlo;
        hi = nhi;
    }
    if (lane == 0) wout[idx] = 0.5 * (lo + hi);
}

} // namespace fl

extern "C" {

// launcher shared by fl_dgemm and the blocked Cholesky routines (fl_chol_blocked.hip)
int fl_dgemm_strided(int transA, int transB, int M, int K, int N, double alpha, const double *A_dev, int lda, size_t strideA,
                     const double *B_dev, int ldb, size_t strideB, double beta, double *C_dev, int ldc, size_t strideC,
                     int batch, int lower_only, void *stream)
{
    if (!A_dev || !B_dev || !C_dev || M <= 0 || N <= 0 || K < 0 || batch <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (lda < (transA ? K : M) || ldb < (transB ? N : K) || ldc < M) return FL_ERR_INVALID_ARGUMENT;
    fl::GemmArgs g;
    g.transA = transA ? 1 : 0;
    g.transB = transB ? 1 : 0;
    g.M = M; g.K = K; g.N = N; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.lower = lower_only ? 1 : 0;
    g.alpha = alpha; g.beta = beta;
    g.A = A_dev; g.B = B_dev; g.C = C_dev;
    g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
    const int tiles128 = ((M + 127) / 128) * ((N + 127) / 128), tiles64 = ((M + 63) / 64) * ((N + 63) / 64);
    const bool big = (long long)tiles128 * batch >= 256; // at least one 128-tile per CU
    // the matrix index is gridDim.y (at most 65535): longer batches go in chunks
    for (int b0 = 0; b0 < batch; b0 += FL_GRID_YZ_MAX) {
        const int nb = batch - b0 < FL_GRID_YZ_MAX ? batch - b0 : FL_GRID_YZ_MAX;
        fl::GemmArgs gc = g;
        gc.A = g.A + (size_t)b0 * strideA;
        gc.B = g.B + (size_t)b0 * strideB;
        gc.C = g.C + (size_t)b0 * strideC;
        const int tiles32 = ((M + 31) / 32) * ((N + 31) / 32);
        hipStream_t st_ = static_cast<hipStream_t>(stream);
        if (big)
            hipLaunchKernelGGL((fl::dgemm_kernel<128, 2, 4>), dim3(tiles128, nb), dim3(512), 0, st_, gc);
        else if ((long long)tiles64 * nb >= 512) // (n = 1536: 227 us here, 312 with the small-problem kernel on 64-tiles)
            hipLaunchKernelGGL((fl::dgemm_kernel<64, 2, 2>), dim3(tiles64, nb), dim3(256), 0, st_, gc);
        else // fewer than two 64-tiles per CU: 32-tiles, operand loads eight k-blocks ahead (dgemm_small_kernel)
            hipLaunchKernelGGL((fl::dgemm_small_kernel<32, 1, 2, 8>), dim3(tiles32, nb), dim3(128), 0, st_, gc);
    }
    return fl::launch_status();
}

int fl_dgemm(int transA, int M, int K, int N, const double *A_dev, int lda, const double *B_dev, int ldb, double *C_dev,
             int ldc, void *stream)
{
    if (K <= 0) return FL_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    return fl_dgemm_strided(transA, 0, M, K, N, 1.0, A_dev, lda, 0, B_dev, ldb, 0, 0.0, C_dev, ldc, 0, 1, 0, stream);
}

size_t fl_dsyev_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    const size_t n2 = (size_t)n + (n & 1);
    return ((size_t)2 * n * n + 2 * n2 + 4) * sizeof(double); // A_t, V, rotations of two steps, norms
}

// A_dev: n x n column-major (lda), lower triangle referenced; on return the diagonal of the rotated matrix is in w_dev
// (UNSORTED) and, for jobz = 'V', the matching eigenvectors are the columns of V = workspace + n*n doubles (ld n).
// The caller sorts (fl_linalg.cpp does, on the host, where the legacy interface wants its arrays anyway).
// *sweeps_out (host) = sweeps used, negative if max_sweeps did not reach the tolerance.
int fl_dsyev_jacobi(char jobz, int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes,
                    int max_sweeps, int *sweeps_out, void *stream)
{
    if (!A_dev || !w_dev || n <= 0 || lda < n) return FL_ERR_INVALID_ARGUMENT;
    if (n > 8192) return FL_ERR_UNSUPPORTED_SIZE; // a column is staged in LDS
    if (!workspace_dev || workspace_bytes < fl_dsyev_workspace_bytes(n)) return FL_ERR_WORKSPACE;
    if (lda != n) return FL_ERR_INVALID_ARGUMENT;  // packed leading dimension (the legacy symbol's layout)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int want = (jobz == 'V' || jobz == 'v') ? 1 : 0;
    const int n2 = n + (n & 1);
    double *Am = static_cast<double *>(workspace_dev), *V = Am + (size_t)n * n, *cs = V + (size_t)n * n, *nrm = cs + 2 * n2;
    const size_t lds_step = (size_t)2 * n * sizeof(double);
    if (lds_step > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(fl::jacobi_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_step) != hipSuccess)
        return FL_ERR_LAUNCH;
    hipLaunchKernelGGL(fl::jacobi_init_kernel, dim3(n), dim3(256), 0, st, n, A_dev, V, n);
    int sweeps = 0, prev = -1, cur = 0;
    bool done = (n == 1);
    double h[2], off_prev = -1.0;
    if (done && hipMemcpyAsync(Am, A_dev, sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) return FL_ERR_LAUNCH;
    while (!done && sweeps < max_sweeps) {
        // A_dev holds B_t (the column-rotated matrix whose row rotations, cs[cur ^ 1], are still pending)
        for (int step = 0; step < n2 - 1; ++step) {
            hipLaunchKernelGGL(fl::jacobi_step_kernel, dim3(n2 / 2), dim3(256), lds_step, st, n, n2, prev, step, A_dev, V, n,
                               cs + (size_t)(cur ^ 1) * n2, cs + (size_t)cur * n2, want);
            prev = step;
            cur ^= 1;
        }
        ++sweeps;
        // A_t = J_t^T B_t, materialised beside B_t for the convergence test (and the final diagonal)
        hipLaunchKernelGGL(fl::jacobi_rows_kernel, dim3(n), dim3(256), (size_t)n * sizeof(double), st, n, n2, prev, A_dev, Am,
                           n, cs + (size_t)(cur ^ 1) * n2);
        if (hipMemsetAsync(nrm, 0, 2 * sizeof(double), st) != hipSuccess) return FL_ERR_LAUNCH;
        hipLaunchKernelGGL(fl::jacobi_norms_kernel, dim3(n), dim3(256), 0, st, n, Am, n, nrm);
        if (hipMemcpyAsync(h, nrm, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return FL_ERR_LAUNCH;
        // off(A) <= n eps ||A||_F -- the level rounding keeps the off-diagonal part at (every element is rotated ~n times
        // per sweep) -- or off(A) has stopped shrinking below 1e-12 ||A||_F (quadratic convergence has hit that floor).
        // Every eigenvalue is then within off(A) of a diagonal entry.
        const double off2 = 2.0 * h[0], fro2 = 2.0 * h[0] + h[1], tol = (n > 4 ? n : 4) * 2.2e-16;
        done = off2 <= tol * tol * fro2 || (off_prev >= 0.0 && off2 <= 1e-24 * fro2 && off2 > 0.25 * off_prev);
        off_prev = off2;
    }
    if (sweeps == 0 && n > 1 && // max_sweeps <= 0: the diagonal of the matrix as given
        hipMemcpyAsync(Am, A_dev, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return FL_ERR_LAUNCH;
    hipLaunchKernelGGL(fl::jacobi_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, Am, n, w_dev);
    if (sweeps_out) *sweeps_out = done ? sweeps : -sweeps;
    return fl::launch_status();
}


// Householder tridiagonalisation in n - 1 launches, then the tridiagonal's eigenvalues by multisection: the shared front of
// fl_dsyev_values and fl_dsyev_vectors (fl_eig_vectors.hip).  ws: (6 n + 4) doubles; on return d = ws + 4 n + 4 (n - 1
// entries; the last diagonal entry stays at A(n-1, n-1)), e = d + n.  tauvec (n doubles) non-NULL: reflector k is kept in
// rows k+1 .. n-1 of column k of Vkeep (ldk >= n, zeroed by the caller) as u_k = sqrt(tau_k) v_k (H_k = I - u_k u_k^T),
// tau_k in tauvec[k], k < n - 2.
int fl_sytrd_values(int n, double *A_dev, double *w_dev, double *ws, double *tauvec, double *Vkeep, int ldk, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *vbuf = ws, *pbuf = ws + 2 * (size_t)n, *tau = ws + 4 * (size_t)n, *dvec = tau + 4, *evec = dvec + n;
    const size_t lds_step = ((size_t)3 * n + 16) * sizeof(double), lds_sturm = (size_t)2 * n * sizeof(double);
    const bool wide = n >= 3072; // (see tridiag_step_kernel; n = 2048: 22.5 ms with 256 threads, 28.4 with 1024; n = 4096: 134 / 110)
    if (lds_step > 48 * 1024 &&
        hipFuncSetAttribute(wide ? reinterpret_cast<const void *>(fl::tridiag_step_kernel<1024>) : reinterpret_cast<const void *>(fl::tridiag_step_kernel<256>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_step) != hipSuccess)
        return FL_ERR_LAUNCH;
    if (lds_sturm > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(fl::sturm_multisection_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sturm) != hipSuccess)
        return FL_ERR_LAUNCH;
    hipLaunchKernelGGL(fl::symmetrize_kernel, dim3(n), dim3(256), 0, st, n, A_dev, n);
    for (int k = 0; k + 1 < n; ++k) {
        const int m = n - k - 1, cur = k & 1, prv = cur ^ 1;
        if (wide) {
            int wgs = (m + 15) / 16; // a column per wave
            wgs = wgs < 1 ? 1 : (wgs > 512 ? 512 : wgs);
            hipLaunchKernelGGL(fl::tridiag_step_kernel<1024>, dim3(wgs), dim3(1024), lds_step, st, n, k, A_dev, n, vbuf + (size_t)prv * n,
                               pbuf + (size_t)prv * n, tau + prv, vbuf + (size_t)cur * n, pbuf + (size_t)cur * n, tau + cur, dvec, evec,
                               tauvec, Vkeep, ldk);
        } else {
            int wgs = (m + 3) / 4; // a column per wave, at most two workgroups per CU (every workgroup repeats the O(m) part)
            wgs = wgs < 1 ? 1 : (wgs > 512 ? 512 : wgs);
            hipLaunchKernelGGL(fl::tridiag_step_kernel<256>, dim3(wgs), dim3(256), lds_step, st, n, k, A_dev, n, vbuf + (size_t)prv * n,
                               pbuf + (size_t)prv * n, tau + prv, vbuf + (size_t)cur * n, pbuf + (size_t)cur * n, tau + cur, dvec, evec,
                               tauvec, Vkeep, ldk);
        }
    }
    // the last diagonal entry took its final update in the last step's column pass (n = 1: the matrix itself)
    hipLaunchKernelGGL(fl::sturm_multisection_kernel, dim3((n + 3) / 4), dim3(256), lds_sturm, st, n, dvec, evec,
                       A_dev + (size_t)(n - 1) * n + (n - 1), w_dev);
    return fl::launch_status();
}

// Eigenvalues only (jobz = 'N'): A_dev (n x n, lda = n, lower triangle referenced) is destroyed; w_dev: the eigenvalues
// in ascending order.  n <= 6144 (three vectors of the step live in LDS); workspace as for fl_dsyev_jacobi.
int fl_dsyev_values(int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (!A_dev || !w_dev || n <= 0 || lda != n) return FL_ERR_INVALID_ARGUMENT;
    if (n > 6144) return FL_ERR_UNSUPPORTED_SIZE;
    if (!workspace_dev || workspace_bytes < fl_dsyev_workspace_bytes(n)) return FL_ERR_WORKSPACE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    return fl_sytrd_values(n, A_dev, w_dev, static_cast<double *>(workspace_dev), nullptr, nullptr, 0, stream);
}

} // extern "C"
