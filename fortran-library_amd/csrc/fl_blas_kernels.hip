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
// DSYEV: cyclic two-sided Jacobi with the round-robin (tournament) ordering: a step applies n/2 disjoint rotations at
// once, A <- J^T (A J), V <- V J; a sweep is n - 1 steps.  Two launches per step:
//   jacobi_cols: B = A J (+ V = V J in place): one thread per (row, pair); every thread forms its pair's rotation from
//                a_pp, a_qq, a_pq of the unmodified A (reads only) and the rotation is kept for the second launch;
//   jacobi_rows: A = J^T B: one workgroup per column, which is staged in LDS so that global accesses stay contiguous.
// Converged when the off-diagonal norm is below eps * the Frobenius norm (checked once per sweep).  Jacobi's
// eigenvalues are at least as accurate as the QR iteration's behind LAPACK's dsyev; eigenvectors are defined up to
// sign, like LAPACK's.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// B = A J, V = V J (in place); cs[2k], cs[2k+1] = rotation of pair k.  grid (ceil(n/256), n2/2), block 256
__global__ __launch_bounds__(256) void jacobi_cols_kernel(int n, int n2, int step, const double *A, double *Bm, double *V,
                                                          int ld, double *cs, int want_vectors)
{
    const int k = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    int p, q;
    jacobi_pair(n2, step, k, p, q);
    if (q >= n) { // the padding player of an odd n: its partner's column is copied unchanged
        if (i < n && p < n) Bm[(size_t)p * ld + i] = A[(size_t)p * ld + i];
        if (i == 0) {
            cs[2 * k] = 1.0;
            cs[2 * k + 1] = 0.0;
        }
        return;
    }
    double c, s;
    jacobi_rotation(A[(size_t)p * ld + p], A[(size_t)q * ld + q], A[(size_t)q * ld + p], c, s);
    if (i == 0) {
        cs[2 * k] = c;
        cs[2 * k + 1] = s;
    }
    if (i >= n) return;
    const double ap = A[(size_t)p * ld + i], aq = A[(size_t)q * ld + i];
    Bm[(size_t)p * ld + i] = c * ap - s * aq;
    Bm[(size_t)q * ld + i] = s * ap + c * aq;
    if (want_vectors) {
        const double vp = V[(size_t)p * ld + i], vq = V[(size_t)q * ld + i];
        V[(size_t)p * ld + i] = c * vp - s * vq;
        V[(size_t)q * ld + i] = s * vp + c * vq;
    }
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
    if ((long long)tiles128 * batch >= 256) // at least one 128-tile per CU
        hipLaunchKernelGGL((fl::dgemm_kernel<128, 2, 4>), dim3(tiles128, batch), dim3(512), 0,
                           static_cast<hipStream_t>(stream), g);
    else
        hipLaunchKernelGGL((fl::dgemm_kernel<64, 2, 2>), dim3(tiles64, batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                           g);
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
    return ((size_t)2 * n * n + n2 + 4) * sizeof(double); // B, V, rotations, norms
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
    double *Bm = static_cast<double *>(workspace_dev), *V = Bm + (size_t)n * n, *cs = V + (size_t)n * n, *nrm = cs + n2;
    hipLaunchKernelGGL(fl::jacobi_init_kernel, dim3(n), dim3(256), 0, st, n, A_dev, V, n);
    int sweeps = 0;
    bool done = (n == 1);
    double h[2], off_prev = -1.0;
    while (!done && sweeps < max_sweeps) {
        for (int step = 0; step < n2 - 1; ++step) {
            hipLaunchKernelGGL(fl::jacobi_cols_kernel, dim3((n + 255) / 256, n2 / 2), dim3(256), 0, st, n, n2, step, A_dev, Bm,
                               V, n, cs, want);
            hipLaunchKernelGGL(fl::jacobi_rows_kernel, dim3(n), dim3(256), (size_t)n * sizeof(double), st, n, n2, step, Bm,
                               A_dev, n, cs);
        }
        ++sweeps;
        if (hipMemsetAsync(nrm, 0, 2 * sizeof(double), st) != hipSuccess) return FL_ERR_LAUNCH;
        hipLaunchKernelGGL(fl::jacobi_norms_kernel, dim3(n), dim3(256), 0, st, n, A_dev, n, nrm);
        if (hipMemcpyAsync(h, nrm, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return FL_ERR_LAUNCH;
        // off(A) <= n eps ||A||_F -- the level rounding keeps the off-diagonal part at (every element is rotated ~n times
        // per sweep) -- or off(A) has stopped shrinking below 1e-12 ||A||_F (quadratic convergence has hit that floor).
        // Every eigenvalue is then within off(A) of a diagonal entry.
        const double off2 = 2.0 * h[0], fro2 = 2.0 * h[0] + h[1], tol = (n > 4 ? n : 4) * 2.2e-16;
        done = off2 <= tol * tol * fro2 || (off_prev >= 0.0 && off2 <= 1e-24 * fro2 && off2 > 0.25 * off_prev);
        off_prev = off2;
    }
    hipLaunchKernelGGL(fl::jacobi_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, A_dev, n, w_dev);
    if (sweeps_out) *sweeps_out = done ? sweeps : -sweeps;
    return fl::launch_status();
}

} // extern "C"
