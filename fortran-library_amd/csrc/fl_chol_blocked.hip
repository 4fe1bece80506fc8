// fl_chol_blocked.hip -- My_dposv / My_dpotri (LinearAlgebra.f90:719-730, 798-812 + dsyL2U 260-265) for LARGE matrices:
// blocked right-looking Cholesky whose O(n^3) part runs on the f64 matrix cores through dgemm_kernel
// (fl_blas_kernels.hip), many workgroups per matrix.  fl_dposv_batched / fl_dpotri_batched take this path from
// n = 512 on (and beyond n = 4096, where the one-workgroup kernels of fl_dense.hpp do not exist at all); below they
// keep the sequential-order kernels, whose sums the oracle replays bit for bit (the solvers' own Cholesky steps --
// NewtonRaphson, the exact-Hessian refresh of BFGS -- always do, inside the fused kernel).  Here the summation order is the MFMA's,
// so parity is to LAPACK rounding (tests: the reference's own results in tests/golden/la_ref.npz at n = 1024 through
// FL_CHOL_BLOCKED_MIN_N, numpy beyond).
//
// Per block column k (NB = 64 columns):
//   1. chol_diag_kernel: L11 = chol(A11) in LDS (one workgroup per matrix), and W11 = L11^{-1} next to it;
//   2. panel: P = A21 W11^T (dgemm 'N','T': the triangular solve as a product with the inverted block), copied over A21;
//   3. trailing update A22 -= P P^T (dgemm 'N','T', alpha = -1, beta = 1, lower tiles only).
// Solve (one right-hand side): forward y_k = W_k (b_k - L_k,0:k y_0:k), backward x_k = W_k^T (y_k - L_k+1:,k^T x_k+1:)
// -- the products with the W_k again dgemm calls (N = 1), O(n^2) in all.
// Inverse: X = L^{-1} by the same forward recurrence on the identity (row blocks of X), then A^{-1} = X^T X
// (dgemm 'T','N'), both triangles written -- which is My_dpotri followed by dsyL2U.
// LAPACK's info: index of the first non-positive pivot; fl_dposv_blocked leaves that matrix as the failure found it
// (block columns before the failing one factorised, the failing diagonal block partially, behind it the Schur
// complement of the completed block steps) and b untouched; fl_dpotri_blocked leaves A unspecified where info != 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"

namespace fl {

constexpr int CNB = 64;

// A11 (nb x nb at A + k0 + k0*lda) -> its lower Cholesky factor in place; W (CNB x CNB, ld CNB, zero above the
// diagonal, identity on the padding) <- L11^{-1}.  info[mat]: untouched if positive definite so far, else k0 + j + 1.
// grid = batch, block = 256.
__global__ __launch_bounds__(256) void chol_diag_kernel(int n, int k0, double *A_all, int lda, size_t strideA, double *W_all,
                                                        size_t strideW, int32_t *info)
{
    __shared__ double S[CNB][CNB + 1];
    __shared__ double Wt[CNB][CNB + 1];
    __shared__ int bad;
    const int mat = blockIdx.x, tid = threadIdx.x;
    double *A = A_all + (size_t)mat * strideA + (size_t)k0 * lda + k0;
    double *W = W_all + (size_t)mat * strideW + (size_t)(k0 / CNB) * CNB * CNB;
    if (info[mat] != 0) { // an earlier block failed: the reference stops there too.  W = 0 makes this block column's
                          // panel product P = A21 W^T vanish, so the trailing update subtracts nothing; the copy of P
                          // over A21 is skipped for this matrix (copy_block_kernel): A stays as the failure left it
        for (int e = tid; e < CNB * CNB; e += 256) W[e] = 0.0;
        return;
    }
    const int nb = (n - k0 < CNB) ? n - k0 : CNB;
    for (int e = tid; e < CNB * CNB; e += 256) {
        const int c = e / CNB, r = e - c * CNB;
        S[c][r] = (r < nb && c < nb && r >= c) ? A[(size_t)c * lda + r] : ((r == c) ? 1.0 : 0.0); // S[col][row], lower
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double piv = S[j][j];
        if (!(piv > 0.0)) {
            if (tid == 0) {
                bad = j + 1;
                info[mat] = k0 + j + 1;
            }
            break; // uniform: every thread reads the same S[j][j]
        }
        const double ajj = sqrt(piv);
        __syncthreads();
        if (tid < CNB && tid >= j) S[j][tid] = (tid == j) ? ajj : S[j][tid] / ajj;
        __syncthreads();
        // trailing columns c > j of the block: S[c][r] -= L[r][j] L[c][j], r >= c (a thread keeps its row, four columns
        // per pass: no index arithmetic in the loop)
        {
            const int r = tid & (CNB - 1);
            const double lrj = S[j][r];
            for (int c = j + 1 + (tid >> 6); c < nb; c += 4)
                if (r >= c) S[c][r] = S[c][r] - lrj * S[j][c];
        }
        __syncthreads();
    }
    __syncthreads();
    // the factorised part goes back even when a pivot failed ("A will be overwritten even fail", LA.f90:717)
    for (int e = tid; e < CNB * CNB; e += 256) {
        const int c = e / CNB, r = e - c * CNB;
        if (r < nb && c < nb && r >= c) A[(size_t)c * lda + r] = S[c][r];
    }
    if (bad) { // (see above: no panel, no trailing update for this matrix from here on)
        for (int e = tid; e < CNB * CNB; e += 256) W[e] = 0.0;
        return;
    }
    // W = L^{-1} by doubling: the inverse of [A 0; B C] is [A^-1 0; -C^-1 B A^-1  C^-1].  Diagonal 1 x 1 blocks first,
    // then pairs of adjacent s x s blocks for s = 1, 2, 4, ..., 32: T = B A^-1 and X = -C^-1 T are s x s products over
    // the whole workgroup (a thread-per-column substitution would be a 2000-step dependent chain of LDS reads).
    for (int e = tid; e < CNB * CNB; e += 256) {
        const int c = e / CNB, r = e - c * CNB;
        Wt[c][r] = (r == c) ? 1.0 / S[c][c] : 0.0;
    }
    __syncthreads();
    __shared__ double Tm[CNB][CNB / 2 + 1]; // T = B A^-1 of every pair of the level: Tm[row of B][column within the pair]
    for (int sz = 1, lg = 0; sz < CNB; sz *= 2, ++lg) {
        // pair p covers rows/cols [2 p sz, 2 p sz + 2 sz): A = first half, C = second half, B = L(second, first)
        const int npairs = CNB / (2 * sz), per = sz * sz;
        for (int e = tid; e < npairs * per; e += 256) {
            const int pr = e >> (2 * lg), q = e & (per - 1), rr = q >> lg, cc = q & (sz - 1);
            const int a0 = 2 * pr * sz, c0 = a0 + sz;
            double t = 0.0;
            for (int k = cc; k < sz; ++k) t += S[a0 + k][c0 + rr] * Wt[a0 + cc][a0 + k]; // B(rr,k) * Ainv(k,cc)
            Tm[c0 + rr][cc] = t;
        }
        __syncthreads();
        for (int e = tid; e < npairs * per; e += 256) {
            const int pr = e >> (2 * lg), q = e & (per - 1), rr = q >> lg, cc = q & (sz - 1);
            const int a0 = 2 * pr * sz, c0 = a0 + sz;
            double t = 0.0;
            for (int k = 0; k <= rr; ++k) t += Wt[c0 + k][c0 + rr] * Tm[c0 + k][cc]; // Cinv(rr,k) * T(k,cc)
            Wt[a0 + cc][c0 + rr] = -t;
        }
        __syncthreads();
    }
    for (int e = tid; e < CNB * CNB; e += 256) {
        const int c = e / CNB, r = e - c * CNB;
        W[(size_t)c * CNB + r] = Wt[c][r];
    }
}

// dst(rows x cols, ldd) = src(rows x cols, lds) for every matrix of a strided batch
// (skip: per-matrix flags, e.g. info -- a matrix whose flag is non-zero is left alone; may be NULL)
__global__ __launch_bounds__(256) void copy_block_kernel(int rows, int cols, const double *src, int lds_, size_t strideS,
                                                         double *dst, int ldd, size_t strideD, const int32_t *skip)
{
    if (skip && skip[blockIdx.z] != 0) return;
    const double *s = src + (size_t)blockIdx.z * strideS;
    double *d = dst + (size_t)blockIdx.z * strideD;
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < rows && c < cols) d[(size_t)c * ldd + r] = s[(size_t)c * lds_ + r];
}
// the matrix index is gridDim.z / gridDim.y (at most 65535): longer batches go in chunks
static void copy_blocks(int batch, hipStream_t st, int rows, int cols, const double *src, int lds_, size_t strideS, double *dst, int ldd,
                        size_t strideD, const int32_t *skip)
{
    for (int b0 = 0; b0 < batch; b0 += FL_GRID_YZ_MAX) {
        const int nb = batch - b0 < FL_GRID_YZ_MAX ? batch - b0 : FL_GRID_YZ_MAX;
        hipLaunchKernelGGL(copy_block_kernel, dim3((rows + 255) / 256, cols, nb), dim3(256), 0, st, rows, cols, src + (size_t)b0 * strideS,
                           lds_, strideS, dst + (size_t)b0 * strideD, ldd, strideD, skip ? skip + b0 : nullptr);
    }
}
// X(n x n, ld) = 0 with a unit diagonal block written where asked (rows k0..k0+nb of the identity)
__global__ __launch_bounds__(256) void fill_kernel(int rows, int cols, double *X, int ld, size_t stride, int diag_row0)
{
    double *x = X + (size_t)blockIdx.z * stride;
    const int r = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (r < rows && c < cols) x[(size_t)c * ld + r] = (diag_row0 >= 0 && c == diag_row0 + r) ? 1.0 : 0.0;
}
// b <- t where the factorisation succeeded (b untouched on failure: My_dposv, LA.f90:718)
__global__ __launch_bounds__(256) void commit_rhs_kernel(int n, const double *t, double *b, const int32_t *info)
{
    const int mat = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && info[mat] == 0) b[(size_t)mat * n + i] = t[(size_t)mat * n + i];
}

struct Chol {
    int batch, n, lda, nblk;
    double *A, *W, *P; // W: [batch][nblk][CNB][CNB] inverted diagonal blocks; P: [batch][n][CNB] panel
    size_t strideA, strideW, strideP;
    int32_t *info;
    hipStream_t st;
};

static int gemm(const Chol &c, int tA, int tB, int M, int K, int N, double alpha, const double *A, int lda, size_t sA,
                const double *B, int ldb, size_t sB, double beta, double *Cm, int ldc, size_t sC, int lower)
{
    if (M <= 0 || N <= 0 || K <= 0) return FL_OK;
    return fl_dgemm_strided(tA, tB, M, K, N, alpha, A, lda, sA, B, ldb, sB, beta, Cm, ldc, sC, c.batch, lower, c.st);
}

// A = L L^T in place (lower), the inverted diagonal blocks in c.W
static int potrf(const Chol &c)
{
    const int n = c.n;
    for (int k0 = 0; k0 < n; k0 += CNB) {
        const int nb = (n - k0 < CNB) ? n - k0 : CNB, rest = n - k0 - nb;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(c.batch), dim3(256), 0, c.st, n, k0, c.A, c.lda, c.strideA, c.W, c.strideW,
                           c.info);
        if (rest <= 0) break;
        const double *A21 = c.A + (size_t)k0 * c.lda + k0 + nb;
        const double *Wk = c.W + (size_t)(k0 / CNB) * CNB * CNB;
        // P = A21 W11^T   (rest x nb): B operand given as N x K = W itself (element (k, n) of W^T is W(n, k))
        int rc = gemm(c, 0, 1, rest, nb, nb, 1.0, A21, c.lda, c.strideA, Wk, CNB, c.strideW, 0.0, c.P, n, c.strideP, 0);
        if (rc != FL_OK) return rc;
        copy_blocks(c.batch, c.st, rest, nb, c.P, n, c.strideP, const_cast<double *>(A21), c.lda, c.strideA, c.info);
        // A22 -= P P^T (lower tiles)
        double *A22 = c.A + (size_t)(k0 + nb) * c.lda + k0 + nb;
        rc = gemm(c, 0, 1, rest, nb, rest, -1.0, c.P, n, c.strideP, c.P, n, c.strideP, 1.0, A22, c.lda, c.strideA, 1);
        if (rc != FL_OK) return rc;
    }
    return launch_status();
}

// T (n x nrhs per matrix, ld n) <- L^{-1} T, block rows top down: T_k = W_k (T_k - L_k,0:k0 T_0:k0)
static int forward(const Chol &c, double *T, int nrhs, size_t strideT, double *tmp, size_t strideTmp)
{
    const int n = c.n;
    for (int k0 = 0; k0 < n; k0 += CNB) {
        const int nb = (n - k0 < CNB) ? n - k0 : CNB;
        if (k0 > 0) { // T_k -= L(k, 0:k0) T(0:k0, :)
            int rc = gemm(c, 0, 0, nb, k0, nrhs, -1.0, c.A + k0, c.lda, c.strideA, T, n, strideT, 1.0, T + k0, n, strideT, 0);
            if (rc != FL_OK) return rc;
        }
        // T_k = W_k T_k (through tmp: a product cannot overwrite its own operand)
        const double *Wk = c.W + (size_t)(k0 / CNB) * CNB * CNB;
        int rc = gemm(c, 0, 0, nb, nb, nrhs, 1.0, Wk, CNB, c.strideW, T + k0, n, strideT, 0.0, tmp, CNB, strideTmp, 0);
        if (rc != FL_OK) return rc;
        copy_blocks(c.batch, c.st, nb, nrhs, tmp, CNB, strideTmp, T + k0, n, strideT, nullptr);
    }
    return launch_status();
}
// T <- L^{-T} T, block rows bottom up: T_k = W_k^T (T_k - L(k0+nb:, k)^T T(k0+nb:, :))
static int backward(const Chol &c, double *T, int nrhs, size_t strideT, double *tmp, size_t strideTmp)
{
    const int n = c.n;
    for (int k0 = ((n - 1) / CNB) * CNB; k0 >= 0; k0 -= CNB) {
        const int nb = (n - k0 < CNB) ? n - k0 : CNB, rest = n - k0 - nb;
        if (rest > 0) {
            int rc = gemm(c, 1, 0, nb, rest, nrhs, -1.0, c.A + (size_t)k0 * c.lda + k0 + nb, c.lda, c.strideA, T + k0 + nb, n,
                          strideT, 1.0, T + k0, n, strideT, 0);
            if (rc != FL_OK) return rc;
        }
        const double *Wk = c.W + (size_t)(k0 / CNB) * CNB * CNB;
        int rc = gemm(c, 1, 0, nb, nb, nrhs, 1.0, Wk, CNB, c.strideW, T + k0, n, strideT, 0.0, tmp, CNB, strideTmp, 0);
        if (rc != FL_OK) return rc;
        copy_blocks(c.batch, c.st, nb, nrhs, tmp, CNB, strideTmp, T + k0, n, strideT, nullptr);
    }
    return launch_status();
}

} // namespace fl

extern "C" {

// doubles of scratch per matrix: inverted diagonal blocks, panel, one block row of products, the right-hand side copy
size_t fl_chol_blocked_workspace_bytes(int batch, int n, int nrhs_tmp)
{
    if (batch <= 0 || n <= 0) return 0;
    const size_t nblk = ((size_t)n + fl::CNB - 1) / fl::CNB;
    const size_t per = nblk * fl::CNB * fl::CNB + (size_t)n * fl::CNB + (size_t)fl::CNB * (nrhs_tmp > 0 ? nrhs_tmp : 1) + (size_t)n;
    return (size_t)batch * per * sizeof(double);
}

static int setup(fl::Chol &c, int batch, int n, double *A, int lda, size_t strideA, int32_t *info, double *ws, hipStream_t st)
{
    c.batch = batch;
    c.n = n;
    c.lda = lda;
    c.nblk = (n + fl::CNB - 1) / fl::CNB;
    c.A = A;
    c.strideA = strideA;
    c.strideW = (size_t)c.nblk * fl::CNB * fl::CNB;
    c.strideP = (size_t)n * fl::CNB;
    c.W = ws;
    c.P = ws + (size_t)batch * c.strideW;
    c.info = info;
    c.st = st;
    return hipMemsetAsync(info, 0, sizeof(int32_t) * batch, st) == hipSuccess ? FL_OK : FL_ERR_LAUNCH;
}

// A [batch][n][lda] column-major SPD (lower triangle referenced) -> Cholesky factor; b [batch][n] -> A^{-1} b
int fl_dposv_blocked(int batch, int n, double *A_dev, int lda, double *b_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                     void *stream)
{
    if (!A_dev || !b_dev || !info_dev || batch <= 0 || n <= 0 || lda < n) return FL_ERR_INVALID_ARGUMENT;
    if (!ws_dev || ws_bytes < fl_chol_blocked_workspace_bytes(batch, n, 1)) return FL_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fl::Chol c;
    double *ws = static_cast<double *>(ws_dev);
    int rc = setup(c, batch, n, A_dev, lda, (size_t)n * lda, info_dev, ws, st);
    if (rc != FL_OK) return rc;
    double *tmp = c.P + (size_t)batch * c.strideP, *t = tmp + (size_t)batch * fl::CNB;
    if ((rc = fl::potrf(c)) != FL_OK) return rc;
    if (hipMemcpyAsync(t, b_dev, sizeof(double) * (size_t)batch * n, hipMemcpyDeviceToDevice, st) != hipSuccess) return FL_ERR_LAUNCH;
    if ((rc = fl::forward(c, t, 1, (size_t)n, tmp, (size_t)fl::CNB)) != FL_OK) return rc;
    if ((rc = fl::backward(c, t, 1, (size_t)n, tmp, (size_t)fl::CNB)) != FL_OK) return rc;
    for (int b0 = 0; b0 < batch; b0 += FL_GRID_YZ_MAX) {
        const int nb = batch - b0 < FL_GRID_YZ_MAX ? batch - b0 : FL_GRID_YZ_MAX;
        hipLaunchKernelGGL(fl::commit_rhs_kernel, dim3((n + 255) / 256, nb), dim3(256), 0, st, n, t + (size_t)b0 * n, b_dev + (size_t)b0 * n,
                           info_dev + b0);
    }
    return fl::launch_status();
}

// A [batch][n][lda] SPD (lower referenced) -> A^{-1}, both triangles (My_dpotri + dsyL2U); X_dev: [batch][n][n] scratch
int fl_dpotri_blocked(int batch, int n, double *A_dev, int lda, double *X_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                      void *stream)
{
    if (!A_dev || !X_dev || !info_dev || batch <= 0 || n <= 0 || lda < n) return FL_ERR_INVALID_ARGUMENT;
    if (!ws_dev || ws_bytes < fl_chol_blocked_workspace_bytes(batch, n, n)) return FL_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fl::Chol c;
    double *ws = static_cast<double *>(ws_dev);
    int rc = setup(c, batch, n, A_dev, lda, (size_t)n * lda, info_dev, ws, st);
    if (rc != FL_OK) return rc;
    double *tmp = c.P + (size_t)batch * c.strideP; // [batch][CNB][n]
    if ((rc = fl::potrf(c)) != FL_OK) return rc;
    // X = L^{-1}: forward substitution on the identity
    for (int b0 = 0; b0 < batch; b0 += FL_GRID_YZ_MAX) {
        const int nb = batch - b0 < FL_GRID_YZ_MAX ? batch - b0 : FL_GRID_YZ_MAX;
        hipLaunchKernelGGL(fl::fill_kernel, dim3((n + 255) / 256, n, nb), dim3(256), 0, st, n, n, X_dev + (size_t)b0 * n * n, n, (size_t)n * n, 0);
    }
    if ((rc = fl::forward(c, X_dev, n, (size_t)n * n, tmp, (size_t)fl::CNB * n)) != FL_OK) return rc;
    // A^{-1} = X^T X.  (Where the factorisation failed -- info != 0 -- the content of A is unspecified: the reference's
    // My_dpotri does not call dpotri then and leaves the partial factor, which no caller of it reads either.)
    rc = fl_dgemm_strided(1, 0, n, n, n, 1.0, X_dev, n, (size_t)n * n, X_dev, n, (size_t)n * n, 0.0, A_dev, lda, (size_t)n * lda,
                          batch, 0, st);
    return rc;
}

// Whitening by a Gram matrix (the Cholesky-QR step of fl_dsyev_vectors): G (n x n, ldg, lower referenced, SPD) -> its
// Cholesky factor L; Y (n x ncols, ld n) <- L^{-1} Y, so that the ROWS of Y become orthonormal when G = Y Y^T.
// *info_dev != 0: G was not positive definite to rounding (Y is then not to be used).  ws: fl_chol_blocked_workspace_bytes(1, n, ncols).
int fl_chol_whiten(int n, int ncols, double *G_dev, int ldg, double *Y_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                   void *stream)
{
    if (!G_dev || !Y_dev || !info_dev || n <= 0 || ncols <= 0 || ldg < n) return FL_ERR_INVALID_ARGUMENT;
    if (!ws_dev || ws_bytes < fl_chol_blocked_workspace_bytes(1, n, ncols)) return FL_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    fl::Chol c;
    int rc = setup(c, 1, n, G_dev, ldg, (size_t)n * ldg, info_dev, static_cast<double *>(ws_dev), st);
    if (rc != FL_OK) return rc;
    double *tmp = c.P + c.strideP; // [CNB][ncols]
    if ((rc = fl::potrf(c)) != FL_OK) return rc;
    return fl::forward(c, Y_dev, ncols, (size_t)n * ncols, tmp, (size_t)fl::CNB * ncols);
}

} // extern "C"
