// fl_dense.hpp -- dense SPD kernels of the path, one workgroup per matrix (device only).
//
// Reference: LinearAlgebra.f90  My_dposv 719-730 (LAPACK dposv 'L': Cholesky solve),
//            My_dpotri 798-812 (dpotrf + dpotri 'L': SPD inverse), dsyL2U 260-265;
// used by NewtonRaphson (NO.f90:1067, 1232) and the exact-Hessian refresh of BFGS (NO.f90:677, 951).
// The reference calls MKL; what is restated here is LAPACK's documented semantics (lower Cholesky,
// info = index of the first non-positive pivot), with every inner sum in "axpy form": the thread that
// owns row i accumulates over k in order, so no reduction is needed and the result is the sequential-order
// sum.  Matrices are column-major [n][NPAD] (the solver's layout); a thread owns the rows of its register
// elements (fl_device.hpp).  Only the lower triangle of a factor is meaningful.
#pragma once
// (included by fl_device.hpp after Geo / load_pad / store_pad / Reducer are defined)

namespace fl {

template <int NW, int EPT> struct Dense {
    using G = Geo<NW, EPT>;
    static constexpr int NPAD = G::NPAD;
    __device__ __forceinline__ static int row_of(int k) { return G::e0(k >> 1) + (k & 1); }

    // A = L L^T in place (lower).  rowbuf: LDS [>= BW*KT] doubles, slot: LDS [2 + BW] doubles.  Returns LAPACK's info.
    // Left-looking, BW columns at a time: every earlier column is read ONCE per block of BW columns (its BW
    // multipliers L(j0+u,k) come from LDS), then the block is finished column by column in registers.  Each
    // element still sees  v = v - L(i,k) L(j,k)  for k = 0, 1, 2, ... in that order, so the factor is bitwise
    // the one of the column-at-a-time sweep (and of the oracle's sequential sums).
#ifndef FL_DENSE_BW
#define FL_DENSE_BW 8 // measured: Newton n=256 x 4096: 143 k it/s (2), 254 k (4), 387 k (8); column-at-a-time 79 k
#endif
    // n <= 128 (EPT = 2): 4 -- the wider block costs the small BFGS kernels 70 VGPRs (4 -> 2 waves/SIMD) for nothing
    // 512 threads x 8 elements (2 waves / SIMD, 256 VGPRs): 6 -- with 8 the Newton kernels spilled 9-200 VGPRs
    static constexpr int BW = (EPT <= 2 && FL_DENSE_BW > 4) ? 4 : ((NW >= 8 && EPT >= 8 && FL_DENSE_BW > 6) ? 6 : FL_DENSE_BW), KT = (NPAD / BW < 256) ? NPAD / BW : 256; // BW*KT <= NPAD doubles of LDS
    static constexpr int UNR = BW; // columns of W^T W per sweep
    __device__ __forceinline__ static int cholesky(double *A, int n, double *rowbuf, double *slot)
    {
        int info = 0;
        for (int j0 = 0; j0 < n && info == 0; j0 += BW) {
            const int bw = (n - j0 < BW) ? n - j0 : BW;
            double v[BW][EPT];
#pragma unroll
            for (int u = 0; u < BW; ++u)
                if (u < bw) load_pad<NW, EPT>(A + (size_t)(j0 + u) * NPAD, v[u]);
            for (int kt = 0; kt < j0; kt += KT) {
                const int kn = (j0 - kt < KT) ? j0 - kt : KT;
                __syncthreads();
                for (int i = G::tid(); i < kn * BW; i += G::T) { // rowbuf[u*KT + kk] = L(j0+u, kt+kk)
                    const int kk = i / BW, u = i - kk * BW;
                    rowbuf[u * KT + kk] = (u < bw) ? A[(size_t)(kt + kk) * NPAD + j0 + u] : 0.0;
                }
                __syncthreads();
                for (int k0 = 0; k0 < kn; k0 += 2) {
                    double c[2][EPT];
                    load_pad<NW, EPT>(A + (size_t)(kt + k0) * NPAD, c[0]);
                    if (k0 + 1 < kn) load_pad<NW, EPT>(A + (size_t)(kt + k0 + 1) * NPAD, c[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (k0 + h < kn) {
#pragma unroll
                            for (int u = 0; u < BW; ++u) {
                                const double l = rowbuf[u * KT + k0 + h];
#pragma unroll
                                for (int r = 0; r < EPT; ++r) v[u][r] = v[u][r] - c[h][r] * l;
                            }
                        }
                    }
                }
            }
            // the block itself: column j0+u is finished, then subtracted from the columns to its right
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                if (u < bw && info == 0) {
                    const int j = j0 + u;
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EPT; ++r)
                        if (row_of(r) == j) slot[0] = v[u][r];
                    __syncthreads();
                    const double piv = slot[0];
                    if (!(piv > 0.0)) {
                        info = j + 1;
                    } else {
                        const double ajj = sqrt(piv);
#pragma unroll
                        for (int r = 0; r < EPT; ++r) {
                            const int i = row_of(r);
                            v[u][r] = (i == j) ? ajj : v[u][r] / ajj;
                            if (i > j && i < j0 + BW) slot[2 + (i - j0)] = v[u][r]; // L(j0+u', j) for the block's later columns
                        }
                        store_pad<NW, EPT>(A + (size_t)j * NPAD, v[u]);
                        __syncthreads();
#pragma unroll
                        for (int u2 = 0; u2 < BW; ++u2) {
                            if (u2 > u && u2 < bw) {
                                const double l = slot[2 + u2];
#pragma unroll
                                for (int r = 0; r < EPT; ++r) v[u2][r] = v[u2][r] - v[u][r] * l;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        return info;
    }

    // solve L L^T x = b, b in registers (in/out).  Forward: column-oriented axpy (dtrsv 'L','N');
    // backward: x_j = (z_j - sum_{i>j} L(i,j) x_i) / L(j,j) with the sum reduced in the kernels' order.
    __device__ __forceinline__ static void solve(const double *L, int n, double (&b)[EPT], Reducer<NW> &R, double *slot)
    {
        for (int j = 0; j < n; ++j) {
            double c[EPT];
            load_pad<NW, EPT>(L + (size_t)j * NPAD, c);
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (row_of(r) == j) slot[j & 1] = b[r] / c[r];
            __syncthreads();
            const double q = slot[j & 1];
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int i = row_of(r);
                if (i == j) b[r] = q;
                else if (i > j && i < n) b[r] = b[r] - q * c[r];
            }
        }
        for (int j = n - 1; j >= 0; --j) {
            double c[EPT];
            load_pad<NW, EPT>(L + (size_t)j * NPAD, c);
            double q[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int i = row_of(r);
                const double t = (i > j && i < n) ? c[r] * b[r] : 0.0;
                q[0] = (r == 0) ? t : q[0] + t;
                if (i == j) {
                    q[1] = b[r];
                    q[2] = c[r];
                }
            }
            R.run(q);
            const double xj = (q[1] - q[0]) / q[2];
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (row_of(r) == j) b[r] = xj;
        }
    }

    // W = inverse of the lower factor, stored by ROWS: Wt[j*NPAD + c] = W(j,c).
    // W(j,:) = (e_j - sum_{k<j} L(j,k) W(k,:)) / L(j,j): each thread owns columns c, sums over k in order.
    // BW rows at a time (every earlier row of W is read once per block; same order of operations per element).
    __device__ __forceinline__ static void inverse_factor(const double *L, double *Wt, int n, double *rowbuf)
    {
        for (int j0 = 0; j0 < n; j0 += BW) {
            const int bw = (n - j0 < BW) ? n - j0 : BW;
            double v[BW][EPT];
#pragma unroll
            for (int u = 0; u < BW; ++u)
#pragma unroll
                for (int r = 0; r < EPT; ++r) v[u][r] = (row_of(r) == j0 + u) ? 1.0 : 0.0;
            for (int kt = 0; kt < j0; kt += KT) {
                const int kn = (j0 - kt < KT) ? j0 - kt : KT;
                __syncthreads();
                for (int i = G::tid(); i < kn * BW; i += G::T) { // rowbuf[u*KT + kk] = L(j0+u, kt+kk)
                    const int kk = i / BW, u = i - kk * BW;
                    rowbuf[u * KT + kk] = (u < bw) ? L[(size_t)(kt + kk) * NPAD + j0 + u] : 0.0;
                }
                __syncthreads();
                for (int k0 = 0; k0 < kn; k0 += 2) {
                    double w[2][EPT];
                    load_pad<NW, EPT>(Wt + (size_t)(kt + k0) * NPAD, w[0]);
                    if (k0 + 1 < kn) load_pad<NW, EPT>(Wt + (size_t)(kt + k0 + 1) * NPAD, w[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (k0 + h < kn) {
#pragma unroll
                            for (int u = 0; u < BW; ++u) {
                                const double l = rowbuf[u * KT + k0 + h];
#pragma unroll
                                for (int r = 0; r < EPT; ++r) v[u][r] = v[u][r] - l * w[h][r];
                            }
                        }
                    }
                }
            }
            // inside the block: L(j0+u, j0+u') for u' <= u (BW x BW values)
            __syncthreads();
            for (int i = G::tid(); i < BW * BW; i += G::T) {
                const int u = i / BW, u2 = i - u * BW;
                rowbuf[i] = (u < bw && u2 <= u) ? L[(size_t)(j0 + u2) * NPAD + j0 + u] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < BW; ++u) {
                if (u < bw) {
#pragma unroll
                    for (int u2 = 0; u2 < BW; ++u2) {
                        if (u2 < u) {
                            const double l = rowbuf[u * BW + u2];
#pragma unroll
                            for (int r = 0; r < EPT; ++r) v[u][r] = v[u][r] - l * v[u2][r];
                        }
                    }
                    const double ljj = rowbuf[u * BW + u];
#pragma unroll
                    for (int r = 0; r < EPT; ++r) v[u][r] = (row_of(r) < n) ? v[u][r] / ljj : 0.0;
                    store_pad<NW, EPT>(Wt + (size_t)(j0 + u) * NPAD, v[u]);
                }
            }
        }
        __syncthreads();
    }

    // Ainv = W^T W (full symmetric matrix, column-major): Ainv(a,b) = sum_k W(k,a) W(k,b), k in order
    __device__ __forceinline__ static void wtw(const double *Wt, double *Ainv, int n, double *rowbuf)
    {
        for (int b0 = 0; b0 < n; b0 += UNR) {
            double acc[UNR][EPT];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int r = 0; r < EPT; ++r) acc[u][r] = 0.0;
            for (int k = b0; k < n; ++k) { // W(k,b) = 0 for k < b
                double w[EPT];
                load_pad<NW, EPT>(Wt + (size_t)k * NPAD, w);
                const double *wk = Wt + (size_t)k * NPAD;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (b0 + u < n) {
                        const double wkb = wk[b0 + u]; // uniform scalar load
#pragma unroll
                        for (int r = 0; r < EPT; ++r) acc[u][r] = acc[u][r] + w[r] * wkb;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (b0 + u < n) store_pad<NW, EPT>(Ainv + (size_t)(b0 + u) * NPAD, acc[u]);
        }
        (void)rowbuf;
        __syncthreads();
    }
};

} // namespace fl
