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
    static constexpr int UNR = 4;
    __device__ __forceinline__ static int row_of(int k) { return G::e0(k >> 1) + (k & 1); }

    // A = L L^T in place (lower).  rowbuf: LDS [NPAD] doubles, slot: LDS [2] doubles.  Returns LAPACK's info.
    __device__ static int cholesky(double *A, int n, double *rowbuf, double *slot)
    {
        int info = 0;
        for (int j = 0; j < n && info == 0; ++j) {
            __syncthreads();
            for (int k = threadIdx.x; k < j; k += G::T) rowbuf[k] = A[(size_t)k * NPAD + j]; // L(j,k), k<j
            __syncthreads();
            double v[EPT];
            load_pad<NW, EPT>(A + (size_t)j * NPAD, v);
            for (int k0 = 0; k0 < j; k0 += UNR) {
                double c[UNR][EPT];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (k0 + u < j) load_pad<NW, EPT>(A + (size_t)(k0 + u) * NPAD, c[u]);
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (k0 + u < j) {
                        const double ljk = rowbuf[k0 + u];
#pragma unroll
                        for (int r = 0; r < EPT; ++r) v[r] = v[r] - c[u][r] * ljk;
                    }
            }
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (row_of(r) == j) slot[j & 1] = v[r];
            __syncthreads();
            const double piv = slot[j & 1];
            if (!(piv > 0.0)) {
                info = j + 1;
                break;
            }
            const double ajj = sqrt(piv);
#pragma unroll
            for (int r = 0; r < EPT; ++r) {
                const int i = row_of(r);
                v[r] = (i == j) ? ajj : v[r] / ajj;
            }
            store_pad<NW, EPT>(A + (size_t)j * NPAD, v);
        }
        __syncthreads();
        return info;
    }

    // solve L L^T x = b, b in registers (in/out).  Forward: column-oriented axpy (dtrsv 'L','N');
    // backward: x_j = (z_j - sum_{i>j} L(i,j) x_i) / L(j,j) with the sum reduced in the kernels' order.
    __device__ static void solve(const double *L, int n, double (&b)[EPT], Reducer<NW> &R, double *slot)
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
    __device__ static void inverse_factor(const double *L, double *Wt, int n, double *rowbuf)
    {
        for (int j = 0; j < n; ++j) {
            __syncthreads();
            for (int k = threadIdx.x; k <= j; k += G::T) rowbuf[k] = L[(size_t)k * NPAD + j]; // L(j,k), k<=j
            __syncthreads();
            double v[EPT];
#pragma unroll
            for (int r = 0; r < EPT; ++r) v[r] = (row_of(r) == j) ? 1.0 : 0.0;
            for (int k0 = 0; k0 < j; k0 += UNR) {
                double w[UNR][EPT];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (k0 + u < j) load_pad<NW, EPT>(Wt + (size_t)(k0 + u) * NPAD, w[u]);
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (k0 + u < j) {
                        const double ljk = rowbuf[k0 + u];
#pragma unroll
                        for (int r = 0; r < EPT; ++r) v[r] = v[r] - ljk * w[u][r];
                    }
            }
            const double ljj = rowbuf[j];
#pragma unroll
            for (int r = 0; r < EPT; ++r) v[r] = (row_of(r) < n) ? v[r] / ljj : 0.0;
            store_pad<NW, EPT>(Wt + (size_t)j * NPAD, v);
        }
        __syncthreads();
    }

    // Ainv = W^T W (full symmetric matrix, column-major): Ainv(a,b) = sum_k W(k,a) W(k,b), k in order
    __device__ static void wtw(const double *Wt, double *Ainv, int n, double *rowbuf)
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
