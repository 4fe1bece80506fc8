// fl_dense_kernels.hip -- batched LinearAlgebra primitives of the path as stand-alone kernels:
//   fl_dposv_batched  <- My_dposv  (LinearAlgebra.f90:719-730, LAPACK dposv 'L')
//   fl_dpotri_batched <- My_dpotri (LinearAlgebra.f90:798-812, dpotrf+dpotri 'L') followed by dsyL2U (260-265)
// One workgroup per matrix, the device routines of fl_dense.hpp (the same ones NewtonRaphson and the exact-
// Hessian refresh of BFGS use inside the fused solver).
#include "fl_device.hpp"
#include "fl_host.hpp"
#include <stdlib.h>

namespace fl {

template <int NW, int EPT>
__global__ __launch_bounds__(NW * 64) void dposv_kernel(int n, double *A_all, double *b_all, int32_t *info)
{
    using D = Dense<NW, EPT>;
    constexpr int NPAD = D::NPAD;
    __shared__ __attribute__((aligned(16))) double lds[NPAD + 16 + 2 * Reducer<NW>::NVMAX * NW]; // row buffer, pivot / block slots, reducer
    const int prob = blockIdx.x;
    double *A = A_all + (size_t)prob * n * NPAD;
    Reducer<NW> R{lds + NPAD + 16, 0};
    const int inf = D::cholesky(A, n, lds, lds + NPAD);
    if (inf == 0) {
        double b[EPT];
        load_user<NW, EPT>(b_all + (size_t)prob * n, n, b);
        D::solve(A, n, b, R, lds + NPAD);
        store_user<NW, EPT>(b_all + (size_t)prob * n, n, b);
    }
    if (threadIdx.x == 0 && info) info[prob] = inf;
}

template <int NW, int EPT>
__global__ __launch_bounds__(NW * 64) void dpotri_kernel(int n, double *A_all, double *W_all, int32_t *info)
{
    using D = Dense<NW, EPT>;
    constexpr int NPAD = D::NPAD;
    __shared__ __attribute__((aligned(16))) double lds[NPAD + 16];
    const int prob = blockIdx.x;
    double *A = A_all + (size_t)prob * n * NPAD, *W = W_all + (size_t)prob * n * NPAD;
    const int inf = D::cholesky(A, n, lds, lds + NPAD);
    if (inf == 0) {
        D::inverse_factor(A, W, n, lds);
        D::wtw(W, A, n, lds);
    }
    if (threadIdx.x == 0 && info) info[prob] = inf;
}

// My_dsysv (LinearAlgebra.f90:695-703: LAPACK dsysv 'L', symmetric indefinite solve) for the KKT systems of
// LagrangianMultiplier (NO.f90:1950-1993).  LAPACK factorises with Bunch-Kaufman pivoting inside closed MKL; what
// is restated here is the solve itself -- Gaussian elimination with partial pivoting on the matrix the lower
// triangle defines, rows never moved (a row is "done" once it has been a pivot), right-hand side carried along,
// back substitution in axpy form.  Every element sees its updates in elimination order: the oracle's flo_dsysv
// replays it bit for bit.  One workgroup per system; O(n^3) streamed through L2/HBM -- sized for N+M of a few
// hundred, not a LAPACK replacement.
template <int NW, int EPT>
__global__ __launch_bounds__(NW * 64) void dsysv_kernel(int n, double *A_all, double *b_all, int32_t *info)
{
    using G = Geo<NW, EPT>;
    constexpr int NPAD = G::NPAD, T = G::T;
    __shared__ double redv[T];
    __shared__ int redi[T];
    __shared__ int piv[NPAD];
    __shared__ double bc[2];
    const int prob = blockIdx.x, tid = threadIdx.x;
    double *A = A_all + (size_t)prob * n * NPAD;
    double *bu = b_all + (size_t)prob * n;
    auto row_of = [&](int r) { return G::e0(r >> 1) + (r & 1); };
    // the upper triangle from the lower one (dsysv 'L' references the lower triangle only)
    for (int j = 1; j < n; ++j) {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = row_of(r);
            if (i < j) A[(size_t)j * NPAD + i] = A[(size_t)i * NPAD + j];
        }
    }
    __syncthreads();
    double b[EPT];
    int pstep[EPT]; // elimination step at which the row was the pivot (n = not yet)
    load_user<NW, EPT>(bu, n, b);
#pragma unroll
    for (int r = 0; r < EPT; ++r) pstep[r] = n;
    int inf = 0;
    for (int k = 0; k < n; ++k) {
        double col[EPT];
        load_pad<NW, EPT>(A + (size_t)k * NPAD, col);
        double bv = -1.0;
        int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = row_of(r);
            if (i < n && pstep[r] == n) {
                const double v = fabs(col[r]);
                if (v > bv || (v == bv && i < bi)) {
                    bv = v;
                    bi = i;
                }
            }
        }
        redv[tid] = bv;
        redi[tid] = bi;
        __syncthreads();
        for (int s2 = T / 2; s2 > 0; s2 >>= 1) {
            if (tid < s2) {
                const double v2 = redv[tid + s2];
                const int i2 = redi[tid + s2];
                if (v2 > redv[tid] || (v2 == redv[tid] && i2 < redi[tid])) {
                    redv[tid] = v2;
                    redi[tid] = i2;
                }
            }
            __syncthreads();
        }
        const double best = redv[0];
        const int p = redi[0];
        if (!(best > 0.0)) { // exactly singular (or NaN): LAPACK's info > 0
            inf = k + 1;
            break;
        }
        if (tid == 0) piv[k] = p;
#pragma unroll
        for (int r = 0; r < EPT; ++r)
            if (row_of(r) == p) {
                bc[0] = col[r];
                bc[1] = b[r];
                pstep[r] = k;
            }
        __syncthreads();
        const double apk = bc[0], bp = bc[1];
        double l[EPT];
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const bool live = (row_of(r) < n) && pstep[r] == n;
            l[r] = live ? col[r] / apk : 0.0;
            if (live) b[r] = b[r] - l[r] * bp;
        }
        for (int j = k + 1; j < n; ++j) {
            double c[EPT];
            double *cj = A + (size_t)j * NPAD;
            load_pad<NW, EPT>(cj, c);
            const double apj = cj[p];
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (l[r] != 0.0) c[r] = c[r] - l[r] * apj;
            store_pad<NW, EPT>(cj, c);
        }
        __syncthreads(); // the next pivot row is read by every thread after all updates of this step
    }
    if (inf == 0) {
        for (int k = n - 1; k >= 0; --k) { // x_k = b_p / A(p,k); rows that were pivots earlier: b_i -= A(i,k) x_k
            const int p = piv[k];
            double col[EPT];
            load_pad<NW, EPT>(A + (size_t)k * NPAD, col);
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (row_of(r) == p) bc[k & 1] = b[r] / col[r];
            __syncthreads();
            const double xk = bc[k & 1];
#pragma unroll
            for (int r = 0; r < EPT; ++r)
                if (pstep[r] < k) b[r] = b[r] - col[r] * xk;
            if (tid == 0) bu[k] = xk;
        }
    }
    if (tid == 0 && info) info[prob] = inf;
}

} // namespace fl

extern "C" {

#define FL_GEO_DISPATCH(KERNEL, ...)                                                                              \
    do {                                                                                                          \
        const int nw = threads / 64;                                                                              \
        if (nw == 1 && ept == 2) hipLaunchKernelGGL((fl::KERNEL<1, 2>), dim3(batch), dim3(64), 0, st, __VA_ARGS__); \
        else if (nw == 1 && ept == 4) hipLaunchKernelGGL((fl::KERNEL<1, 4>), dim3(batch), dim3(64), 0, st, __VA_ARGS__); \
        else if (nw == 1 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<1, 8>), dim3(batch), dim3(64), 0, st, __VA_ARGS__);  \
        else if (nw == 2 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<2, 8>), dim3(batch), dim3(128), 0, st, __VA_ARGS__); \
        else if (nw == 4 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<4, 8>), dim3(batch), dim3(256), 0, st, __VA_ARGS__); \
        else if (nw == 8 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<8, 8>), dim3(batch), dim3(512), 0, st, __VA_ARGS__); \
        else return FL_ERR_UNSUPPORTED_SIZE; /* a geometry none of the dense kernels is built for */                 \
    } while (0)

// From this order on the blocked multi-workgroup Cholesky (fl_chol_blocked.hip: O(n^3) on the f64 matrix cores) takes
// over from the one-workgroup kernels (sequential-order sums, bit-replayed by the oracle).  Any n then works: the
// leading dimension stays fl_reduction_geometry's threads*ept.  Tests lower it through the environment to hold the
// blocked path to the reference's n = 1024 results.
static int g_blocked_min_n = -1;
static int blocked_min_n()
{
    if (g_blocked_min_n < 0) {
        const char *e = getenv("FL_CHOL_BLOCKED_MIN_N");
        // measured (profiles/r02/chol_ab.txt): n = 256 x 4096 matrices 8.3 ms sequential / 13.2 blocked (posv), but
        // 512 x 1024: 15.4 / 10.9, 1024 x 256: 51.4 / 13.6, 4096 x 16: 972 / 61 -- the blocked path from n = 512 on
        g_blocked_min_n = (e && atoi(e) > 0) ? atoi(e) : 512;
    }
    return g_blocked_min_n;
}
// sets the order from which fl_dposv_batched / fl_dpotri_batched use the blocked path; returns the previous value
int fl_set_chol_blocked_min_n(int n)
{
    const int old = blocked_min_n();
    if (n > 0) g_blocked_min_n = n;
    return old;
}
// scratch from the stream-ordered allocator (released when the work queued before the free has finished)
struct StreamScratch {
    void *p = nullptr;
    hipStream_t st;
    StreamScratch(size_t bytes, hipStream_t s) : st(s)
    {
        if (hipMallocAsync(&p, bytes ? bytes : 8, st) != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();
        }
    }
    ~StreamScratch() { if (p) (void)hipFreeAsync(p, st); }
};

int fl_dposv_batched(int batch, int n, double *A_dev, double *b_dev, int32_t *info_dev, void *stream)
{
    if (!A_dev || !b_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n >= blocked_min_n()) {
        const size_t wsb = fl_chol_blocked_workspace_bytes(batch, n, 1);
        StreamScratch ws(wsb, st), inf(info_dev ? 0 : sizeof(int32_t) * batch, st);
        if (!ws.p || (!info_dev && !inf.p)) return FL_ERR_WORKSPACE;
        return fl_dposv_blocked(batch, n, A_dev, threads * ept, b_dev, info_dev ? info_dev : static_cast<int32_t *>(inf.p),
                                ws.p, wsb, st);
    }
    // one workgroup per matrix with its rows in registers: the register geometries only
    if (n > 4096) return FL_ERR_UNSUPPORTED_SIZE;
    FL_GEO_DISPATCH(dposv_kernel, n, A_dev, b_dev, info_dev);
    return fl::launch_status();
}

int fl_dpotri_batched(int batch, int n, double *A_dev, double *work_dev, int32_t *info_dev, void *stream)
{
    if (!A_dev || !work_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n >= blocked_min_n()) { // work_dev ([batch][n][ld] by contract) holds X = L^{-1} as [batch][n][n]
        const size_t wsb = fl_chol_blocked_workspace_bytes(batch, n, n);
        StreamScratch ws(wsb, st), inf(info_dev ? 0 : sizeof(int32_t) * batch, st);
        if (!ws.p || (!info_dev && !inf.p)) return FL_ERR_WORKSPACE;
        return fl_dpotri_blocked(batch, n, A_dev, threads * ept, work_dev, info_dev ? info_dev : static_cast<int32_t *>(inf.p),
                                 ws.p, wsb, st);
    }
    if (n > 4096) return FL_ERR_UNSUPPORTED_SIZE;
    FL_GEO_DISPATCH(dpotri_kernel, n, A_dev, work_dev, info_dev);
    return fl::launch_status();
}

int fl_dsysv_batched(int batch, int n, double *A_dev, double *b_dev, int32_t *info_dev, void *stream)
{
    if (!A_dev || !b_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (n > 4096 || fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FL_GEO_DISPATCH(dsysv_kernel, n, A_dev, b_dev, info_dev);
    return fl::launch_status();
}

} // extern "C"
