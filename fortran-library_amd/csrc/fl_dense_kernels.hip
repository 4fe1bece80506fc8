// fl_dense_kernels.hip -- batched LinearAlgebra primitives of the path as stand-alone kernels:
//   fl_dposv_batched  <- My_dposv  (LinearAlgebra.f90:719-730, LAPACK dposv 'L')
//   fl_dpotri_batched <- My_dpotri (LinearAlgebra.f90:798-812, dpotrf+dpotri 'L') followed by dsyL2U (260-265)
// One workgroup per matrix, the device routines of fl_dense.hpp (the same ones NewtonRaphson and the exact-
// Hessian refresh of BFGS use inside the fused solver).
#include "fl_device.hpp"

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

} // namespace fl

extern "C" {

#define FL_GEO_DISPATCH(KERNEL, ...)                                                                              \
    do {                                                                                                          \
        const int nw = threads / 64;                                                                              \
        if (nw == 1 && ept == 2) hipLaunchKernelGGL((fl::KERNEL<1, 2>), dim3(batch), dim3(64), 0, st, __VA_ARGS__); \
        else if (nw == 1 && ept == 4) hipLaunchKernelGGL((fl::KERNEL<1, 4>), dim3(batch), dim3(64), 0, st, __VA_ARGS__); \
        else if (nw == 2 && ept == 4) hipLaunchKernelGGL((fl::KERNEL<2, 4>), dim3(batch), dim3(128), 0, st, __VA_ARGS__); \
        else if (nw == 2 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<2, 8>), dim3(batch), dim3(128), 0, st, __VA_ARGS__); \
        else if (nw == 4 && ept == 8) hipLaunchKernelGGL((fl::KERNEL<4, 8>), dim3(batch), dim3(256), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((fl::KERNEL<8, 8>), dim3(batch), dim3(512), 0, st, __VA_ARGS__);                   \
    } while (0)

int fl_dposv_batched(int batch, int n, double *A_dev, double *b_dev, int32_t *info_dev, void *stream)
{
    if (!A_dev || !b_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FL_GEO_DISPATCH(dposv_kernel, n, A_dev, b_dev, info_dev);
    return hipGetLastError() == hipSuccess ? FL_OK : FL_ERR_NO_DEVICE;
}

int fl_dpotri_batched(int batch, int n, double *A_dev, double *work_dev, int32_t *info_dev, void *stream)
{
    if (!A_dev || !work_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FL_GEO_DISPATCH(dpotri_kernel, n, A_dev, work_dev, info_dev);
    return hipGetLastError() == hipSuccess ? FL_OK : FL_ERR_NO_DEVICE;
}

} // extern "C"
