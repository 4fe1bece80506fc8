// fl_host.hpp -- host-side helpers shared by the translation units of libFL.so
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstddef>
#include "../../include/fl_nlopt.h"

#define FL_GRID_YZ_MAX 65535 // gridDim.y / gridDim.z limit: launchers that put the matrix index there go in chunks

namespace fl {

// status of the launch(es) just issued on this thread: a failed launch (too much LDS for the device, an invalid
// grid, a missing code object for the GPU found) is FL_ERR_LAUNCH -- not to be confused with "no GPU at all"
// (FL_ERR_NO_DEVICE, decided by hipGetDeviceCount before anything is launched).  hipGetLastError also clears the
// sticky error, so one bad call does not fail the next one.
static inline int launch_status(hipError_t e) { return e == hipSuccess ? FL_OK : FL_ERR_LAUNCH; }
static inline int launch_status() { return launch_status(hipGetLastError()); }

// Central-difference Jacobian with the step rule of MKL's djacobi, which the reference calls for f'' when no fdd is passed
// (NO.f90:676, 981, 1067, 1258; TrustRegion 1779, 1833: eps = 1d-8).  MKL is closed; the rule was read off the points at
// which the real djacobi of the build image calls its fcn and is held to it bit for bit by tests/golden/mkl_djacobi.npz
// (tools/make_mkl_golden.py, tests/test_mkl_pins.py):
//     |x_j| >  eps:  fcn at x_j (1 + eps) and x_j (1 - eps),  h = eps * x_j  (signed)
//     |x_j| <= eps:  fcn at x_j + eps     and x_j - eps,      h = eps
//     fjac(:, j) = (f_plus - f_minus) * (0.5 / h)
// fcn(x, f): f[m] at x[n]; fjac is the Fortran array fjac(m, n) (column-major); x is restored on return.
template <class F> static inline void central_difference_jacobian(F &&fcn, int n, int m, double *fjac, double *x, double eps,
                                                                 double *fp, double *fm)
{
    for (int j = 0; j < n; ++j) {
        const double xj = x[j];
        double h;
        if (std::fabs(xj) > eps) {
            h = eps * xj;
            x[j] = xj * (1.0 + eps);
            fcn(x, fp);
            x[j] = xj * (1.0 - eps);
            fcn(x, fm);
        } else {
            h = eps;
            x[j] = xj + eps;
            fcn(x, fp);
            x[j] = xj - eps;
            fcn(x, fm);
        }
        x[j] = xj;
        const double w = 0.5 / h;
        for (int i = 0; i < m; ++i) fjac[(size_t)j * m + i] = (fp[i] - fm[i]) * w;
    }
}

} // namespace fl

// internal entry points shared between translation units (not part of include/fl_nlopt.h)
extern "C" {
// C = alpha op(A) op(B) + beta C for a strided batch on the f64 matrix cores (fl_blas_kernels.hip); transB: B given N x K
int fl_dgemm_strided(int transA, int transB, int M, int K, int N, double alpha, const double *A_dev, int lda, size_t strideA,
                     const double *B_dev, int ldb, size_t strideB, double beta, double *C_dev, int ldc, size_t strideC,
                     int batch, int lower_only, void *stream);
// blocked multi-workgroup Cholesky solve / inverse (fl_chol_blocked.hip)
size_t fl_chol_blocked_workspace_bytes(int batch, int n, int nrhs_tmp);
int fl_dposv_blocked(int batch, int n, double *A_dev, int lda, double *b_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                     void *stream);
int fl_dpotri_blocked(int batch, int n, double *A_dev, int lda, double *X_dev, int32_t *info_dev, void *ws_dev, size_t ws_bytes,
                      void *stream);
}
