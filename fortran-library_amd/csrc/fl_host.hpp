// fl_host.hpp -- host-side helpers shared by the translation units of libFL.so
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fl_nlopt.h"

namespace fl {

// status of the launch(es) just issued on this thread: a failed launch (too much LDS for the device, an invalid
// grid, a missing code object for the GPU found) is FL_ERR_LAUNCH -- not to be confused with "no GPU at all"
// (FL_ERR_NO_DEVICE, decided by hipGetDeviceCount before anything is launched).  hipGetLastError also clears the
// sticky error, so one bad call does not fail the next one.
static inline int launch_status(hipError_t e) { return e == hipSuccess ? FL_OK : FL_ERR_LAUNCH; }
static inline int launch_status() { return launch_status(hipGetLastError()); }

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
