// fl_bfgs_gemm.hip -- the BFGS inverse-Hessian update exactly as the reference writes it,
//     U = I - rho y s^T ;  H <- U^T (H U) + rho s s^T        (NO.f90:958-962 = 989-993 = 1010-1014)
// i.e. two dense n x n x n matrix products, on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// The solver kernels use the algebraically equal O(n^2) rank-2 form (fl_device.hpp, direction_bfgs);
// this file is the as-written O(n^3) contraction for callers who want the reference's two-matmul
// arithmetic, and the MFMA-bound kernel of the path (4 n^3 flop per update).
//
// Both products are ONE kernel:   out[a*ld + b] = sum_k in[k*ld + a] * U[k][b]  (+ eps s_a s_b)
//   pass 1: in = H (column-major, H[a][k] at k*ld+a)  -> out = T = H U stored row-major (T[a][b] at a*ld+b)
//   pass 2: in = T (T[k][a] at k*ld+a)                -> out[a*ld+b] = sum_k T[k][a] U[k][b] = H'[b][a]
//           = column-major H' again, eps = rho adds rho s s^T in the epilogue.
// U is never materialised: its 4 x 16 B-fragments (delta_kb - (rho y_k) s_b) are formed in registers,
// so only one operand streams from HBM (coalesced 1 KiB rows -> LDS, double buffered).
// Tile: 128 x 128 per workgroup, BK = 16; WGM x WGN waves share it (default 2 x 4: 64 x 32 = 8 accumulator tiles each).
// Fragment layouts of v_mfma_f64_16x16x4_f64: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15],
// C/D reg r of lane l = C[(l>>4) + 4 r][l&15].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"
#include "fl_reduce.hpp"

namespace fl {

using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int GBM = 128, GBN = 128, GBK = 16;

// rho = 1/(y.s), ry = rho*y for every problem (one workgroup per problem)
__global__ __launch_bounds__(256) void bfgs_rho_kernel(int n, const double *s_all, const double *y_all, double *rho_all,
                                                       double *ry_all)
{
    __shared__ double part[4];
    const int prob = blockIdx.x, tid = threadIdx.x;
    const double *s = s_all + (size_t)prob * n, *y = y_all + (size_t)prob * n;
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) acc = acc + y[i] * s[i];
    acc = wave_allreduce(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    const double rho = 1.0 / (((part[0] + part[1]) + part[2]) + part[3]);
    if (tid == 0) rho_all[prob] = rho;
    for (int i = tid; i < n; i += 256) ry_all[(size_t)prob * n + i] = rho * y[i];
}

// WGM x WGN waves per workgroup, each owning a (128/WGM) x (128/WGN) block of the tile = TM x TN accumulator tiles.
// Measured on n = 4096 (profiles/r01/mfma_f64.txt): 2 x 2 waves (16 accumulator tiles each, one wave per SIMD) 37.7,
// 4 x 4 (4 tiles, four waves per SIMD) 49.7, 4 x 2 61.7, 2 x 4 (8 tiles, two waves per SIMD) 65.4, 1 x 8 66.5 TFLOP/s
// of the 78.6 TFLOP/s datasheet peak: two waves per SIMD with 8 accumulator tiles each keep the f64 MFMA pipe fed.
#ifndef FL_GEMM_WGM
#define FL_GEMM_WGM 2
#define FL_GEMM_WGN 4
#endif
template <int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void bfgs_gemm_kernel(int n, int ld, const double *in_all, double *out_all,
                                                                   size_t mat_stride, const double *ry_all,
                                                                   const double *s_all, const double *rho_all, int add_ss)
{
    constexpr int NT = WGM * WGN * 64, TM = GBM / 16 / WGM, TN = GBN / 16 / WGN;
    constexpr int PER = GBK * GBM / NT; // doubles each thread stages per k-block (8 with 256 threads, 2 with 1024)
    static_assert(PER >= 2 && PER % 2 == 0, "staging in 16-byte pieces");
    __shared__ __attribute__((aligned(16))) double As[2][GBK][GBM];
    __shared__ double Rs[2][GBK];
    const int prob = blockIdx.y;
    const int tiles_b = (n + GBN - 1) / GBN;
    const int ta = blockIdx.x / tiles_b, tb = blockIdx.x % tiles_b;
    const int a0 = ta * GBM, b0 = tb * GBN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave / WGN, wb = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;
    const double *in = in_all + (size_t)prob * mat_stride;
    double *out = out_all + (size_t)prob * mat_stride;
    const double *ry = ry_all + (size_t)prob * n, *s = s_all + (size_t)prob * n;

    f64x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[m][t] = f64x4{0.0, 0.0, 0.0, 0.0};
    double sb[TN];
    int bcol[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        bcol[t] = b0 + 16 * TN * wb + 16 * t + lr;
        sb[t] = bcol[t] < n ? s[bcol[t]] : 0.0;
    }
    // staging: thread t moves PER doubles of the 16 x 128 tile: row kk = t / (128/PER), columns PER*(t % (128/PER))..
    constexpr int TPR = GBM / PER; // threads per row
    const int skk = tid / TPR, sa = (tid % TPR) * PER;
    double2 st[PER / 2];
    auto gload = [&](int k0) {
        const int k = k0 + skk;
        const double *row = in + (size_t)k * ld + a0 + sa;
#pragma unroll
        for (int u = 0; u < PER / 2; ++u) {
            const int a = a0 + sa + 2 * u;
            st[u] = (k < n && a + 1 < ld) ? *reinterpret_cast<const double2 *>(row + 2 * u) : make_double2(0.0, 0.0);
            if (a >= n) st[u].x = 0.0;
            if (a + 1 >= n) st[u].y = 0.0;
        }
    };
    auto lstore = [&](int buf, int k0) {
#pragma unroll
        for (int u = 0; u < PER / 2; ++u) *reinterpret_cast<double2 *>(&As[buf][skk][sa + 2 * u]) = st[u];
        if (tid < GBK) Rs[buf][tid] = (k0 + tid < n) ? ry[k0 + tid] : 0.0;
    };
    gload(0);
    lstore(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < n; k0 += GBK) {
        const bool more = k0 + GBK < n;
        if (more) gload(k0 + GBK);
#pragma unroll
        for (int ks = 0; ks < GBK / 4; ++ks) {
            const int kk = 4 * ks + lq;
            const int k = k0 + kk;
            const double ryk = Rs[buf][kk];
            double af[TM], bf[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) af[m] = As[buf][kk][16 * TM * wa + 16 * m + lr];
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[t] = ((k == bcol[t]) ? 1.0 : 0.0) - ryk * sb[t]; // U[k][b]
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[t], acc[m][t], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1, k0 + GBK);
        __syncthreads();
        buf ^= 1;
    }
    const double eps = add_ss ? rho_all[prob] : 0.0;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = a0 + 16 * TM * wa + 16 * m + lq + 4 * r;
            if (a < n) {
                const double esa = eps * s[a];
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    if (bcol[t] < n) out[(size_t)a * ld + bcol[t]] = acc[m][t][r] + esa * sb[t];
            }
        }
    }
}

} // namespace fl

extern "C" {

size_t fl_bfgs_update_gemm_workspace_bytes(int chunk, int n)
{
    int threads = 0, ept = 0;
    if (chunk <= 0 || fl_reduction_geometry(n, &threads, &ept) != FL_OK) return 0;
    const size_t ld = (size_t)threads * ept;
    return (size_t)chunk * ((size_t)n * ld + (size_t)n + 1) * sizeof(double);
}

int fl_bfgs_update_gemm_batched(int batch, int n, double *H_dev, const double *s_dev, const double *y_dev,
                                void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (!H_dev || !s_dev || !y_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    const size_t ld = (size_t)threads * ept, mat = (size_t)n * ld;
    const size_t per = (mat + (size_t)n + 1) * sizeof(double);
    if (!workspace_dev || workspace_bytes < per) return FL_ERR_WORKSPACE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t chunk = workspace_bytes / per;
    if (chunk > (size_t)batch) chunk = (size_t)batch;
    double *T = static_cast<double *>(workspace_dev);
    double *ry = T + chunk * mat, *rho = ry + chunk * (size_t)n;
    const int tiles = ((n + fl::GBM - 1) / fl::GBM) * ((n + fl::GBN - 1) / fl::GBN);
    for (size_t p0 = 0; p0 < (size_t)batch; p0 += chunk) {
        const int c = (int)((size_t)batch - p0 < chunk ? (size_t)batch - p0 : chunk);
        double *H = H_dev + p0 * mat;
        const double *s = s_dev + p0 * (size_t)n, *y = y_dev + p0 * (size_t)n;
        hipLaunchKernelGGL(fl::bfgs_rho_kernel, dim3(c), dim3(256), 0, st, n, s, y, rho, ry);
        hipLaunchKernelGGL((fl::bfgs_gemm_kernel<FL_GEMM_WGM, FL_GEMM_WGN>), dim3(tiles, c), dim3(FL_GEMM_WGM * FL_GEMM_WGN * 64), 0, st, n, (int)ld, H, T,
                           mat, ry, s, rho, 0);
        hipLaunchKernelGGL((fl::bfgs_gemm_kernel<FL_GEMM_WGM, FL_GEMM_WGN>), dim3(tiles, c), dim3(FL_GEMM_WGM * FL_GEMM_WGN * 64), 0, st, n, (int)ld, T, H,
                           mat, ry, s, rho, 1);
    }
    return fl::launch_status();
}

} // extern "C"
