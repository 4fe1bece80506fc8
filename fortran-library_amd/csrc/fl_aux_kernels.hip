// fl_aux_kernels.hip -- synthetic-input generators (Philox-4x32-10) and the
// stand-alone batched L-BFGS two-loop recursion (Before(), NO.f90:586-608) used
// to measure the recursion against the HBM roofline in isolation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"
#include "fl_reduce.hpp"

namespace fl {

// Philox-4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0,k1)
__host__ __device__ inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0;
        c[1] = n1;
        c[2] = n2;
        c[3] = n3;
        k0 += W0;
        k1 += W1;
    }
}
// 53 random bits -> (0,1)
__host__ __device__ inline double u01(uint32_t hi, uint32_t lo)
{
    const uint64_t m = ((uint64_t)hi << 21) | (lo >> 11);
    return ((double)m + 0.5) * (1.0 / 9007199254740992.0);
}

// element pair q of problem k: counter = (q, 0, k, stream); two doubles per call
__global__ void synth_uniform_kernel(uint64_t seed, int batch, int n, double lo, double hi, double *out)
{
    const int pairs = (n + 1) >> 1;
    const size_t total = (size_t)batch * pairs;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(i / pairs), q = (uint32_t)(i % pairs);
        uint32_t c[4] = {q, 0u, k, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        double *row = out + (size_t)k * n;
        const int e = 2 * (int)q;
        row[e] = lo + (hi - lo) * u01(c[0], c[1]);
        if (e + 1 < n) row[e + 1] = lo + (hi - lo) * u01(c[2], c[3]);
    }
}

// d[k][i] = 1 + (kappa_k - 1) * i/(n-1);  kappa_k = exp(log(lo) + u_k (log(hi)-log(lo))), counter (0,1,k,1)
__global__ void synth_spectrum_kernel(uint64_t seed, int batch, int n, double llo, double lhi, double *d)
{
    const size_t total = (size_t)batch * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(i / n);
        const int e = (int)(i % n);
        uint32_t c[4] = {0u, 1u, k, 1u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double kappa = exp(llo + u01(c[0], c[1]) * (lhi - llo));
        d[i] = 1.0 + (kappa - 1.0) * ((double)e / (double)(n > 1 ? n - 1 : 1));
    }
}

// ---------------------------------------------------------------------------
// Stand-alone two-loop recursion.  Same register layout, reduction order and
// load pipeline as the solver kernel (fl_solver_kernels.hip): one workgroup per
// problem, p in registers, 4*m history rows streamed from HBM.
template <int NW, int EPT>
__global__ __launch_bounds__(NW * 64) void two_loop_kernel(int n, int mem, int recent, const double *hist_all,
                                                           const double *rho_all, const double *g_all, double *p_all)
{
    constexpr int T = NW * 64, NPAD = T * EPT, NCH = EPT / 2;
    __shared__ double slots[2 * NW];
    __shared__ double alpha_s[FL_MAX_MEMORY], rho_s[FL_MAX_MEMORY];
    const int prob = blockIdx.x, tid = threadIdx.x;
    const double *hist = hist_all + (size_t)prob * (size_t)(2 * mem) * NPAD;
    if (tid < mem) rho_s[tid] = rho_all[(size_t)prob * mem + tid];
    int parity = 0;
    auto reduce = [&](double v) {
        v = wave_allreduce(v);
        if constexpr (NW > 1) {
            double *s = slots + parity * NW;
            if ((tid & 63) == 0) s[tid >> 6] = v;
            __syncthreads();
            double t = s[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) t = t + s[w];
            parity ^= 1;
            return t;
        } else {
            return v;
        }
    };
    auto ld = [&](const double *row, double (&v)[EPT]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const double2 t = *reinterpret_cast<const double2 *>(row + ((c * T + tid) << 1));
            v[2 * c] = t.x;
            v[2 * c + 1] = t.y;
        }
    };
    auto dotp = [&](const double (&a)[EPT], const double (&b)[EPT]) {
        double acc = a[0] * b[0];
#pragma unroll
        for (int k = 1; k < EPT; ++k) acc = acc + a[k] * b[k];
        return acc;
    };
    double p[EPT], sA[EPT], yA[EPT], sB[EPT], yB[EPT];
    {
        const double *grow = g_all + (size_t)prob * n;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int e = (c * T + tid) << 1;
            p[2 * c] = e < n ? grow[e] : 0.0;
            p[2 * c + 1] = e + 1 < n ? grow[e + 1] : 0.0;
        }
    }
    __syncthreads();
    auto slot_of = [&](int j) {
        int s = recent - j;
        return s < 0 ? s + mem : s;
    };
    auto fetch = [&](int j, double (&s_)[EPT], double (&y_)[EPT]) {
        const double *row = hist + (size_t)(2 * slot_of(j)) * NPAD;
        ld(row, s_);
        ld(row + NPAD, y_);
    };
    double yy = 0.0;
    auto down = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
        const int sl = slot_of(j);
        const double al = rho_s[sl] * reduce(dotp(s_, p));
        if (tid == 0) alpha_s[sl] = al;
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = p[k] - al * y_[k];
    };
    auto upw = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
        const int sl = slot_of(j);
        const double co = alpha_s[sl] - rho_s[sl] * reduce(dotp(y_, p));
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = p[k] + co * s_[k];
    };
    fetch(0, sA, yA);
    yy = reduce(dotp(yA, yA));
    for (int j = 0; j < mem; j += 2) {
        if (j + 1 < mem) fetch(j + 1, sB, yB);
        down(j, sA, yA);
        if (j + 1 < mem) {
            if (j + 2 < mem) fetch(j + 2, sA, yA);
            down(j + 1, sB, yB);
        }
    }
    const double rr = rho_s[recent];
#pragma unroll
    for (int k = 0; k < EPT; ++k) p[k] = p[k] / rr / yy;
    __syncthreads();
    // buffers by parity of j (even -> A, odd -> B): the oldest pair is still resident from the way down
    int j = mem - 1;
    if ((j & 1) == 0) {
        if (j - 1 >= 0) fetch(j - 1, sB, yB);
        upw(j, sA, yA);
        --j;
    }
    for (; j >= 1; j -= 2) {
        fetch(j - 1, sA, yA);
        upw(j, sB, yB);
        if (j - 2 >= 0) fetch(j - 2, sB, yB);
        upw(j - 1, sA, yA);
    }
    double *prow = p_all + (size_t)prob * n;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e = (c * T + tid) << 1;
        if (e < n) prow[e] = -p[2 * c];
        if (e + 1 < n) prow[e + 1] = -p[2 * c + 1];
    }
}

} // namespace fl

extern "C" {

int fl_synth_uniform(uint64_t seed, int batch, int n, double lo, double hi, double *out_dev, void *stream)
{
    if (!out_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipLaunchKernelGGL(fl::synth_uniform_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       batch, n, lo, hi, out_dev);
    return fl::launch_status();
}

int fl_synth_diag_spectrum(uint64_t seed, int batch, int n, double kappa_lo, double kappa_hi, double *d_dev,
                           void *stream)
{
    if (!d_dev || batch <= 0 || n <= 0 || !(kappa_lo > 0.0) || !(kappa_hi >= kappa_lo))
        return FL_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipLaunchKernelGGL(fl::synth_spectrum_kernel, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), seed,
                       batch, n, log(kappa_lo), log(kappa_hi), d_dev);
    return fl::launch_status();
}

int fl_lbfgs_two_loop_batched(int batch, int n, int memory, int recent, const double *hist_dev,
                              const double *rho_dev, const double *g_dev, double *p_dev, void *stream)
{
    if (!hist_dev || !rho_dev || !g_dev || !p_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (memory < 1 || memory > FL_MAX_MEMORY || recent < 0 || recent >= memory) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    // the stand-alone recursion exists for the register geometries (n <= 4096) only
    if (n > 4096 || fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nw = threads / 64;
#define FL_TL(NW_, EPT_)                                                                                          \
    hipLaunchKernelGGL((fl::two_loop_kernel<NW_, EPT_>), dim3(batch), dim3(NW_ * 64), 0, st, n, memory, recent,    \
                       hist_dev, rho_dev, g_dev, p_dev)
    if (nw == 1 && ept == 2) FL_TL(1, 2);
    else if (nw == 1 && ept == 4) FL_TL(1, 4);
    else if (nw == 1 && ept == 8) FL_TL(1, 8);
    else if (nw == 4 && ept == 4) FL_TL(4, 4);
    else if (nw == 2 && ept == 8) FL_TL(2, 8);
    else if (nw == 8 && ept == 2) FL_TL(8, 2);
    else if (nw == 1 && ept == 16) FL_TL(1, 16);
    else if (nw == 4 && ept == 8) FL_TL(4, 8);
    else if (nw == 8 && ept == 8) FL_TL(8, 8);
    else return FL_ERR_UNSUPPORTED_SIZE;
#undef FL_TL
    return fl::launch_status();
}

} // extern "C"
