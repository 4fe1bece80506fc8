// v_mfma_f64_16x16x4_f64 on this GPU: issue rate in SHADER CYCLES and the clock the chip holds under the loop.
//
// Round 1's version of this probe divided wall time by an assumed 2.4 GHz and called the result a ceiling (47 TFLOP/s)
// -- below the 65 TFLOP/s the shipped BFGS GEMM reaches, so it was no ceiling: a bare f64 MFMA loop makes the chip
// lower its clock (MI355X_MICROARCH.md, "DVFS give-back").  This version separates the two things:
//   * cycles per MFMA per SIMD from s_memtime stamps around the loop (the hardware's issue rate, clock-independent);
//   * the in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz (guide, item 6), median over workgroups;
//   * TFLOP/s from wall time, which is what the other two multiply out to.
// Tiles per wave as in the GEMM kernels: TM x TN accumulators fed by TM different A and TN different B registers.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o tools/mfma_f64_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int TM, int TN> __global__ __launch_bounds__(256) void k(double *out, long long *stamps, int iters, double a0, double b0)
{
    f64x4 acc[TM][TN];
    double a[TM], b[TN];
    for (int i = 0; i < TM; ++i) a[i] = a0 + (threadIdx.x + 64 * i) * 1e-9;
    for (int j = 0; j < TN; ++j) b[j] = b0 - (threadIdx.x + 64 * j) * 1e-9;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) acc[i][j] = f64x4{0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { // stamps go to a buffer of their own: no output value is computed from them
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}
template <int TM, int TN> void run(int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000; // 256-thread workgroups: one wave per SIMD each
    double *out;
    long long *st;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&st, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<TM, TN><<<blocks, 256>>>(out, st, iters, 1.0, 0.5); // warm: the clock settles under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<TM, TN><<<blocks, 256>>>(out, st, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), st, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int b = 0; b < blocks; ++b) {
        cyc.push_back((double)h[2 * b] / ((double)iters * TM * TN));
        ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 16 * 16 * 4 * TM * TN * (double)iters * blocks * 4;
    // a wave issues TM*TN MFMAs per iteration; with w waves per SIMD the SIMD's pipe sees w times that in the same time
    printf("{\"mfma_f64_16x16x4\": {\"tiles_per_wave\": \"%dx%d\", \"waves_per_simd\": %d, \"TFLOPs\": %.2f, "
           "\"cycles_per_mfma_per_wave\": %.1f, \"cycles_per_mfma_per_simd\": %.1f, \"clock_GHz_in_kernel\": %.3f, "
           "\"TFLOPs_at_that_rate_and_clock\": %.2f}}\n",
           TM, TN, waves_per_simd, flop / ms / 1e9, cyc[blocks / 2], cyc[blocks / 2] / waves_per_simd, ghz[blocks / 2],
           2048.0 / (cyc[blocks / 2] / waves_per_simd) * ghz[blocks / 2] * 1024 / 1e3);
    hipFree(out);
    hipFree(st);
}
int main()
{
    run<1, 1>(1);
    run<2, 2>(1);
    run<4, 2>(1);
    run<4, 2>(2); // the GEMM kernels' layout: 8 accumulator tiles per wave, two waves per SIMD
    run<4, 4>(1);
    run<2, 2>(4);
    return 0;
}
