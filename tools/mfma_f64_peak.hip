// Back-to-back v_mfma_f64_16x16x4_f64 issue rate on this GPU: the denominator for the "mfma" roofline
// (the FP64 matrix peak is not in the local guides -- measure it).
#include <hip/hip_runtime.h>
#include <cstdio>
using f64x4 = __attribute__((ext_vector_type(4))) double;
template <int NACC> __global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, 100, 1.0, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * 4 * NACC * (double)iters * blocks * 4;
    printf("{\"mfma_f64_16x16x4\": {\"accumulators\": %d, \"waves_per_simd\": %d, \"TFLOPs\": %.2f, \"cycles_per_mfma_at_2.4GHz\": %.1f}}\n",
           NACC, waves_per_simd, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * waves_per_simd));
    hipFree(out);
}
int main()
{
    run<1>(1);
    run<4>(1);
    run<16>(1);
    run<16>(2);
    run<4>(4);
    return 0;
}
