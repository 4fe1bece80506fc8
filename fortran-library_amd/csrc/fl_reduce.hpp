// fl_reduce.hpp -- fixed-order workgroup reductions for the solver kernels (device only).
//
// Order (replayed on the CPU by the tests, oracle FLO_SUM_TREE):
//   lane partial -> 64-lane xor butterfly with offsets 1,2,4,8,16,32 -> waves left to right.
// The butterfly never touches the LDS pipe: offsets 1..8 are DPP moves (quad_perm,
// row_half_mirror, row_mirror -- after the previous steps every lane of a quad / half
// row holds the same value, so a mirror delivers exactly lane^4 / lane^8's value),
// offsets 16 and 32 are gfx950's v_permlane16_swap / v_permlane32_swap.  IEEE addition
// is commutative, so every lane ends with bitwise the same sum as v[l] + v[l^off].
#pragma once
#include <hip/hip_runtime.h>

namespace fl {

template <int CTRL> __device__ __forceinline__ double dpp_xor_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double swap16_add(double v) // lanes l and l^16
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ double swap32_add(double v) // lanes l and l^32
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}

__device__ __forceinline__ double wave_allreduce(double v)
{
    v = dpp_xor_add<0xB1>(v);  // quad_perm:[1,0,3,2]  = lane^1
    v = dpp_xor_add<0x4E>(v);  // quad_perm:[2,3,0,1]  = lane^2
    v = dpp_xor_add<0x141>(v); // row_half_mirror      = lane^4 (quads are uniform)
    v = dpp_xor_add<0x140>(v); // row_mirror           = lane^8 (half rows are uniform)
    v = swap16_add(v);
    v = swap32_add(v);
    return v;
}

// workgroup all-reduce of NV values; every thread gets bitwise identical totals
template <int NW> struct Reducer {
    double *slots; // LDS [2][NVMAX][NW]
    int parity;
    static constexpr int NVMAX = 10;
    template <int NV> __device__ __forceinline__ void run(double (&v)[NV])
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = wave_allreduce(v[i]);
        if constexpr (NW > 1) {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            double *s = slots + parity * (NVMAX * NW);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NV; ++i) s[i * NW + wave] = v[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                double t = s[i * NW];
#pragma unroll
                for (int w = 1; w < NW; ++w) t = t + s[i * NW + w];
                v[i] = t;
            }
            parity ^= 1;
        }
    }
};

} // namespace fl
