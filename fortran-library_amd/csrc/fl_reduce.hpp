// fl_reduce.hpp -- fixed-order workgroup reductions for the solver kernels (device only).
//
// Order (replayed on the CPU by the tests, oracle FLO_SUM_TREE), per wave of 64 lane partials a[0..63]:
//   b[l] = a[l] + a[l+32]   (l < 32)      v_permlane32_swap
//   c[l] = b[l] + b[l+16]   (l < 16)      v_permlane16_swap
//   d[i] = c[i] + c[15-i]   (i < 8)       DPP row_mirror
//   e[i] = d[i] + d[7-i]    (i < 4)       DPP row_half_mirror
//   f[i] = e[i] + e[i^2]    (i < 2)       DPP quad_perm [2,3,0,1]
//   wave = f[0] + f[1]                    DPP quad_perm [1,0,3,2]
// then the waves left to right.  IEEE addition is commutative, so every lane that takes part ends with bitwise
// the same value whichever operand it holds "first".  Nothing goes through the LDS pipe inside a wave.
//
// Several values are reduced at the price of little more than one: the two swap levels work as a reduce-scatter.
// One swap pair + one add folds the 32-lane step of TWO values (value A ends in lanes 0-31, B in lanes 32-63),
// the same at the 16-lane step, so four values end up in the four rows of ONE register and share the four DPP
// steps: 4 values cost 6 swaps + 8 DPP moves + 7 adds instead of 4 x (4 swaps + 8 moves + 6 adds).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

namespace fl {

template <int CTRL> __device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
// lane i of every row <- lane i-1 of the same row, lane 0 <- +0.0 (row_shr:1, bound_ctrl: out-of-row sources read as zero)
__device__ __forceinline__ double dpp_shr1_zero(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// the four in-row steps; rows may hold different values
__device__ __forceinline__ double row_allreduce(double v)
{
    v = dpp_add<0x140>(v); // row_mirror:       lane i <-> 15-i
    v = dpp_add<0x141>(v); // row_half_mirror:  lane i <-> 7-i
    v = dpp_add<0x4E>(v);  // quad_perm [2,3,0,1]
    v = dpp_add<0xB1>(v);  // quad_perm [1,0,3,2]
    return v;
}
// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]
__device__ __forceinline__ double fold32(double a, double b)
{
    const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
// rows 0,2 <- a.row0 + a.row1, a.row2 + a.row3;  rows 1,3 <- the same of b
__device__ __forceinline__ double fold16(double a, double b)
{
    const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}

// one value, every lane gets the wave's sum
__device__ __forceinline__ double wave_allreduce(double v)
{
    v = fold32(v, v);
    v = fold16(v, v);
    return row_allreduce(v);
}

// up to four values -> one register: value k (k < N) sits in every lane of row group_row(k)
//   N = 1: all rows;  N = 2: rows {0,1} / {2,3};  N = 3, 4: rows 0, 2, 1, 3 for k = 0..3
template <int N> __device__ __forceinline__ double wave_reduce_group(const double *v)
{
    static_assert(N >= 1 && N <= 4, "group of 1..4 values");
    double q;
    if constexpr (N == 1) {
        q = fold32(v[0], v[0]);
        q = fold16(q, q);
    } else if constexpr (N == 2) {
        q = fold32(v[0], v[1]);
        q = fold16(q, q);
    } else {
        const double m = fold32(v[0], v[1]);
        const double o = fold32(v[2], v[N == 4 ? 3 : 2]);
        q = fold16(m, o);
    }
    return row_allreduce(q);
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// workgroup all-reduce of NV values; every thread gets bitwise identical totals
template <int NW> struct Reducer {
    double *slots; // LDS [2][NVMAX][NW]
    int parity;
    static constexpr int NVMAX = 10;
    template <int NV> __device__ __forceinline__ void run(double (&v)[NV])
    {
        constexpr int NG = (NV + 3) / 4;
        double q[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * 4 + 4 <= NV) q[g] = wave_reduce_group<4>(v + g * 4);
            else if constexpr (NV % 4 == 3) q[g] = wave_reduce_group<3>(v + g * 4);
            else if constexpr (NV % 4 == 2) q[g] = wave_reduce_group<2>(v + g * 4);
            else if constexpr (NV % 4 == 1) q[g] = wave_reduce_group<1>(v + g * 4);
        }
        if constexpr (NW > 1) {
            // 512-thread kernels (256 VGPRs per wave): recompute the slot indices here instead of keeping them
            // (or spilling them) across the whole solver loop -- see Geo::tid()
            int tx = (int)(threadIdx.x & (unsigned)(NW * 64 - 1)); // (index inside the problem's group of NW waves: Geo::ltid())
            if constexpr (NW >= 8) asm volatile("" : "+v"(tx));
            const int lane = tx & 63, wave = tx >> 6;
            double *s = slots + parity * (NVMAX * NW);
            // row r of a full group holds value ((r & 1) << 1) | (r >> 1); partial groups: see wave_reduce_group
            const int row = lane >> 4;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int left = NV - g * 4; // values in this group (compile time after unrolling)
                int k;
                if (left >= 3) k = ((row & 1) << 1) | (row >> 1);
                else if (left == 2) k = row >> 1;
                else k = 0;
                const bool owner = (lane & 15) == 0 && k < left && (left >= 3 || (left == 2 ? (row & 1) == 0 : row == 0));
                if (owner) s[(g * 4 + k) * NW + wave] = q[g];
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
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int g = i >> 2, k = i & 3, left = NV - g * 4;
                const int ln = left >= 3 ? ((k & 1) * 32 + (k >> 1) * 16) : (left == 2 ? 32 * k : 0);
                v[i] = read_lane(q[g], ln);
            }
        }
    }
};

} // namespace fl
