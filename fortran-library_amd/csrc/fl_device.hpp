// fl_device.hpp -- device-side building blocks of the batched solvers (gfx950).
//
// Reference semantics: /root/reference/source/NonlinearOptimization.f90 ("NO.f90")
//   SteepestDescent 55-188, ConjugateGradient 193-394 (DY 352-372, PR 373-393),
//   LBFGS 398-625 (pre-iteration 472-510, Before 586-608, After 609-624),
//   BFGS 632-1022 (first step 683-716, After_NoHessian 996-1015),
//   AugmentedLagrangian 2005-2241 (L, Ld, L_Ld 2193-2228),
//   line searchers 1286-1698 (fl_linesearch.hpp).
//
// One workgroup owns one problem.  Its vectors live in registers: thread t of the
// T = 64*NW threads holds EPT elements, dealt in 16-byte chunks round-robin
// (element (c*T + t)*2 + j), so every global access is a coalesced dwordx4 stream.
// The solver is written as a resumable machine: start() / advance() return the next
// evaluation request (FL_REQ_* bits, fl_linesearch.hpp).  The fused kernels answer
// the request with a compiled-in objective and loop; the reverse-communication kernel
// (fl_rci.hip) saves the machine to HBM and lets the caller evaluate.
#pragma once
#ifndef __HIPCC_RTC__ // (run-time compilation, csrc/fl_user_rtc.cpp: hiprtc brings the device runtime and the fixed-width types)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#if __has_include("fl_nlopt.h") // installed layout (prefix/include/fl/...) and run-time compilation: found on the include path
#include "fl_nlopt.h"
#else
#include "../../include/fl_nlopt.h"
#endif
#include "fl_linesearch.hpp"
#include "fl_reduce.hpp"

#define FL_REQ_NOMOVE 8 // evaluate at the current x (do not form x0 + a p)
#define FL_REQ_H 16     // reverse communication: the Hessian at the current x is wanted (fdd, NO.f90:37)
#define FL_REQ_C 32     // reverse communication, augmented Lagrangian: c(x) is wanted (subroutine c, NO.f90:1928)
#define FL_REQ_CD 64    // ... and the constraint Jacobian cd(x) (subroutine cd: cdx(N,M), NO.f90:1931)
#define FL_MAX_CONSTRAINTS 16

namespace fl {

struct SolveArgs {
    int n, batch, mem, maxit, strong, fused, cg_method;
    int exact_step; // BFGS ExactStep (NO.f90:630): > 0 = exact inverse Hessian every exact_step iterations
    double tol, minstep, c1, c2, incr;
    double *x;
    const double *d, *b;
    double *hist; // L-BFGS ring [batch][2*mem][NPAD]  |  BFGS inverse Hessian [batch][n][NPAD]
    double *f_out, *gg_out;
    int *iters, *status, *nf, *ng;
    // augmented Lagrangian (NO.f90:2005): m block-sphere constraints
    int aug_m;
    double miu0, precision; // outer tolerance = Precision (unsquared, NO.f90:2047, 2072)
    double *lambda;         // [batch][aug_m] in/out
    int *outer;
    double *cnorm2;
    const void *user; // a caller-compiled objective's own data (include/fl_user_objective.hpp); not read by the built-in ones
    // Staged launches of the augmented Lagrangian (fl_solver_kernels.hip: launch_aug_staged).  A problem may PAUSE at an outer
    // iteration's boundary -- where the reference starts a fresh inner solve from (x, lambda, miu) anyway -- and be resumed by
    // a later launch with more waves per problem; what a resumed problem needs beyond x and lambda is in pstate.
    const int *list;    // this launch's problems (NULL: blockIdx.x); their number is sched[1]
    int *sched;         // [0] problems finished so far (all launches), [1] problems listed for this launch
    double *pstate;     // [batch][PSTATE]: miu, c.c, (outer, inner iterations, nf, ng as two doubles' worth of ints)
    int pause_below;    // pause when at most this many problems of the batch are unfinished (0: never)
    int resume;         // problems of this launch continue from pstate
    int pause_grid;     // workgroups of a listed launch (an upper bound of sched[1])
};
#define FL_STATUS_PAUSED 3 // internal: never seen by a caller (every staged sequence ends with a launch that cannot pause)
#define FL_PSTATE 4

template <int NW, int EPT> struct Geo {
    static constexpr int T = NW * 64;
    static constexpr int NPAD = T * EPT;
    static constexpr int NCH = EPT / 2;
    // The thread's index INSIDE ITS PROBLEM'S GROUP of T threads.  A workgroup normally is one such group (the mask folds
    // away: the compiler knows threadIdx.x < T from the launch bounds); the replicated kernels (fl_solve_rep_kernel) run
    // several identical groups per workgroup, each a full copy of the machine with LDS of its own.
    __device__ __forceinline__ static int ltid() { return (int)(threadIdx.x & (unsigned)(T - 1)); }
    // The thread index behind an empty asm: index arithmetic derived from it is then redone where it is used
    // (one or two integer instructions) instead of being hoisted out of the solver's main loop and kept -- or
    // spilled -- for the whole kernel.  Measured on the code objects: fl_solve_kernel<2,8,DIAGQUAD,LBFGS> 217 -> 211
    // VGPRs, the 4x8 BFGS kernels 12-44 spilled VGPRs -> 0.
    __device__ __forceinline__ static int tid()
    {
        int t = ltid();
        asm volatile("" : "+v"(t));
        return t;
    }
    __device__ __forceinline__ static int e0(int c) { return ((c * T + tid()) << 1); }
};

__device__ __forceinline__ double uni(double v) { return LineSearch::uni(v); }

// ---- global <-> register vectors.  User arrays are [batch][n] (row stride n);
// rows are 16-byte aligned iff n is even.
template <int NW, int EPT> __device__ __forceinline__ void load_user(const double *row, int n, double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
    const bool vec = (n & 1) == 0;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
        const int e = G::e0(c);
        if (vec && e + 1 < n) {
            const double2 t = *reinterpret_cast<const double2 *>(row + e);
            v[2 * c] = t.x;
            v[2 * c + 1] = t.y;
        } else {
            v[2 * c] = (e < n) ? row[e] : 0.0;
            v[2 * c + 1] = (e + 1 < n) ? row[e + 1] : 0.0;
        }
    }
}
template <int NW, int EPT> __device__ __forceinline__ void store_user(double *row, int n, const double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
    const bool vec = (n & 1) == 0;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
        const int e = G::e0(c);
        if (vec && e + 1 < n) {
            *reinterpret_cast<double2 *>(row + e) = make_double2(v[2 * c], v[2 * c + 1]);
        } else {
            if (e < n) row[e] = v[2 * c];
            if (e + 1 < n) row[e + 1] = v[2 * c + 1];
        }
    }
}
// padded rows (NPAD doubles, 16-byte aligned): HBM workspaces and LDS parking
// FL_SADDR (geometries of at most FL_SADDR waves): the row pointer is workgroup-uniform, so the address is written as
// (uniform base) + (32-bit byte offset of the thread) + (chunk offset, a constant) -- the form the `global_load ... v_off,
// s[base]` addressing takes with ONE kept offset register and no vector arithmetic per load; the laundered index of
// Geo::tid() costs four vector instructions per load (32 per two-loop step of the headline kernel).
#ifndef FL_SADDR
#define FL_SADDR 2
#endif
template <int NW, int EPT> __device__ __forceinline__ void load_pad(const double *row, double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
    if constexpr (NW <= FL_SADDR) {
        const unsigned off = (unsigned)G::ltid() * 16u;
        const char *base = reinterpret_cast<const char *>(row);
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const double2 t = *reinterpret_cast<const double2 *>(base + (off + (unsigned)(c * G::T * 16)));
            v[2 * c] = t.x;
            v[2 * c + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const double2 t = *reinterpret_cast<const double2 *>(row + G::e0(c));
            v[2 * c] = t.x;
            v[2 * c + 1] = t.y;
        }
    }
}
// the same for a row that is read ONCE per pass and not again before the whole matrix has gone by (the dense inverse Hessian
// of BFGS in its deferred form, n^2 doubles per problem against an L2 of 4 MB per XCD): NON-TEMPORAL accesses.  FL_BFGS_NT: 1 = the
// loads of the product H (y, g), 2 = + the fold's loads, 3 = + the fold's stores.  BASELINE config 4 (1024 x n = 4096, 20
// iterations; two rounds each): 0: 376.8 / 382.1 ms, 1: 358.0 / 361.8, 2: 356.8 / 361.2, 3: 355.7 / 356.3
// (profiles/r04/c4_nontemporal_ab.txt; with it, three and four columns in flight stay behind two: 360 / 366 ms)
#ifndef FL_BFGS_NT
#define FL_BFGS_NT 3
#endif
template <int NW, int EPT> __device__ __forceinline__ void load_pad_nt(const double *row, double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
    typedef double fl_d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
        const fl_d2 t = __builtin_nontemporal_load(reinterpret_cast<const fl_d2 *>(row + G::e0(c)));
        v[2 * c] = t.x;
        v[2 * c + 1] = t.y;
    }
}
template <int NW, int EPT> __device__ __forceinline__ void store_pad_stream(double *row, const double (&v)[EPT]); // (below store_pad)
template <int NW, int EPT> __device__ __forceinline__ void load_pad_stream(const double *row, double (&v)[EPT])
{
#if FL_BFGS_NT
    load_pad_nt<NW, EPT>(row, v);
#else
    load_pad<NW, EPT>(row, v);
#endif
}
template <int NW, int EPT> __device__ __forceinline__ void store_pad(double *row, const double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
    if constexpr (NW <= FL_SADDR) {
        const unsigned off = (unsigned)G::ltid() * 16u;
        char *base = reinterpret_cast<char *>(row);
#pragma unroll
        for (int c = 0; c < G::NCH; ++c)
            *reinterpret_cast<double2 *>(base + (off + (unsigned)(c * G::T * 16))) = make_double2(v[2 * c], v[2 * c + 1]);
    } else {
#pragma unroll
        for (int c = 0; c < G::NCH; ++c)
            *reinterpret_cast<double2 *>(row + G::e0(c)) = make_double2(v[2 * c], v[2 * c + 1]);
    }
}
template <int NW, int EPT> __device__ __forceinline__ void store_pad_stream(double *row, const double (&v)[EPT])
{
#if FL_BFGS_NT >= 3
    using G = Geo<NW, EPT>;
    typedef double fl_d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
        fl_d2 t;
        t.x = v[2 * c];
        t.y = v[2 * c + 1];
        __builtin_nontemporal_store(t, reinterpret_cast<fl_d2 *>(row + G::e0(c)));
    }
#else
    store_pad<NW, EPT>(row, v);
#endif
}

template <int EPT> __device__ __forceinline__ double dot_part(const double (&a)[EPT], const double (&b)[EPT])
{
    double acc = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < EPT; ++k) acc = acc + a[k] * b[k];
    return acc;
}

// ------------------------------------------------------------ objectives
// eval(): gradient of the thread's EPT elements and up to two partial sums
// (f = combine(S0, S1)).  Padded elements (e >= n) carry x = 0 and must give g = 0
// and zero terms.
template <int OBJ, int NW, int EPT> struct Objective;

template <int NW, int EPT> struct Objective<FL_OBJ_QUARTIC, NW, EPT> { // test/test.f90:630-663
    static constexpr int LDS_DOUBLES = 0;
    __device__ __forceinline__ void init(const SolveArgs &, int, double *) {}
    __device__ __forceinline__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int,
                                         double *)
    {
        s1 = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const double x3 = x[k] * x[k] * x[k]; // x**3 = (x*x)*x
            const double t = x3 * x[k];           // x**4 = ((x*x)*x)*x
            g[k] = 4.0 * x3;
            s0 = (k == 0) ? t : s0 + t;
        }
    }
    __device__ __forceinline__ static double combine(double s0, double) { return s0; }
    // column j of f''(x) for the thread's rows: diag(12 x^2) (test/test.f90:665-675)
    __device__ __forceinline__ void hess_column(int j, const double (&x)[EPT], int, const double *, double (&h)[EPT])
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            h[k] = (Geo<NW, EPT>::e0(k >> 1) + (k & 1) == j) ? 12.0 * x[k] * x[k] : 0.0;
    }
};

// d, b of the diagonal quadratic: in registers (EPT <= 4), or -- 8 elements per thread, up to 256 threads -- in two
// LDS rows that every thread reads back only where it wrote (no barrier): 32 VGPRs less for the whole solve, which
// takes the L-BFGS kernel for n = 1024 from 181 to under 168 VGPRs = 3 waves per SIMD instead of 2
// (profiles/r02/quad_lds_ab.txt).  The 512-thread kernels have no LDS to spare: they re-read d, b per evaluation
// (Solver::LEAN).
#ifndef FL_QUAD_LDS
#define FL_QUAD_LDS 0
#endif
template <int NW, int EPT> struct Objective<FL_OBJ_DIAGQUAD, NW, EPT> { // f=0.5*sum(d*x*x)-sum(b*x), g=d*x-b
    using G = Geo<NW, EPT>;
    static constexpr bool IN_LDS = FL_QUAD_LDS && EPT >= 8 && NW <= 4;
    static constexpr int LDS_DOUBLES = IN_LDS ? 2 * G::NPAD : 0;
    double d[IN_LDS ? 1 : EPT], b[IN_LDS ? 1 : EPT];
    __device__ __forceinline__ void init(const SolveArgs &A, int prob, double *xs)
    {
        if constexpr (IN_LDS) {
            double t[EPT];
            load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, t);
            store_pad<NW, EPT>(xs, t);
            load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, t);
            store_pad<NW, EPT>(xs + G::NPAD, t);
        } else {
            load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, d);
            load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, b);
        }
    }
    __device__ __forceinline__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int,
                                         double *xs)
    {
        double dl[EPT], bl[EPT];
        if constexpr (IN_LDS) {
            load_pad<NW, EPT>(xs, dl);
            load_pad<NW, EPT>(xs + G::NPAD, bl);
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const double dk = IN_LDS ? dl[k] : d[IN_LDS ? 0 : k], bk = IN_LDS ? bl[k] : b[IN_LDS ? 0 : k];
            const double dx = dk * x[k];
            const double t0 = dx * x[k], t1 = bk * x[k];
            g[k] = dx - bk;
            s0 = (k == 0) ? t0 : s0 + t0;
            s1 = (k == 0) ? t1 : s1 + t1;
        }
    }
    __device__ __forceinline__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
    __device__ __forceinline__ void hess_column(int j, const double (&)[EPT], int, const double *xs, double (&h)[EPT])
    {
        double dl[EPT];
        if constexpr (IN_LDS) load_pad<NW, EPT>(xs, dl);
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            h[k] = (G::e0(k >> 1) + (k & 1) == j) ? (IN_LDS ? dl[k] : d[IN_LDS ? 0 : k]) : 0.0;
    }
};

template <int NW, int EPT> struct Objective<FL_OBJ_ROSENBROCK, NW, EPT> {
    // chained Rosenbrock; x is staged through LDS (one halo element each side) so
    // that every thread can read its chunk's neighbours x[e-1], x[e+2]
    using G = Geo<NW, EPT>;
    static constexpr int LDS_DOUBLES = G::NPAD + 2;
    __device__ __forceinline__ void init(const SolveArgs &, int, double *xs)
    {
        if (G::ltid() == 0) {
            xs[0] = 0.0;
            xs[G::NPAD + 1] = 0.0;
        }
    }
    __device__ __forceinline__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int n,
                                         double *xs)
    {
        s1 = 0.0;
        __syncthreads(); // previous trial's neighbour reads are complete
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const int e = G::e0(c);
            xs[1 + e] = x[2 * c];
            xs[2 + e] = x[2 * c + 1];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const int e = G::e0(c);
            const double xa = x[2 * c], xb = x[2 * c + 1];
            const double xl = xs[e], xr = xs[e + 3]; // x[e-1], x[e+2]
            const double ul = xa - xl * xl;          // u_{e-1}
            const double ua = xb - xa * xa;          // u_e
            const double ub = xr - xb * xb;          // u_{e+1}
            const double va = 1.0 - xa, vb = 1.0 - xb;
            const double A_a = (e >= 1) ? 200.0 * ul : 0.0;
            const double A_b = 200.0 * ua;
            double ta = 0.0, tb = 0.0, ga = 0.0, gb = 0.0;
            if (e <= n - 2) {
                ta = 100.0 * (ua * ua) + va * va;
                ga = A_a - 400.0 * xa * ua - 2.0 * va;
            } else if (e == n - 1) {
                ga = A_a;
            }
            if (e + 1 <= n - 2) {
                tb = 100.0 * (ub * ub) + vb * vb;
                gb = A_b - 400.0 * xb * ub - 2.0 * vb;
            } else if (e + 1 == n - 1) {
                gb = A_b;
            }
            g[2 * c] = ga;
            g[2 * c + 1] = gb;
            s0 = (c == 0) ? ta : s0 + ta;
            s0 = s0 + tb;
        }
    }
    __device__ __forceinline__ static double combine(double s0, double) { return s0; }
    // column j of the tridiagonal Hessian for the thread's rows; xs = the LDS image of x left by eval()
    //   H(i,i) = [200 if i>=1] + (1200 x_i^2 - 400 x_{i+1} + 2 if i<=n-2),  H(i,i+1) = H(i+1,i) = -400 x_i
    __device__ __forceinline__ void hess_column(int j, const double (&x)[EPT], int n, const double *xs,
                                                double (&h)[EPT])
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = G::e0(k >> 1) + (k & 1);
            double v = 0.0;
            if (i < n) {
                if (i == j) {
                    if (i >= 1) v = 200.0;
                    if (i <= n - 2) v = v + (1200.0 * x[k] * x[k] - 400.0 * xs[2 + i] + 2.0);
                } else if (i == j - 1) {
                    v = -400.0 * x[k];
                } else if (i == j + 1) {
                    v = -400.0 * xs[i]; // x[i-1]
                }
            }
            h[k] = v;
        }
    }
};

// objective evaluated by the caller (reverse communication, fl_rci.hip): nothing compiled in
#define FL_OBJ_EXTERNAL 3
template <int NW, int EPT> struct Objective<FL_OBJ_EXTERNAL, NW, EPT> {
    static constexpr int LDS_DOUBLES = 0;
    __device__ __forceinline__ void init(const SolveArgs &, int, double *) {}
    __device__ __forceinline__ void eval(const double (&)[EPT], double (&)[EPT], double &s0, double &s1, int, double *)
    {
        s0 = s1 = 0.0;
    }
    __device__ __forceinline__ static double combine(double s0, double) { return s0; }
    __device__ __forceinline__ void hess_column(int, const double (&)[EPT], int, const double *, double (&h)[EPT])
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) h[k] = 0.0; // no Hessian by reverse communication (exact_step <= 0 only)
    }
};

// objective compiled in by the CALLER (include/fl_user_objective.hpp): a class template with the interface of the
// specialisations above -- LDS_DOUBLES, init(), eval(), combine() -- named by the macro FL_USER_OBJECTIVE before this
// header is included.  It gets the fused kernel, its geometry and its scheduling (wave priority, on-chip ring pairs, lazy
// g.g), which the reverse-communication form (FL_OBJ_EXTERNAL) cannot have.
#define FL_OBJ_USER 4
#ifdef FL_USER_OBJECTIVE
template <int NW, int EPT> struct Objective<FL_OBJ_USER, NW, EPT> : public FL_USER_OBJECTIVE<NW, EPT> {};
#endif
// whose register budget an objective's kernels are tuned like (occupancy caps, x0 in LDS, ...): the built-in ones like
// themselves; a caller's objective like FL_USER_TUNE_LIKE -- by default like none of them (no caps: nothing can spill),
// FL_OBJ_DIAGQUAD for an element-wise objective that keeps at most two data vectors in registers
#ifndef FL_USER_TUNE_LIKE
#define FL_USER_TUNE_LIKE FL_OBJ_USER
#endif
template <int OBJ> constexpr int tuned_like() { return OBJ == FL_OBJ_USER ? FL_USER_TUNE_LIKE : OBJ; }

// CONSTRAINTS compiled in by the caller (AUG = FL_AUG_USER; include/fl_user_objective.hpp, fl_user_compile_auglag): the
// reference's callbacks c(cx,x,M,N) and cd(cdx,x,M,N) (NO.f90:1928-1934) as a class template named by FL_USER_CONSTRAINTS
//     template <int NW, int EPT> struct MyConstraints {
//         __device__ void init(const fl::SolveArgs &A, int prob);      // A.aug_m = M (<= 8), A.user: the caller's data
//         // the thread's share of every constraint: c_j(x) = (sum over all threads of cpart[j]) + offset(j)
//         // (elements beyond n carry x = 0 and must contribute nothing; the constant of c_j -- e.g. -1 -- is offset(j),
//         //  added AFTER the workgroup's sum)
//         __device__ void partial(const double (&x)[EPT], double (&cpart)[8], int n);
//         __device__ double offset(int j) const;
//         // g[k] += sum_j v[j] * dc_j/dx_k for the thread's elements   (v = miu c - lambda: Ld, NO.f90:2205)
//         __device__ void add_gradient(const double (&x)[EPT], const double (&v)[8], double (&g)[EPT], int n);
//     };
// The kernel sums cpart over the workgroup in its fixed order, in the same reduction as the objective's two sums.
#define FL_AUG_USER 2
#define FL_USER_MAX_CONSTRAINTS 8
#ifdef FL_USER_CONSTRAINTS
template <int NW, int EPT> struct UserConstraints : public FL_USER_CONSTRAINTS<NW, EPT> {};
#else
template <int NW, int EPT> struct UserConstraints { // (never used: no kernel with AUG = FL_AUG_USER is instantiated without the macro)
    __device__ __forceinline__ void init(const SolveArgs &, int) {}
    __device__ __forceinline__ void partial(const double (&)[EPT], double (&)[8], int) {}
    __device__ __forceinline__ double offset(int) const { return 0.0; }
    __device__ __forceinline__ void add_gradient(const double (&)[EPT], const double (&)[8], double (&)[EPT], int) {}
};
#endif
struct NoConstraints {};
template <bool B, class T, class F> struct pick_type { using type = T; }; // (std::conditional: not there under hiprtc)
template <class T, class F> struct pick_type<false, T, F> { using type = F; };

} // namespace fl
#include "fl_dense.hpp"
namespace fl {

// ------------------------------------------------------------ the solver machine
// METHOD: FL_SOLVER_SD | FL_SOLVER_CG | FL_SOLVER_LBFGS | FL_SOLVER_BFGS | FL_SOLVER_NEWTON.
// AUG = 1 wraps the objective in the augmented Lagrangian with aug_m block-sphere
// constraints c_j = sum_{i in block j} x_i^2 - 1 (blocks of n/aug_m) and runs the
// reference's outer loop (NO.f90:2150-2185) around the inner solver.
// EXACT (BFGS only): the exact-Hessian refresh (ExactStep > 0, NO.f90:674-682, 949-956) is compiled in.  The fused
// kernels are instantiated both ways and the host picks by opt->exact_step, so that the ExactStep <= 0 kernel -- pure
// quasi-Newton updates, BASELINE config 4 -- does not carry the Cholesky kernels' registers.
template <int NW, int EPT, int OBJ, int METHOD, int AUG, int EXACT = (METHOD == FL_SOLVER_BFGS)> struct Solver {
    using G = Geo<NW, EPT>;
    using Obj = Objective<OBJ, NW, EPT>;
    static constexpr int NPAD = G::NPAD;
#ifndef FL_BF_UNROLL
#define FL_BF_UNROLL 4
#endif
    // columns of H in flight per thread in the streaming passes.  512-thread kernels (BASELINE config 4): two -- six interleaved
    // repetitions each, profiles/r04/c4_unroll_ab.txt: 2 columns 370.4 ms, 4 columns 376.3, 8 columns 377.7 (round 3 had read the
    // same 4 % / 2 % as noise from two repetitions)
#ifndef FL_BF_UNROLL_W8
#define FL_BF_UNROLL_W8 2
#endif
    static constexpr int BF_UNROLL = NW >= 8 ? FL_BF_UNROLL_W8 : FL_BF_UNROLL;
    static constexpr bool NEEDS_G0 = (METHOD != FL_SOLVER_SD && METHOD != FL_SOLVER_NEWTON);
    // LDS carve (doubles)
    static constexpr int L_RED = 0;                              // [2][NVMAX][NW]
    static constexpr int L_RHO = L_RED + 2 * Reducer<NW>::NVMAX * NW;
    static constexpr int L_ALPHA = L_RHO + FL_MAX_MEMORY;
    static constexpr int L_LAM = L_ALPHA + FL_MAX_MEMORY;        // lambda[FL_MAX_CONSTRAINTS]
    static constexpr int L_CX = L_LAM + FL_MAX_CONSTRAINTS;      // c(x)[2][FL_MAX_CONSTRAINTS]
    // scratch of the Cholesky kernels (pivot and multipliers of a block: 2 + BW doubles).  Its own slot: with the
    // augmented Lagrangian around NewtonRaphson / exact BFGS, c(x) must survive the factorisation (inner_finished reads
    // it when the inner solver stops on MaxIteration right after a direction)
    // SPEC_K > 1: the objective-only shrink / grow loops of the line searchers (StrongWolfe NO.f90:1517-1521 and its
    // _fdwithf twin 1636-1640; Wolfe 1308-1313, 1325-1329) walk a_k = a_{k-1} / incrmt (or * incrmt) and only their EXIT
    // depends on the objective values.  The fused augmented-Lagrangian kernels run such a loop as a tight loop of their
    // own (fast_forward) instead of one pass through the whole machine per trial: SPEC_K consecutive step lengths are
    // evaluated per pass -- SPEC_K independent arithmetic chains, the reductions of all of them sharing one tree pass
    // (fl_reduce.hpp: four values cost little more than one) -- the exit test of the reference is applied to them in
    // order, what lies behind the first exit is discarded, and the machine is handed the exit trial with the state it
    // would have reached by itself.  Every value has the bits of a trial evaluated on its own, and nf counts what the
    // reference would have called.  C5 (45 objective-only trials per gradient, one wave per problem, a tail of long
    // problems on a mostly idle chip): profiles/r03/c5_spec_ab.txt.
#ifndef FL_SPEC_K
#define FL_SPEC_K 4
#endif
#ifndef FL_SPEC_X0_RELOAD
#define FL_SPEC_X0_RELOAD 0
#endif
    // (a caller's objective that declares itself element-wise -- tuned like the diagonal quadratic -- takes part too)
    static constexpr int SPEC_K = (AUG == 1 && (OBJ == FL_OBJ_DIAGQUAD || OBJ == FL_OBJ_QUARTIC || (OBJ == FL_OBJ_USER && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD)) &&
                                   NW < 8 && FL_SPEC_K > 1) ? FL_SPEC_K : 1;
    static_assert(SPEC_K == 1 || SPEC_K == 2 || SPEC_K == 4, "1, 2 or 4 speculative trials");
    // the kernels that take part in staged launches (pause at an outer iteration's boundary / resume: fl_solver_kernels.hip,
    // launch_aug_staged) -- those that have helper-wave forms; the others do not carry the code
    static constexpr bool STAGED = AUG && NW == 1 && SPEC_K > 1 && OBJ != FL_OBJ_EXTERNAL && (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_CG);
    static constexpr int L_CXS = L_CX + 2 * FL_MAX_CONSTRAINTS;  // c(x) of the speculative trials [2][SPEC_K][FL_MAX_CONSTRAINTS]
    // ... followed by one slot per thread that takes the stores of the lanes that own no block sum (evaluate_spec stores
    // from every lane, to a selected address: no exec-mask region splits the pass)
    static constexpr int L_SLOT = L_CXS + (SPEC_K > 1 ? 2 * SPEC_K * FL_MAX_CONSTRAINTS + G::T : 0); // (c(x) is double buffered: cx_ptr())
    static constexpr int L_XS = (L_SLOT + 16 + 1) & ~1;
    static constexpr int L_G0 = (L_XS + (Obj::LDS_DOUBLES > 0 ? Obj::LDS_DOUBLES : 0) + 1) & ~1;
    // BFGS: s, q, g broadcast arrays; the first one doubles as the g_old parking slot (the
    // broadcast arrays are only live inside direction_bfgs, g_old only outside it)
    // L-BFGS, fused kernels only: the LDS_PAIRS newest (s, y) pairs also live in LDS (every thread reads back only
    // the elements it wrote -- the rows are private spill space, no barrier involved), so the recursion fetches
    // 4*(cnt - LDS_PAIRS) - 2 rows from HBM instead of 4*cnt - 4.  Sized to leave the occupancy the registers
    // allow (2 waves / SIMD = 8/NW workgroups per CU, 20*NW KiB of LDS each; a pair takes NW*EPT KiB).
    // g_old is parked in the y row of the slot the next pair will overwrite.
#ifndef FL_LDS_PAIRS_E8
#define FL_LDS_PAIRS_E8 2
#endif
    // measured (profiles/r01/lds_pairs_ab.txt): n=1024 (2x8) 226.5 -> 220.2 ms with 2 pairs; the small geometries
    // lose occupancy to the LDS footprint (n=256, 1x4: 3.26 ms with 0 pairs, 4.08 with 2, 3.56 with 4) -> 0 there
#ifndef FL_LDS_PAIRS_E4
#define FL_LDS_PAIRS_E4 0
#endif
#ifndef FL_LDS_PAIRS_E2
#define FL_LDS_PAIRS_E2 0
#endif
#define FL_LDS_PAIRS(NW_, EPT_) ((EPT_) >= 8 ? FL_LDS_PAIRS_E8 : ((EPT_) == 4 ? FL_LDS_PAIRS_E4 : FL_LDS_PAIRS_E2))
#ifndef FL_LDS_BUDGET
#define FL_LDS_BUDGET 2560 // doubles of LDS per wave: 20 KiB, what 2 waves / SIMD (8 waves per CU) leave each
#endif
    static constexpr int LDS_PAIRS_FIT = (FL_LDS_BUDGET * NW - L_G0) / (2 * NPAD);
    // AUG_LEAN18: the augmented-Lagrangian L-BFGS / CG kernels of the one-wave geometry (256 < n <= 512, C5) are bound by the
    // latency of their trial chain (~75 trials per inner iteration, one two-loop), not by ring traffic: they give up the
    // on-chip pairs, the row prefetch of the recursion (direction_lbfgs_plain) and the register copy of x0 for a third
    // wave per SIMD (208 -> <= 168 VGPRs).
#ifndef FL_AUG_LEAN18
#define FL_AUG_LEAN18 1
#endif
    static constexpr bool AUG_LEAN18 = FL_AUG_LEAN18 && AUG && NW == 1 && EPT == 8 && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD &&
                                       (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_CG);
    static constexpr int LDS_PAIRS_WANT = (METHOD == FL_SOLVER_LBFGS && OBJ != FL_OBJ_EXTERNAL && !AUG_LEAN18) ? FL_LDS_PAIRS(NW, EPT) : 0;
    static constexpr int LDS_PAIRS = LDS_PAIRS_WANT < LDS_PAIRS_FIT ? LDS_PAIRS_WANT : LDS_PAIRS_FIT;
    static constexpr int L_BF = L_G0 + ((NEEDS_G0 && METHOD != FL_SOLVER_BFGS) ? (LDS_PAIRS > 0 ? 2 * LDS_PAIRS * NPAD : NPAD) : 0);
    // ... and the REG_PAIRS newest pairs stay in REGISTERS from one iteration to the next (the LDS ring then holds the
    // LDS_PAIRS pairs after them).  The n = 1024 kernel runs 4 problems per CU either way (2 waves / SIMD by
    // registers, 40 KiB of LDS each): what limits it is the ring traffic -- 0, 2 LDS pairs: 242, 196 ms per solve
    // (profiles/r02/ab_pairs.txt) -- and its 181 VGPRs leave 75 of the 256 idle.
#ifndef FL_REG_PAIRS_E8
#define FL_REG_PAIRS_E8 1 // measured (profiles/r02/ab_pairs.txt): 0, 1, 2 register pairs: 198, 176, 179 ms per solve
#endif
#ifndef FL_REG_PAIRS_E4
#define FL_REG_PAIRS_E4 2 // n = 256 (1x4): 0 / 2 / 3 / 4 pairs 91.8 / 101.3 / 86.8 / 87.0 M it/s; n = 512 (2x4): 51.0 / 56.2 / 57.6 / 56.0
#endif
#ifndef FL_REG_PAIRS_E8W
#define FL_REG_PAIRS_E8W 1 // 4 x 8 and 8 x 8 (n > 1024): n = 2048: 0 / 1 / 2 pairs 110.0 / 101.1 / 103.2 ms, n = 4096: 153.1 / 142.4 / 144.4
#endif
    static constexpr int REG_PAIRS = (METHOD == FL_SOLVER_LBFGS && OBJ != FL_OBJ_EXTERNAL && !AUG)
                                         ? (EPT >= 8 ? (NW <= 2 ? FL_REG_PAIRS_E8 : FL_REG_PAIRS_E8W) : (EPT == 4 ? FL_REG_PAIRS_E4 : 0)) : 0;
    static_assert(REG_PAIRS >= 0 && REG_PAIRS <= 4, "0 .. 4 register pairs");
    // Newton: one row buffer for the Cholesky kernels (BFGS reuses its broadcast arrays)
    static constexpr int L_DEF = L_BF + (METHOD == FL_SOLVER_BFGS ? 3 * NPAD : (METHOD == FL_SOLVER_NEWTON ? NPAD : 0));
    // BFGS for n > 128 (until round 4: n > 1024), fused kernels: the rank-2 updates are DEFERRED -- H is left alone for BF_DEFER iterations
    // (one read pass per iteration instead of a read pass and a read+write pass), then the pending updates are folded
    // in together (direction_bfgs_deferred).  L_DEF: rho_l, cs_l of the pending updates.
#ifndef FL_BFGS_DEFER
#define FL_BFGS_DEFER 8
#endif
#ifndef FL_BFGS_DEFER_NPAD
// padded length from which the updates are deferred (also sizes the workspace: fl_workspace_bytes_for; fl_bfgs_deferred_updates tells
// the caller).  Rounds 2-3: 2048.  Round 4: 256 = every n > 128 -- 20 iterations of 4096 / 8192 / 16 384 problems of n = 1024 / 512 /
// 256: 385 -> 106 ms, 190 -> 51, 94 -> 26 (the first 8 iterations move no H at all while the identity is implicit; in the steady
// state 8 n^2 + 16 n^2 / 8 bytes per iteration against 24 n^2: 2.4 x).  Not for n <= 128 (one 1 x 2 wave per problem; the
// reference's own test problem n = 10: 5.1 -> 6.3 ms with the pending updates' corrections) -- profiles/r04/bfgs_deferral_by_n.txt
#define FL_BFGS_DEFER_NPAD 256
#endif
    static constexpr int BF_DEFER = (METHOD == FL_SOLVER_BFGS && OBJ != FL_OBJ_EXTERNAL && NPAD >= FL_BFGS_DEFER_NPAD) ? FL_BFGS_DEFER : 0;
    static constexpr int BF_FOLD_COLS = NPAD >= 2048 ? 128 : NPAD / (2 * FL_BFGS_DEFER); // columns whose s_l[j], q_l[j] are staged in LDS at a time while folding
    // SD / CG on the diagonal quadratics at 8 elements per thread (C3: no history, nothing streams from HBM, the trials'
    // latency is all there is): the line search's x0 waits in an LDS row of its own -- one read per trial, every thread
    // reads back what it wrote -- which brings the kernel from 141 to <= 128 VGPRs = 4 waves per SIMD instead of 3
    // (fl_solver_kernels.hip: min_waves_per_simd).
#ifndef FL_X0_LDS
#define FL_X0_LDS 1
#endif
    static constexpr bool X0_LDS = AUG_LEAN18 || (FL_X0_LDS && (METHOD == FL_SOLVER_SD || METHOD == FL_SOLVER_CG) && !AUG &&
                                   tuned_like<OBJ>() == FL_OBJ_DIAGQUAD && EPT == 8 && NW >= 2); // (one wave per problem: 44 spills under the cap)
    static constexpr int L_X0 = L_DEF + 2 * BF_DEFER;
    static constexpr int LDS_TOTAL = L_X0 + (X0_LDS ? NPAD : 0);
    using DN = Dense<NW, EPT>;

    const SolveArgs &A;
    double *lds;
    int prob, n;
    Reducer<NW> R;
    Obj obj;
    double x[EPT], g[EPT], p[EPT], x0[X0_LDS ? 1 : EPT];
    // uniform scalars
    double fnew, gg, pp, phid, phidold, a;
    int iters, nf, ng, status, phase, pending;
    int recent, cnt;    // L-BFGS ring
    int lrec;           // slot of the newest pair in the LDS ring (LDS_PAIRS > 0)
    double rs_[REG_PAIRS > 0 ? REG_PAIRS : 1][EPT], ry_[REG_PAIRS > 0 ? REG_PAIRS : 1][EPT]; // register pairs, newest first
    int main_it, h_valid; // BFGS / Newton: main-loop iteration counter (iIteration), inverse Hessian initialised
    int ndef, h_ident;    // deferred BFGS: pending updates; H is still the implicit a_id * I of the first step
    double a_id;
    int hess_stage;       // reverse communication: where to resume once the caller has supplied the Hessian
    static constexpr bool HESS_RCI = (OBJ == FL_OBJ_EXTERNAL);
    // Reverse communication, SD / CG / L-BFGS: of the parked rows [p, x0, g_old, g] a step moves only what it must.  p and
    // x0 change when a line search BEGINS (stored then: ls_begun), g_old is written to and read from its HBM row directly
    // where the machine uses it (g0_park() below), and g never has to survive a step: every line search ends on a
    // gradient request, and the direction that uses that gradient is formed in the very step that takes it.  A trial
    // step then loads x, p, x0 and the caller's g and stores the next trial point: 5 rows instead of 11.
    static constexpr bool RCI_LAZY = (OBJ == FL_OBJ_EXTERNAL) && !AUG &&
                                     (METHOD == FL_SOLVER_SD || METHOD == FL_SOLVER_CG || METHOD == FL_SOLVER_LBFGS);
    double *rci_vec;      // this problem's parked rows (set by the step kernel before load())
    bool ls_begun;        // a line search began in this step
    double yy_recent, rho_recent;
    LineSearch ls;
    // augmented Lagrangian
    double miu, cc;
    int outer_it, inner_iters_total;
    // constraint block of each of the thread's elements (-1 = padding), one signed byte each: four to a register (they are
    // live for the whole solve; eight registers of them were what kept four speculative trials from three waves per SIMD)
    // (not in the 512-thread kernels: there the unpacking's temporaries are what spills)
    static constexpr int BLK_PER = (NW < 8) ? 4 : 1;
    int blkp[AUG ? (EPT + BLK_PER - 1) / BLK_PER : 1];
    __device__ __forceinline__ int blk(int k) const
    {
        if constexpr (BLK_PER == 1) return blkp[k];
        else return (blkp[k >> 2] << (24 - 8 * (k & 3))) >> 24;
    }
    typename pick_type<AUG == FL_AUG_USER, UserConstraints<NW, EPT>, NoConstraints>::type con; // the caller's constraints
    int cpar;               // which of the two c(x) buffers holds the last evaluation's constraints
    int spar;               // ... and which half of the speculative trials' buffers the last batch wrote
    int cshift;             // log2 of the lanes per constraint block where blocks are aligned lane groups (4, 5, 6), else 0

    enum { PH_INIT = 0, PH_LS = 1, PH_DONE = 2, PH_HESS = 3 };
    // internal continuations of advance() (they never leave it): every continuation is inlined ONCE and the
    // branches only choose which one runs.  The first version called after_init_rest() / direction_and_begin() from
    // two places each, which put four copies of the Cholesky kernels and two of every H pass into one kernel --
    // 400-500 spilled VGPRs at 8 elements per thread.
    enum { GO_INIT_REST = -1, GO_DIRECTION = -2, GO_REFRESH = -3, GO_INIT_TAIL = -4, GO_DIRECTION_TAIL = -5 };

    __device__ __forceinline__ Solver(const SolveArgs &A_, double *lds_)
        : A(A_), lds(lds_), prob((STAGED && A_.list) ? A_.list[blockIdx.x] : (int)blockIdx.x), n(A_.n), R{lds_ + L_RED, 0}
    {
    }

    __device__ __forceinline__ double *hist_base() const
    {
        if constexpr (METHOD == FL_SOLVER_LBFGS) return A.hist + (size_t)prob * (size_t)(2 * A.mem) * NPAD;
        // BFGS: H [n][NPAD]; with ExactStep > 0 also U (exact Hessian / its Cholesky factor) and W (inverse factor)
        // (fused kernels, n > 1024: plus 2*BF_DEFER rows for the pending updates' s_l, q_l)
        if constexpr (METHOD == FL_SOLVER_BFGS)
            return A.hist + (size_t)prob * ((size_t)(A.exact_step > 0 ? 3 : 1) * (size_t)n + 2 * BF_DEFER) * NPAD;
        if constexpr (METHOD == FL_SOLVER_NEWTON) return A.hist + (size_t)prob * (size_t)n * NPAD;
        return nullptr;
    }
    __device__ __forceinline__ double *deferred_rows() const // [2*BF_DEFER][NPAD]: s_0, q_0, s_1, q_1, ...
    {
        return hist_base() + (size_t)(A.exact_step > 0 ? 3 : 1) * (size_t)n * NPAD;
    }
    // LDS rows of the slot-th pair of the LDS ring (s row, then y row)
    __device__ __forceinline__ double *lds_pair(int slot) const { return lds + L_G0 + (size_t)(2 * slot) * NPAD; }
    // c(x) of the last evaluation.  Two buffers: where the constraints are reduced per lane group (evaluate()), a trial
    // writes the buffer the previous trial did not use, so the only barrier of a trial is the objective reduction's --
    // a wave still reading trial k's c(x) cannot be overtaken by trial k + 1's writers, and trial k + 2's have passed
    // trial k + 1's barrier.
    __device__ __forceinline__ double *cx_ptr() const { return lds + L_CX + cpar * FL_MAX_CONSTRAINTS; }
    // where g_old waits during the line search (BFGS: in the second broadcast array -- the first one is the row
    // buffer of the Cholesky kernels, which may run before g_old is wanted)
    __device__ __forceinline__ double *g0_park() const
    {
        if constexpr (RCI_LAZY) return rci_vec + 2 * NPAD; // (its HBM row itself: touched only where a search begins / ends)
        if constexpr (METHOD == FL_SOLVER_BFGS) return lds + L_BF + NPAD;
        if constexpr (LDS_PAIRS > 0) {
            const int next = (lrec + 1 == LDS_PAIRS) ? 0 : lrec + 1;
            return lds_pair(next) + NPAD;
        } else {
            return lds + L_G0;
        }
    }

    // ---------------------------------------------------------------- register diet of the dense phases
    // The dense phases (H passes, folds, Cholesky kernels) sweep whole matrices with blocks of columns in registers.
    // At 8 elements per thread the machine's own vectors would not fit beside them (the first version spilled
    // 200-500 VGPRs to scratch there), so for their duration the iterate waits in the caller's row x[prob][:] -- it
    // IS the current iterate, init() read it from there and finish() writes it there -- and the objective's data
    // (d, b of the diagonal quadratic) are re-read afterwards.  Every thread reads back only what it wrote.
    static constexpr bool PARK = (EPT >= 8) && (METHOD == FL_SOLVER_BFGS || METHOD == FL_SOLVER_NEWTON);
    // ... and the line search's scalars are pinned to scalar registers (the L-BFGS / CG kernels measured faster
    // without: fl_linesearch.hpp)
    // LEAN: what else is shed where 256 VGPRs per wave are short (dense kernels at 8 elements per thread; every
    // 512-thread augmented-Lagrangian kernel): d, b of the diagonal quadratic are re-read per evaluation.
    static constexpr bool LEAN = PARK || (AUG && NW >= 8);
    // fused kernels without constraints: g.g is reduced once per line search instead of once per trial
#ifndef FL_LAZY_GG
#define FL_LAZY_GG 1
#endif
    // Wave priority (s_setprio).  Two waves share a SIMD: one is typically in the bulk arithmetic of a trial (independent
    // multiplies / adds), the other somewhere in a serial chain -- a reduction, the scalar line-search step, the two-loop
    // recursion with its row loads.  PRIO = 2 raises a wave's priority from a trial's reduction until the machine hands
    // back the next request (line-search step, convergence tests, new direction) and drops it for the next trial's bulk
    // part, so the chain -- and the recursion's loads -- never wait for the other wave's arithmetic.  Measured on the
    // headline (profiles/r02/ab_prio.txt): 170.2 -> 164.3 ms; raised only over the line-search step: 165.4; only over the
    // direction: 165.5; toggled around every reduction: 169.2.
#ifndef FL_SETPRIO_LBFGS
#define FL_SETPRIO_LBFGS 2
#endif
#ifndef FL_SETPRIO_OTHER
#define FL_SETPRIO_OTHER 2 // (C2 2.85 -> 2.73 ms, C3 70.2 -> 65.8 ms, C5 348 -> 335 ms)
#endif
    static constexpr int PRIO = (OBJ == FL_OBJ_EXTERNAL) ? 0 : ((METHOD == FL_SOLVER_LBFGS && !AUG) ? FL_SETPRIO_LBFGS : FL_SETPRIO_OTHER);
    static constexpr bool LAZY_GG = FL_LAZY_GG && !AUG && OBJ != FL_OBJ_EXTERNAL;
#ifndef FL_UNI_LEAN
#define FL_UNI_LEAN 2
#endif
#ifndef FL_UNI_AUG18
#define FL_UNI_AUG18 2
#endif
    static constexpr int UNI_LEVEL = LEAN ? FL_UNI_LEAN : (AUG_LEAN18 ? FL_UNI_AUG18 : FL_UNI_LEVEL);
    __device__ __forceinline__ void park()
    {
        if constexpr (PARK && !HESS_RCI) store_user<NW, EPT>(A.x + (size_t)prob * n, n, x); // (RCI: x is already there)
    }
    __device__ __forceinline__ void unpark()
    {
        if constexpr (PARK) {
            load_user<NW, EPT>(A.x + (size_t)prob * n, n, x);
            obj.init(A, prob, lds + L_XS);
        }
    }

    // ---------------------------------------------------------------- setup
    __device__ __forceinline__ void init(const double *xrow = nullptr) // xrow: where the current point lies (default x[prob][:])
    {
        obj.init(A, prob, lds + L_XS);
        load_user<NW, EPT>(xrow ? xrow : A.x + (size_t)prob * n, n, x);
        rci_vec = nullptr;
        ls_begun = false;
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = 0.0;
        iters = nf = ng = 0;
        status = FL_STATUS_CONVERGED;
        recent = -1;
        lrec = -1;
        cnt = 0;
        main_it = 0;
        h_valid = 0;
        ndef = 0;
        h_ident = 0;
        a_id = 0.0;
        hess_stage = 0;
        yy_recent = rho_recent = 0.0;
        fnew = gg = pp = phid = phidold = a = 0.0;
        outer_it = 0;
        inner_iters_total = 0;
        cc = 0.0;
        miu = 0.0;
        cshift = 0;
        cpar = 0;
        if constexpr (SPEC_K > 1) spar = 0;
        if constexpr (AUG) {
            if constexpr (AUG == FL_AUG_USER) con.init(A, prob);
            if constexpr (OBJ != FL_OBJ_EXTERNAL && AUG != FL_AUG_USER) { // the built-in constraint family: block spheres
                const int w = n / A.aug_m;
                if (w > 0 && (2 * G::T) % w == 0) cshift = (w == 128) ? 6 : (w == 64 ? 5 : (w == 32 ? 4 : 0));
                if constexpr (BLK_PER > 1) {
#pragma unroll
                    for (int q = 0; q < (EPT + BLK_PER - 1) / BLK_PER; ++q) blkp[q] = 0;
                }
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int e = G::e0(k >> 1) + (k & 1);
                    const int bk = (e < n) ? e / w : -1;
                    if constexpr (BLK_PER == 1) blkp[k] = bk;
                    else blkp[k >> 2] |= (bk & 0xff) << (8 * (k & 3)); // (aug_m <= FL_MAX_CONSTRAINTS = 16)
                }
            }
            miu = A.miu0 > 1.0 ? A.miu0 : 1.0; // miu=max(1d0,miu0)
            if (STAGED && A.resume) { // continued from a paused launch: the outer loop's own state (x and lambda are in the caller's rows)
                const double *ps = A.pstate + (size_t)prob * FL_PSTATE;
                miu = uni(ps[0]);
                cc = uni(ps[1]);
                const int *pi = reinterpret_cast<const int *>(ps + 2);
                outer_it = __builtin_amdgcn_readfirstlane(pi[0]);
                inner_iters_total = __builtin_amdgcn_readfirstlane(pi[1]);
                nf = __builtin_amdgcn_readfirstlane(pi[2]);
                ng = __builtin_amdgcn_readfirstlane(pi[3]);
            }
            const int tl = G::tid();
            if (tl < A.aug_m) lds[L_LAM + tl] = A.lambda[(size_t)prob * A.aug_m + tl];
            __syncthreads();
        }
    }
    __device__ __forceinline__ int start()
    {
        ls.zn = 0; // (must_stop() reads it before the first search begins)
        phase = PH_INIT;
        pending = FL_REQ_F | FL_REQ_G | FL_REQ_NOMOVE;
        return pending;
    }

    // ---------------------------------------------------------------- evaluation (built-in objectives)
    __device__ __forceinline__ void move(double at)
    {
        if constexpr (X0_LDS) {
            double t[EPT];
            load_pad<NW, EPT>(lds + L_X0, t);
#pragma unroll
            for (int k = 0; k < EPT; ++k) x[k] = t[k] + at * p[k];
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k) x[k] = x0[k] + at * p[k];
        }
    }
    // f, g.p, g.g at x; with AUG the objective is the augmented Lagrangian
    // WANT_G = false (augmented Lagrangian only): an objective-only trial -- the reference's shrink loop
    // calls L, not Ld, ~45 times per gradient -- skips the gradient, the constraint
    // Jacobian term and the g.p / g.g reduction; g, gp, ggo are then left untouched.
    template <bool WANT_G = true> __device__ __forceinline__ void evaluate(double &f, double &gp, double &ggo)
    {
        double r[4];
        if constexpr (LEAN) obj.init(A, prob, lds + L_XS); // d, b are re-read per evaluation instead of living in registers
        if constexpr (AUG && !WANT_G) {
            double gl[EPT]; // dead: the compiler drops the gradient arithmetic
            obj.eval(x, gl, r[0], r[1], n, lds + L_XS);
        } else {
            obj.eval(x, g, r[0], r[1], n, lds + L_XS);
        }
        if constexpr (AUG == FL_AUG_USER) {
            // the caller's constraints: the thread's shares of c_0 .. c_(m-1) and the objective's two sums in ONE reduction
            const int m = A.aug_m;
            double q[2 + FL_USER_MAX_CONSTRAINTS], cp[FL_USER_MAX_CONSTRAINTS];
#pragma unroll
            for (int u = 0; u < FL_USER_MAX_CONSTRAINTS; ++u) cp[u] = 0.0;
            con.partial(x, cp, n);
            q[0] = r[0];
            q[1] = r[1];
#pragma unroll
            for (int u = 0; u < FL_USER_MAX_CONSTRAINTS; ++u) q[2 + u] = (u < m) ? cp[u] : 0.0;
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
            __syncthreads(); // readers of the previous c(x) are done
            double *cxs = cx_ptr();
            R.run(q);
            if (G::ltid() == 0) {
#pragma unroll
                for (int u = 0; u < FL_USER_MAX_CONSTRAINTS; ++u) cxs[u] = (u < m) ? q[2 + u] + con.offset(u) : 0.0;
            }
            __syncthreads();
            double lc = 0.0, c2 = 0.0; // L = f - lambda.c + miu/2 c.c (NO.f90:2198)
            for (int j = 0; j < m; ++j) {
                lc = lc + lds[L_LAM + j] * cxs[j];
                c2 = c2 + cxs[j] * cxs[j];
            }
            f = uni(Obj::combine(q[0], q[1]) - lc + miu / 2.0 * c2);
            if constexpr (!WANT_G) return;
            double v[FL_USER_MAX_CONSTRAINTS]; // Ldx=Ldx+matmul(cdx,miu*cx-lambda) (NO.f90:2205)
#pragma unroll
            for (int j = 0; j < FL_USER_MAX_CONSTRAINTS; ++j) v[j] = (j < m) ? miu * cxs[j] - lds[L_LAM + j] : 0.0;
            con.add_gradient(x, v, g, n);
            double q2[2] = {dot_part<EPT>(g, p), dot_part<EPT>(g, g)};
            R.run(q2);
            gp = uni(q2[0]);
            ggo = uni(q2[1]);
        } else if constexpr (AUG) {
            // c_j: masked full-width sums of x^2 (one reduction of aug_m values with the objective's)
            const int m = A.aug_m;
            // the thread's share of c_j + 1: its elements of block j squared, summed in element order (only the
            // constraints of the reduction at hand are formed: 16 of them at once cost 32 VGPRs)
            auto cpart = [&](int j) {
                double acc = 0.0;
                if (j < m) {
#pragma unroll
                    for (int k = 0; k < EPT; ++k) {
                        const double t = (blk(k) == j) ? x[k] * x[k] : 0.0;
                        acc = (k == 0) ? t : acc + t;
                    }
                }
                return acc;
            };
            double r2[2] = {r[0], r[1]};
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
            if (cshift) cpar ^= 1; // (no barrier: the other buffer's readers are a whole trial behind, see cx_ptr())
            else __syncthreads();  // readers of the previous trial's c(x) are done
            double *cxs = cx_ptr();
            if (cshift) {
                // Blocks that coincide with aligned groups of 16 / 32 / 64 lanes of one chunk (block width 32 / 64 /
                // 128; C5: n = 512, 8 blocks of 64): the masked full-width sum of block j only ever adds exact zeros
                // outside its group, so the same bits come from the group's own levels of the reduction tree -- one
                // short reduction per chunk instead of one full-width reduction per constraint.
                const int tl = G::tid();
#pragma unroll
                for (int c = 0; c < G::NCH; ++c) {
                    double v = x[2 * c] * x[2 * c] + x[2 * c + 1] * x[2 * c + 1];
                    if (cshift == 6) v = wave_allreduce(v);
                    else if (cshift == 5) v = row_allreduce(fold16(v, v));
                    else v = row_allreduce(v);
                    const int j = c * (G::T >> cshift) + (tl >> cshift);
                    if ((tl & ((1 << cshift) - 1)) == 0 && j < m) cxs[j] = v - 1.0;
                }
                R.run(r2); // (its barrier also publishes c(x))
            } else if (NW < 8 && m <= 8) { // the objective's two sums and up to 8 constraints in ONE reduction phase
                                    // (not in the 512-thread kernels: ten values at once cost them spills; every value's
                                    // sum has the same order either way)
                double q[10];
                q[0] = r2[0];
                q[1] = r2[1];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[2 + u] = cpart(u);
                R.run(q);
                r2[0] = q[0];
                r2[1] = q[1];
                if (G::ltid() == 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) cxs[u] = q[2 + u] - 1.0;
                }
            } else {
                R.run(r2);
#pragma unroll
                for (int j0 = 0; j0 < FL_MAX_CONSTRAINTS; j0 += 4) { // the constraints four at a time
                    if (j0 < m) {
                        double q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) q[u] = cpart(j0 + u);
                        R.run(q);
                        if (G::ltid() == 0) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) cxs[j0 + u] = q[u] - 1.0;
                        }
                    }
                }
            }
            if (!cshift) __syncthreads();
            // L = f - lambda.c + miu/2 c.c (NO.f90:2198); v = miu*c - lambda (NO.f90:2205)
            double lc = 0.0, c2 = 0.0;
            for (int j = 0; j < m; ++j) {
                lc = lc + lds[L_LAM + j] * cxs[j];
                c2 = c2 + cxs[j] * cxs[j];
            }
            f = uni(Obj::combine(r2[0], r2[1]) - lc + miu / 2.0 * c2);
            if constexpr (!WANT_G) return;
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                if (blk(k) >= 0) {
                    int j = blk(k);
                    asm volatile("" : "+v"(j)); // (the LDS addresses of c_j, lambda_j are formed here, not kept from init())
                    const double v = miu * cxs[j] - lds[L_LAM + j];
                    g[k] = g[k] + (2.0 * x[k]) * v; // Ldx=Ldx+matmul(cdx,miu*cx-lambda)
                }
            }
            double q2[2] = {dot_part<EPT>(g, p), dot_part<EPT>(g, g)};
            R.run(q2);
            gp = uni(q2[0]);
            ggo = uni(q2[1]);
        } else if constexpr (LAZY_GG) {
            // g.g is wanted once per line search (the convergence test after it), not per trial: advance() reduces it
            // when the search has ended -- 15 of the ~94 f64 operations of a trial
            double r3[3] = {r[0], r[1], dot_part<EPT>(g, p)};
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
            R.run(r3);
            f = uni(Obj::combine(r3[0], r3[1]));
            gp = uni(r3[2]);
        } else {
            r[2] = dot_part<EPT>(g, p);
            r[3] = dot_part<EPT>(g, g);
            if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(3);
            R.run(r);
            f = uni(Obj::combine(r[0], r[1]));
            gp = uni(r[2]);
            ggo = uni(r[3]);
        }
    }
    __device__ __forceinline__ double reduce_gg()
    {
        double q[1] = {dot_part<EPT>(g, g)};
        R.run(q);
        return uni(q[0]);
    }

    // ---------------------------------------------------------------- speculative objective-only trials (SPEC_K)
    // is the pending objective-only request inside one of the loops a <- a / incrmt (StrongWolfe's and Wolfe's Armijo-violated
    // branches)?  (Wolfe's growing loop a <- a * incrmt, NO.f90:1308-1313, stays with the machine: one more copy of the
    // pass for a searcher nobody picks by default.)
    __device__ __forceinline__ bool spec_shrinking() const
    {
        if (cshift == 0) return false; // (only where the constraints are reduced per lane group: evaluate())
        return ls.st == LineSearch::SW_V_F || ls.st == LineSearch::W_SHRINK;
    }
    // fs[k] = L(x0 + as[k] p), k < K, for given step lengths.  x, g are not touched.  CS = cshift (a template parameter so
    // that the whole pass is straight-line code: the reductions of different chunks and trials interleave).
    template <int K, int CS> __device__ __forceinline__ void evaluate_spec(const double (&as)[K], double (&fs)[K], const double *x0row)
    {
        static_assert(AUG, "objective-only trials exist in the augmented-Lagrangian kernels only");
        const int m = A.aug_m;
        double r[2 * K], cv[K][G::NCH];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double xk[EPT], gl[EPT]; // gl is dead: the compiler drops the gradient arithmetic
            double xb[EPT];
            if constexpr (X0_LDS) { // x0 is read from its LDS row again for every trial (FL_SPEC_X0_RELOAD): 16 VGPRs
                                    // less across the pass, which is what lets four trials fit three waves per SIMD
                load_pad<NW, EPT>(x0row, xb);
                if constexpr (FL_SPEC_X0_RELOAD) asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) xk[e] = (X0_LDS ? xb[e] : x0[X0_LDS ? 0 : e]) + as[k] * p[e];
            obj.eval(xk, gl, r[2 * k], r[2 * k + 1], n, lds + L_XS);
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) cv[k][c] = xk[2 * c] * xk[2 * c] + xk[2 * c + 1] * xk[2 * c + 1];
        }
        spar ^= 1; // (the other half's readers are a whole batch behind: see cx_ptr())
        double *cxs = lds + L_CXS + spar * (K * FL_MAX_CONSTRAINTS);
        const int tl = G::tid();
        const int lane = tl & 63;
        double *junk = lds + L_CXS + 2 * K * FL_MAX_CONSTRAINTS + tl;
        // the block sums: the levels of the fixed tree below the block's lane group, as in evaluate(); values of several
        // trials share the registers of one pass where the tree leaves room (fl_reduce.hpp)
#pragma unroll
        for (int c = 0; c < G::NCH; ++c) {
            const int j = c * (G::T >> CS) + (tl >> CS);
            if constexpr (CS == 6) { // block = the wave: value k ends in row group_row(k)
                double v4[K];
#pragma unroll
                for (int k = 0; k < K; ++k) v4[k] = cv[k][c];
                const double q = wave_reduce_group<K>(v4);
                const int row = lane >> 4;
                const int k = (K == 4) ? (((row & 1) << 1) | (row >> 1)) : (row >> 1);
                const bool own = (lane & 15) == 0 && (K == 4 || (row & 1) == 0) && j < m;
                *(own ? cxs + k * FL_MAX_CONSTRAINTS + j : junk) = q - 1.0;
            } else if constexpr (CS == 5) { // block = two rows: two trials per pass (rows 0, 2: the first, rows 1, 3: the second)
#pragma unroll
                for (int k = 0; k < K; k += 2) {
                    const double q = row_allreduce(fold16(cv[k][c], cv[k + 1][c]));
                    const int kk = k + ((lane >> 4) & 1);
                    *(((lane & 15) == 0 && j < m) ? cxs + kk * FL_MAX_CONSTRAINTS + j : junk) = q - 1.0;
                }
            } else { // block = one row
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double q = row_allreduce(cv[k][c]);
                    *(((lane & 15) == 0 && j < m) ? cxs + k * FL_MAX_CONSTRAINTS + j : junk) = q - 1.0;
                }
            }
        }
        R.run(r); // (its barrier also publishes c(x); one wave: LDS operations of a wave complete in order)
        // lambda.c and c.c of all K trials at once: lane j of DPP row k takes c_j of trial k and lambda_j, and the sums run
        // down the row in the reference's order, S_j = S_(j-1) + P_j from S = +0 (dot_product, NO.f90:2198) -- m - 1
        // shifted additions for all trials together instead of 2 m multiply-adds per trial in every lane
        const int li = lane & (K * FL_MAX_CONSTRAINTS - 1);
        const double cj = cxs[li], lam = lds[L_LAM + (li & (FL_MAX_CONSTRAINTS - 1))];
        const double P = lam * cj, Q = cj * cj;
        double sp = 0.0 + P, sq = 0.0 + Q;
        for (int i = 1; i < m; ++i) {
            sp = dpp_shr1_zero(sp) + P;
            sq = dpp_shr1_zero(sq) + Q;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double lc = read_lane(sp, k * FL_MAX_CONSTRAINTS + m - 1), c2 = read_lane(sq, k * FL_MAX_CONSTRAINTS + m - 1);
            fs[k] = Obj::combine(r[2 * k], r[2 * k + 1]) - lc + miu / 2.0 * c2;
        }
    }
    // The pending request is the objective at ls.a_eval inside one of the loops above: run the loop here up to the trial
    // at which the reference leaves it, and return that trial's objective value with the machine's variables as its own
    // steps would have left them (non-exit step: fx=fv; aold=a; fold=fx; a=aold/incr -- sw_v_next, w_shrink_next),
    // so that ls.step() of the returned value takes the exit branch.  Exit test of SW_V_F / W_SHRINK: Armijo holds or
    // a < 1e-15 (NO.f90:1518-1521, 1543; 1325-1329, 1337).
    template <int K, int CS> __device__ __forceinline__ double fast_forward_cs()
    {
        double aold = ls.aold, fold = ls.fold, a_x, f_x;
        int consumed = 0;
        auto next = [&](double a_) { return uni(a_ / ls.incr); }; // a=aold/incrmt: the expression the machine itself forms
        double as[K];
        as[0] = ls.a_eval;
#pragma unroll
        for (int k = 1; k < K; ++k) as[k] = next(as[k - 1]);
        for (;;) {
            double an[K], fs[K]; // the next pass's step lengths do not depend on this pass's values: their divisions overlap it
            an[0] = next(as[K - 1]);
#pragma unroll
            for (int k = 1; k < K; ++k) an[k] = next(an[k - 1]);
            evaluate_spec<K, CS>(as, fs, lds + L_X0);
            unsigned exits = 0; // bit k: the reference leaves the loop at trial k (Armijo holds, or a < 1e-15)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double bound = ls.fx0 + ls.c1 * as[k] * ls.phid0;
                exits |= (unsigned)((fs[k] <= bound) | (as[k] < 1e-15)) << k;
            }
            exits = __builtin_amdgcn_readfirstlane(exits);
            if (exits) {
                const int kx = __builtin_ctz(exits);
                a_x = as[0];
                f_x = fs[0];
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const bool take = kx >= k;
                    aold = take ? a_x : aold;
                    fold = take ? f_x : fold;
                    a_x = take ? as[k] : a_x;
                    f_x = take ? fs[k] : f_x;
                }
                consumed += kx;
                break;
            }
            aold = as[K - 1];
            fold = fs[K - 1];
            consumed += K;
#pragma unroll
            for (int k = 0; k < K; ++k) as[k] = an[k];
        }
        ls.a = ls.a_eval = uni(a_x);
        ls.aold = uni(aold);
        ls.fold = uni(fold);
        nf += consumed; // (the exit trial itself is counted by advance())
        // x <- the exit trial's point (the machine asks for the gradient there next).  Neither x nor g is live across the
        // loop above: x is formed here, and every exit of these loops is followed by a gradient request (SW_V_G /
        // SW_LAST_G, W_SHRINK_G / W_LAST_G) that rewrites g before anything reads it -- said to the register
        // allocator by leaving g undefined here, which frees 32 VGPRs inside the loop.
        move(ls.a_eval);
#pragma unroll
        for (int k = 0; k < EPT; ++k) asm volatile("" : "=v"(g[k]));
        return uni(f_x);
    }
    // The same loop TRIAL-PARALLEL over the REP waves of a workgroup (fl_solve_rep_kernel; round 4).  A batch that
    // under-fills the chip -- one GPU's share of BASELINE config 5 on eight GPUs: 1024 problems for 1024 SIMDs -- is bound by
    // its slowest problem, and that problem's time is this loop (profiles/r04/phase_timers_c5.txt: 58 of 72 ms; the reference
    // walks a <- a / 1.05 hundreds of steps after every restart of the inner solver).  More waves per problem in the usual way
    // (elements over waves) shorten a trial only by its element-wise part and change the summation order.  Instead wave 0 (the
    // MASTER) runs the machine as fl_solve_kernel does, with every element of the problem in its registers, and REP - 1 HELPER
    // waves hold the objective's data and wait at the workgroup barrier; in this loop wave r evaluates trials r K .. r K + K - 1
    // of each pass of REP K consecutive step lengths.  Every trial is summed inside ONE wave exactly as the unhelped kernel
    // sums it, so the result -- minimiser, counts, every bit -- is that of the throughput geometry: the helpers are invisible
    // to the caller and to the oracle.  Per pass the waves exchange (a, f) of their trials and their exit bits through LDS (two
    // alternating buffers, one barrier); the chain of divisions a_(t+1) = a_t / incrmt is walked by every wave in full (its
    // roundings are part of the reference's arithmetic).  A helper skips the epilogue (the machine is the master's).
#ifndef FL_WIDE_CHAIN_FIRST
#define FL_WIDE_CHAIN_FIRST 1
#endif
    template <int K, int CS, int REP> __device__ __forceinline__ double fast_forward_wide_cs(double *xch, int rep, const double *x0row, bool master)
    {
        constexpr int W = K * REP;
        int *xmask = reinterpret_cast<int *>(xch + 4 * W); // [2][REP]
        auto next = [&](double a_) { return uni(a_ / ls.incr); };
        double as[K];
        {
            double c = ls.a_eval;
            for (int i = 0; i < rep * K; ++i) c = next(c);
            as[0] = c;
#pragma unroll
            for (int k = 1; k < K; ++k) as[k] = next(as[k - 1]);
        }
        double aold = ls.aold, fold = ls.fold, a_x = 0.0, f_x = 0.0;
        int consumed = 0, par = 0;
        for (;;) {
            double an[K], fs[K]; // the wave's step lengths of the next pass: W - K + 1 divisions on from its last one
#if FL_WIDE_CHAIN_FIRST
            {
                double c = as[K - 1];
#pragma unroll
                for (int i = 0; i < W - K + 1; ++i) c = next(c);
                an[0] = c;
#pragma unroll
                for (int k = 1; k < K; ++k) an[k] = next(an[k - 1]);
            }
#endif
            evaluate_spec<K, CS>(as, fs, x0row);
#if !FL_WIDE_CHAIN_FIRST
            {
                double c = as[K - 1];
#pragma unroll
                for (int i = 0; i < W - K + 1; ++i) c = next(c);
                an[0] = c;
#pragma unroll
                for (int k = 1; k < K; ++k) an[k] = next(an[k - 1]);
            }
#endif
            unsigned mine = 0; // bit k: the reference leaves the loop at this wave's trial k (Armijo holds, or a < 1e-15)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double bound = ls.fx0 + ls.c1 * as[k] * ls.phid0;
                mine |= (unsigned)((fs[k] <= bound) | (as[k] < 1e-15)) << k;
            }
            double *buf = xch + par * (2 * W);
            if (G::ltid() == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    buf[2 * (rep * K + k)] = as[k];
                    buf[2 * (rep * K + k) + 1] = fs[k];
                }
                xmask[par * REP + rep] = (int)mine;
            }
            __syncthreads();
            unsigned exits = 0;
#pragma unroll
            for (int r = 0; r < REP; ++r) exits |= (unsigned)xmask[par * REP + r] << (r * K);
            exits = __builtin_amdgcn_readfirstlane(exits);
            if (exits) {
                if (master) {
                    const int kx = __builtin_ctz(exits);
                    a_x = buf[2 * kx];
                    f_x = buf[2 * kx + 1];
                    if (kx > 0) {
                        aold = buf[2 * kx - 2];
                        fold = buf[2 * kx - 1];
                    } else if (consumed > 0) { // the last trial of the previous pass (the other buffer: not rewritten yet)
                        const double *prev = xch + (par ^ 1) * (2 * W);
                        aold = prev[2 * W - 2];
                        fold = prev[2 * W - 1];
                    }
                    consumed += kx;
                }
                break;
            }
            consumed += W;
            par ^= 1;
#pragma unroll
            for (int k = 0; k < K; ++k) as[k] = an[k];
        }
        double fr = 0.0;
        if (master) {
            ls.a = ls.a_eval = uni(a_x);
            ls.aold = uni(aold);
            ls.fold = uni(fold);
            nf += consumed;
            move(ls.a_eval);
            fr = uni(f_x);
        } else { // (a helper's x is never read: undefined here, so that it does not occupy registers across the loop)
#pragma unroll
            for (int k = 0; k < EPT; ++k) asm volatile("" : "=v"(x[k]));
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) asm volatile("" : "=v"(g[k])); // (as in fast_forward_cs: rewritten before it is read)
        __syncthreads(); // every wave has read the exchange buffers before the next search's first pass rewrites them
        return fr;
    }
    template <int K, int REP> __device__ __forceinline__ double fast_forward_wide(double *xch, int rep, const double *x0row, bool master)
    {
        if (cshift == 5) return fast_forward_wide_cs<K, 5, REP>(xch, rep, x0row, master);
        if (cshift == 6) return fast_forward_wide_cs<K, 6, REP>(xch, rep, x0row, master);
        return fast_forward_wide_cs<K, 4, REP>(xch, rep, x0row, master);
    }
    // what a helper needs of the master's state for one shrink loop: the search direction (a row), the loop's scalars, the
    // penalty parameter and the multipliers.  pub: [NPAD] p | PUB_SCALARS scalars | [FL_MAX_CONSTRAINTS] lambda
    static constexpr int PUB_SCALARS = 8;
    static constexpr int PUB_X0 = X0_LDS ? 0 : NPAD; // x0 too where the master keeps it in registers (else the helpers read its LDS row)
    static constexpr int PUB_DOUBLES = NPAD + PUB_X0 + PUB_SCALARS + FL_MAX_CONSTRAINTS;
    __device__ __forceinline__ void publish_search(double *pub)
    {
        store_pad<NW, EPT>(pub, p);
        if constexpr (!X0_LDS) store_pad<NW, EPT>(pub + NPAD, x0);
        const int tl = G::tid();
        if (tl == 0) {
            double *q = pub + NPAD + PUB_X0;
            q[0] = ls.a_eval; q[1] = ls.incr; q[2] = ls.fx0; q[3] = ls.c1; q[4] = ls.phid0; q[5] = miu;
        }
        if (tl < A.aug_m) pub[NPAD + PUB_X0 + PUB_SCALARS + tl] = lds[L_LAM + tl];
    }
    __device__ __forceinline__ void helper_take(const double *pub)
    {
        load_pad<NW, EPT>(pub, p);
        if constexpr (!X0_LDS) load_pad<NW, EPT>(pub + NPAD, x0);
        const double *q = pub + NPAD + PUB_X0;
        ls.a_eval = uni(q[0]); ls.incr = uni(q[1]); ls.fx0 = uni(q[2]); ls.c1 = uni(q[3]); ls.phid0 = uni(q[4]);
        miu = uni(q[5]);
        ls.aold = ls.fold = 0.0;
        const int tl = G::tid();
        if (tl < A.aug_m) lds[L_LAM + tl] = pub[NPAD + PUB_X0 + PUB_SCALARS + tl]; // (read back by this wave only: in order)
    }
    template <int K> __device__ __forceinline__ double fast_forward()
    {
        if (cshift == 5) return fast_forward_cs<K, 5>();
        if (cshift == 6) return fast_forward_cs<K, 6>();
        return fast_forward_cs<K, 4>();
    }

    // ---------------------------------------------------------------- the growing loop of StrongWolfe (GROW_K)
    // "search for larger a" (NO.f90:1499-1515; _fdwithf twin 1620-1634): a <- a * incrmt with f AND f' at every trial until
    // Armijo fails, f stops falling or the slope turns positive.  The reference walks it in 5 % steps, ~12 of the ~14 trials
    // of an L-BFGS iteration on the benched quadratics; only the exit depends on the values.  Same idea as fast_forward,
    // for kernels without constraints: GROW_K consecutive trials per pass share one reduction and skip the pass through
    // the whole machine; the machine gets the exit trial with the state its own steps would have left (non-exit step:
    // aold=a; fold=fx; phidold=phidnew; a=aold*incr), x and g are formed again at the exit trial's point.
    // Measured (gpurun_out/r03_ab_grow_*.txt -> profiles/r03/grow_ab.txt): headline L-BFGS 65 536 x 1024: 163.9 -> 154.9 ms
    // with two trials per pass (three: 155.3); C2 (Rosenbrock n = 256: a barrier pair per evaluation either way) 2.66 ->
    // 2.66 ms; the SD / CG kernels have no registers to spare for it (capped at 128 VGPRs: 3-10 would spill; 1 x 16 is at 252).
#ifndef FL_GROW_K
#define FL_GROW_K 2
#endif
    static constexpr int GROW_K = (!AUG && OBJ != FL_OBJ_EXTERNAL && METHOD == FL_SOLVER_LBFGS && EPT <= 8 && FL_GROW_K > 1) ? FL_GROW_K : 1;
    static_assert(GROW_K >= 1 && GROW_K <= 3, "3 sums per trial, at most 10 values per reduction");
    __device__ __forceinline__ bool grow_loop_pending() const { return ls.st == LineSearch::SW_GROW; }
    template <int K> __device__ __forceinline__ void fast_forward_grow(double &fv_out, double &pv_out)
    {
        double aold = ls.aold, fold = ls.fold, pold = ls.phidold;
        double a_x = ls.a_eval, f_x = 0.0, p_x = 0.0;
        int consumed = 0;
        bool found = false;
        auto next = [&](double a_) { return uni(a_ * ls.incr); }; // a=aold*incrmt: the expression the machine itself forms
        double as[K];
        as[0] = ls.a_eval;
#pragma unroll
        for (int k = 1; k < K; ++k) as[k] = next(as[k - 1]);
        for (;;) {
            double r[3 * K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double xk[EPT], gk[EPT];
                if constexpr (X0_LDS) {
                    load_pad<NW, EPT>(lds + L_X0, xk);
#pragma unroll
                    for (int e = 0; e < EPT; ++e) xk[e] = xk[e] + as[k] * p[e];
                } else {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) xk[e] = x0[X0_LDS ? 0 : e] + as[k] * p[e];
                }
                if constexpr (LEAN) obj.init(A, prob, lds + L_XS);
                obj.eval(xk, gk, r[3 * k], r[3 * k + 1], n, lds + L_XS);
                r[3 * k + 2] = dot_part<EPT>(gk, p);
            }
            R.run(r);
            double fs[K], ps[K];
            unsigned exits = 0;
            double fprev = fold;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                fs[k] = uni(Obj::combine(r[3 * k], r[3 * k + 1]));
                ps[k] = uni(r[3 * k + 2]);
                const double bound = ls.fx0 + ls.c1 * as[k] * ls.phid0;
                exits |= (unsigned)((fs[k] > bound) | (fs[k] >= fprev) | (ps[k] > 0.0)) << k;
                fprev = fs[k];
            }
            exits = __builtin_amdgcn_readfirstlane(exits);
            if (exits) {
                const int kx = __builtin_ctz(exits);
                a_x = as[0];
                f_x = fs[0];
                p_x = ps[0];
#pragma unroll
                for (int k = 1; k < K; ++k) {
                    const bool take = kx >= k;
                    aold = take ? a_x : aold;
                    fold = take ? f_x : fold;
                    pold = take ? p_x : pold;
                    a_x = take ? as[k] : a_x;
                    f_x = take ? fs[k] : f_x;
                    p_x = take ? ps[k] : p_x;
                }
                consumed += kx;
                found = true;
                break;
            }
            aold = as[K - 1];
            fold = fs[K - 1];
            pold = ps[K - 1];
            consumed += K;
            as[0] = next(aold);
#pragma unroll
            for (int k = 1; k < K; ++k) as[k] = next(as[k - 1]);
            if (!(as[0] < 1e300)) { // (a step length running away to infinity: hand the loop back to the machine as it stands)
                a_x = as[0];
                break;
            }
        }
        ls.a = ls.a_eval = uni(a_x);
        ls.aold = uni(aold);
        ls.fold = uni(fold);
        ls.phidold = uni(pold);
        nf += consumed; // (every consumed trial was an f + f' request; the exit trial itself is counted by advance())
        ng += consumed;
        // x, g at the exit trial's point (the search may accept it as it is)
        move(ls.a_eval);
        double s0, s1;
        if constexpr (LEAN) obj.init(A, prob, lds + L_XS);
        obj.eval(x, g, s0, s1, n, lds + L_XS);
        if (found) {
            fv_out = f_x;
            pv_out = p_x;
        } else { // (ran away: evaluate the pending trial for real)
            double r3[3] = {s0, s1, dot_part<EPT>(g, p)};
            R.run(r3);
            fv_out = uni(Obj::combine(r3[0], r3[1]));
            pv_out = uni(r3[2]);
        }
    }
    // Reverse communication with the CALLER's constraints (AugmentedLagrangian's c, cd callbacks, NO.f90:1928-1934):
    // L = f - lambda.c + miu/2 c.c (L, NO.f90:2198) and, with the caller's gradient already in g,
    // grad L = fd + matmul(cdx, miu*cx - lambda) (Ld, NO.f90:2205; cd_user: [m][n], row j = grad c_j), then g.p, g.g.
    // The sum over the constraints is taken per element in the order j = 0, 1, ... like the oracle's restatement.
    __device__ __forceinline__ void take_external_aug(double fuser, bool have_f, bool have_g, const double *c_user,
                                                      const double *cd_user, double &f, double &gp, double &ggo)
    {
        const int m = A.aug_m;
        double *cxs = cx_ptr();
        __syncthreads(); // readers of the previous c(x) are done
        const int tl = G::tid();
        if (tl < m) cxs[tl] = c_user[tl];
        __syncthreads();
        if (have_f) {
            double lc = 0.0, c2 = 0.0;
            for (int j = 0; j < m; ++j) {
                lc = lc + lds[L_LAM + j] * cxs[j];
                c2 = c2 + cxs[j] * cxs[j];
            }
            f = uni(fuser - lc + miu / 2.0 * c2);
        }
        if (have_g) {
            double t[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) t[k] = 0.0;
            for (int j = 0; j < m; ++j) {
                const double v = miu * cxs[j] - lds[L_LAM + j];
                double row[EPT];
                load_user<NW, EPT>(cd_user + (size_t)j * n, n, row);
#pragma unroll
                for (int k = 0; k < EPT; ++k) t[k] = t[k] + row[k] * v;
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) g[k] = g[k] + t[k];
            double q2[2] = {dot_part<EPT>(g, p), dot_part<EPT>(g, g)};
            R.run(q2);
            gp = uni(q2[0]);
            ggo = uni(q2[1]);
        }
    }

    // ---------------------------------------------------------------- machine
    // Called with the result of the pending request; returns the next request (0 = finished).
    __device__ __forceinline__ int advance(double fv, double pv, double gg_new)
    {
        nf += (pending & FL_REQ_F) ? 1 : 0;
        ng += (pending & FL_REQ_G) ? 1 : 0;
        int rq;
        if (phase == PH_INIT) {
            if constexpr (LAZY_GG) gg_new = reduce_gg();
            rq = after_init(fv, gg_new);
        } else if (HESS_RCI && phase == PH_HESS) { // the caller has written the Hessian it was asked for
            rq = (hess_stage == 0) ? GO_INIT_REST : GO_DIRECTION;
        } else {
            if constexpr (!LAZY_GG) gg = gg_new;
            if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(3);
            rq = __builtin_amdgcn_readfirstlane(ls.step(fv, pv));
            ls.template uniformize<UNI_LEVEL>();
            if (rq == 0) {
                if constexpr (PRIO == 6) __builtin_amdgcn_s_setprio(3);
                if constexpr (LAZY_GG) gg = reduce_gg(); // x, g are those of the accepted point (the last evaluation)
                rq = after_linesearch();
            }
        }
        bool initial = false, refreshed = false;
        if (rq == GO_INIT_REST) {
            initial = true;
            rq = after_init_rest();
        }
        if (rq == GO_DIRECTION) rq = direction_head();
        if constexpr (METHOD == FL_SOLVER_BFGS && EXACT) {
            if (rq == GO_REFRESH) { // the one copy of the Cholesky kernels
                refreshed = bfgs_exact_refresh();
                if (initial) {
                    if (refreshed) {
                        h_valid = 1;
                        status = FL_STATUS_MAXIT;
                        rq = begin_linesearch();
                    } else {
                        rq = GO_INIT_TAIL;
                    }
                } else {
                    rq = GO_DIRECTION_TAIL;
                }
            }
        }
        if (rq == GO_INIT_TAIL) rq = after_init_tail();
        if (rq == GO_DIRECTION_TAIL) rq = direction_and_begin(refreshed);
        pending = rq;
        if constexpr (PRIO >= 1) __builtin_amdgcn_s_setprio(0);
        return rq;
    }
    __device__ __forceinline__ double request_point() const { return ls.a_eval; }
    // An objective that is not a number ends the problem (FL_STATUS_NOT_FINITE): called by the kernels' loops instead of
    // advance().  The reference's searchers compare their way out of every loop (zoom: NO.f90:1557-1579), so on NaN they -- and
    // this restatement of them -- never leave: a host program that hangs is the caller's problem, a kernel that never ends
    // takes the device with it.  No effect on any finite run.  (In the loop, not inside advance(): an early return there cost
    // the dense augmented-Lagrangian kernels 10-70 spilled VGPRs.)
    __device__ __forceinline__ bool not_finite(double fv) const { return (pending & FL_REQ_F) && fv != fv; }
    // ... and so does a zoom that never narrows (FL_STATUS_STALLED).  The reference's zoom has no iteration limit (NO.f90:1557-1579,
    // Wolfe's: 1347-1370): found on the quartic + block spheres at x -> 1/8 with both end slopes positive at the rounding level, where
    // the interpolation returns to the same two points for ever -- the oracle, restating the reference, never returns there either
    // (tests/test_gpu_helpers.py: problem 1724 of that family).  The machine counts its zoom trials (LineSearch::zn); beyond
    // FL_ZOOM_CAP the problem ends where it is.  ONE flag for both conditions, so that the kernels' loops carry one test: a
    // second test of their own cost C3 2-5 %.
    __device__ __forceinline__ bool must_stop(double fv) const { return not_finite(fv) || ls.stalled(); }
    __device__ __forceinline__ void stop_not_finite() // (behind must_stop(): which of the two it was is still in ls.zn)
    {
        if (ls.stalled()) {
            status = FL_STATUS_STALLED;
        } else {
            status = FL_STATUS_NOT_FINITE;
            fnew = __builtin_nan(""); // (not the value itself: it would have to stay in registers across advance())
        }
        phase = PH_DONE;
        pending = 0;
    }

    __device__ __forceinline__ int begin_linesearch()
    {
        // the *_fdwithf searchers only in main loops with f_fd present; the first search of L-BFGS / BFGS
        // and L-BFGS' pre-iterations never use f_fd (NO.f90:448-460, 486-498, 689-701)
        int fused = A.fused;
        if constexpr (AUG) fused = 1; // AugmentedLagrangian always passes f_fd=L_Ld (NO.f90:2153, 2161)
        if constexpr (METHOD == FL_SOLVER_LBFGS) fused = fused && (iters >= A.mem);
        if constexpr (METHOD == FL_SOLVER_BFGS || METHOD == FL_SOLVER_NEWTON) {
            // main loop `do iIteration=1,maxit` (NO.f90:717-929, 1077-1214); BFGS' first step precedes it
            const bool in_main = (METHOD == FL_SOLVER_NEWTON) || h_valid;
            if (in_main) {
                if (main_it >= A.maxit) {
                    status = FL_STATUS_MAXIT;
                    return inner_finished();
                }
                ++main_it;
            }
            fused = fused && in_main;
        }
        if constexpr (X0_LDS) {
            store_pad<NW, EPT>(lds + L_X0, x); // xold=x (doubles as the line search's x0)
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k) x0[k] = x[k];
        }
        if constexpr (NEEDS_G0) store_pad<NW, EPT>(g0_park(), g); // fdold=fdnew, parked in LDS
        ls_begun = true;
        phidold = phid;
        const int strong = (METHOD == FL_SOLVER_CG && A.cg_method == FL_CG_PR) ? 1 : A.strong;
        phase = PH_LS;
        const int rq = __builtin_amdgcn_readfirstlane(ls.begin(strong, fused, A.c1, A.c2, A.incr, a, fnew, phid));
        ls.template uniformize<UNI_LEVEL>();
        return rq;
    }

    // initial f(x), f'(x) are in (NO.f90:87-96, 230-239, 436-445, 668-687)
    __device__ __forceinline__ int after_init(double f, double gg0)
    {
        fnew = f;
        gg = gg0;
        if constexpr (HESS_RCI && (METHOD == FL_SOLVER_NEWTON || METHOD == FL_SOLVER_BFGS)) {
            if (METHOD == FL_SOLVER_NEWTON || A.exact_step > 0) { // info=fdd(H,x,dim): ask the caller (NO.f90:675, 1065)
                phase = PH_HESS;
                hess_stage = 0;
                return FL_REQ_H | FL_REQ_SAME;
            }
        }
        return GO_INIT_REST;
    }
    __device__ __forceinline__ int after_init_rest()
    {
        if constexpr (METHOD == FL_SOLVER_NEWTON) { // NO.f90:1064-1076: no gradient test when the Hessian is SPD
            status = FL_STATUS_MAXIT;
            if (newton_direction(true)) return begin_linesearch();
        }
        if constexpr (METHOD == FL_SOLVER_BFGS && EXACT) { // NO.f90:674-682: exact inverse Hessian first, if asked for
            if (A.exact_step > 0) return GO_REFRESH;
        }
        return GO_INIT_TAIL;
    }
    __device__ __forceinline__ int after_init_tail()
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -g[k];
        phid = -gg; // p=-fdnew; phidnew=-dot_product(fdnew,fdnew)
        pp = gg;
        status = FL_STATUS_CONVERGED;
        if (gg < A.tol) return inner_finished(); // if(-phidnew<tol) return
        a = uni((fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(gg));
        status = FL_STATUS_MAXIT;
        if (max_linesearches() <= 0) return inner_finished();
        return begin_linesearch();
    }
    __device__ __forceinline__ int max_linesearches() const
    {
        // L-BFGS: 1 + (mem-1) pre-iterations + maxit; SD/CG: maxit; BFGS / Newton count main-loop
        // iterations in begin_linesearch (main_it)
        if constexpr (METHOD == FL_SOLVER_LBFGS) return A.mem + A.maxit;
        if constexpr (METHOD == FL_SOLVER_BFGS || METHOD == FL_SOLVER_NEWTON) return 0x7fffffff;
        return A.maxit;
    }

    // ---- dense directions (analytic Hessian of the built-in objective = the reference's fdd branch)
    __device__ __forceinline__ void fill_hessian(double *Hm)
    {
        if constexpr (HESS_RCI) return; // reverse communication: the caller has written it (FL_REQ_H)
        if constexpr (AUG) {
            // Hessian of the augmented Lagrangian as the reference's Ldd forms it (NO.f90:2229-2241):
            //   Lddx = (fdd + [matmul(cddx(i,:,:), miu*cx - lambda)]_i) + matmul(cdx, transpose(cdx))
            // -- note: no miu on the last term, as written.  Block spheres: c_j'' = 2 I on block j, grad c_j = 2 x on
            // block j, so column j gets 2 v_b on the diagonal and (2 x_i)(2 x_j) on the rows of its block b.
            // c(x) of the last evaluation (= the current x) is in LDS; x is staged in the first row buffer.
            double *xstage = lds + L_BF;
            const double *cxs = cx_ptr();
            __syncthreads();
            store_pad<NW, EPT>(xstage, x);
            __syncthreads();
            const int w = n / A.aug_m;
            // two sweeps over the columns (every thread re-reads only what it wrote): the objective's part first, then
            // the constraint terms -- together they were 31 VGPRs too many for the 512-thread kernel
            for (int j = 0; j < n; ++j) {
                double h[EPT];
                obj.hess_column(j, x, n, lds + L_XS, h);
                store_pad<NW, EPT>(Hm + (size_t)j * NPAD, h);
            }
            for (int j = 0; j < n; ++j) {
                double h[EPT];
                load_pad<NW, EPT>(Hm + (size_t)j * NPAD, h);
                const int bj = j / w;
                const double vb = miu * cxs[bj] - lds[L_LAM + bj], xj2 = 2.0 * xstage[j];
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const int i = G::e0(k >> 1) + (k & 1);
                    const double t = (i == j) ? 2.0 * vb : 0.0;
                    const double pterm = (blk(k) == bj) ? (2.0 * x[k]) * xj2 : 0.0;
                    h[k] = (h[k] + t) + pterm;
                }
                store_pad<NW, EPT>(Hm + (size_t)j * NPAD, h);
            }
            __syncthreads();
            return;
        }
        for (int j = 0; j < n; ++j) {
            double h[EPT];
            obj.hess_column(j, x, n, lds + L_XS, h);
            store_pad<NW, EPT>(Hm + (size_t)j * NPAD, h);
        }
        __syncthreads();
    }
    // p = -matmul(H, g): thread i sums H(i,j) g_j over j in order; gbuf = LDS [NPAD]
    __device__ __forceinline__ void neg_matvec(const double *Hm, double *gbuf)
    {
        __syncthreads();
        store_pad<NW, EPT>(gbuf, g);
        __syncthreads();
        double acc[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) acc[k] = 0.0;
        for (int j = 0; j < n; j += BF_UNROLL) {
            double h[BF_UNROLL][EPT];
#pragma unroll
            for (int u = 0; u < BF_UNROLL; ++u)
                if (j + u < n) load_pad<NW, EPT>(Hm + (size_t)(j + u) * NPAD, h[u]);
#pragma unroll
            for (int u = 0; u < BF_UNROLL; ++u)
                if (j + u < n) {
                    const double gj = gbuf[j + u];
#pragma unroll
                    for (int k = 0; k < EPT; ++k) acc[k] = acc[k] + h[u][k] * gj;
                }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -acc[k];
    }
    __device__ __forceinline__ void direction_scalars() // phidnew=dot_product(fdnew,p); a=1d0
    {
        double q[2] = {dot_part<EPT>(g, p), dot_part<EPT>(p, p)};
        R.run(q);
        phid = uni(q[0]);
        pp = uni(q[1]);
        a = 1.0;
    }
    // NewtonRaphson: p=-fdnew; info=fdd(Hessian,x,dim); call My_dposv(Hessian,p,dim,info) (NO.f90:1065-1067, 1232)
    __device__ __forceinline__ bool newton_direction(bool initial)
    {
        double *Hm = hist_base();
        fill_hessian(Hm);
        park();
        const int info = DN::cholesky(Hm, n, lds + L_BF, lds + L_SLOT);
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -g[k];
        if (info == 0) {
            DN::solve(Hm, n, p, R, lds + L_SLOT);
            unpark();
            direction_scalars();
            return true;
        }
        unpark();
        if (!initial) { // Hessian is not positive definite, use steepest descent direction (NO.f90:1235-1236)
            phid = -gg;
            pp = gg;
            a = a * phidold / phid;
        }
        return false;
    }
    // BFGS: i=fdd(U,x,dim); call My_dpotri(U,dim,i); sycp(H,U); syL2U(H); p=-matmul(H,fdnew) (NO.f90:951-954)
    __device__ __forceinline__ bool bfgs_exact_refresh()
    {
        double *Hm = hist_base(), *U = Hm + (size_t)n * NPAD, *W = U + (size_t)n * NPAD;
        double *gbuf = lds + L_BF + 2 * NPAD; // third broadcast array: g waits here (the Cholesky kernels use the first)
        fill_hessian(U);
        park();
        if constexpr (PARK) {
            store_pad<NW, EPT>(gbuf, g);
            store_pad<NW, EPT>(W, x0); // wanted again only if the factorisation fails -- and then W is still untouched
        }
        const int info = DN::cholesky(U, n, lds + L_BF, lds + L_SLOT);
        if (info == 0) {
            DN::inverse_factor(U, W, n, lds + L_BF);
            DN::wtw(W, Hm, n, lds + L_BF);
        }
        if constexpr (PARK) load_pad<NW, EPT>(gbuf, g);
        if (info != 0) {
            if constexpr (PARK) load_pad<NW, EPT>(W, x0);
            unpark();
            return false;
        }
        ndef = 0; // pending updates belonged to the matrix that has just been replaced
        h_ident = 0;
        neg_matvec(Hm, gbuf);
        unpark();
        direction_scalars();
        return true;
    }

    // the line search returned: convergence tests on the new gradient, then the new direction
    __device__ __forceinline__ int after_linesearch()
    {
        a = ls.a;
        fnew = ls.fx;
        ++iters;
        if (gg < A.tol) { // NO.f90:174, 355, 612, 998
            status = FL_STATUS_CONVERGED;
            return inner_finished();
        }
        if (pp * a * a < A.minstep) { // "step length has converged"
            status = FL_STATUS_STEP_CONVERGED;
            return inner_finished();
        }
        if (iters >= max_linesearches()) {
            status = FL_STATUS_MAXIT;
            return inner_finished();
        }
        if constexpr (HESS_RCI && (METHOD == FL_SOLVER_NEWTON || METHOD == FL_SOLVER_BFGS)) {
            if (METHOD == FL_SOLVER_NEWTON || (A.exact_step > 0 && h_valid && main_it % A.exact_step == 0)) {
                phase = PH_HESS; // info=fdd(...,x,dim): the Hessian at the new x comes from the caller
                hess_stage = 1;
                return FL_REQ_H | FL_REQ_SAME;
            }
        }
        return GO_DIRECTION;
    }
    // After(): every ExactStep-th main-loop iteration BFGS tries the exact inverse Hessian first (i=mod(iIteration,freq),
    // NO.f90:949-956)
    __device__ __forceinline__ int direction_head()
    {
        if constexpr (METHOD == FL_SOLVER_BFGS && EXACT) {
            if (A.exact_step > 0 && h_valid && main_it % A.exact_step == 0) return GO_REFRESH;
        }
        return GO_DIRECTION_TAIL;
    }
    __device__ __forceinline__ int direction_and_begin(bool refreshed)
    {
        double g0[EPT];
        if constexpr (NEEDS_G0 && METHOD != FL_SOLVER_BFGS) load_pad<NW, EPT>(g0_park(), g0);
        if constexpr (METHOD == FL_SOLVER_SD) { // NO.f90:185-186
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = -g[k];
            phid = -gg;
            pp = gg;
            a = a * phidold / phid;
        } else if constexpr (METHOD == FL_SOLVER_CG) {
            direction_cg(g0);
        } else if constexpr (METHOD == FL_SOLVER_LBFGS) {
            direction_lbfgs(g0);
        } else if constexpr (METHOD == FL_SOLVER_NEWTON) {
            newton_direction(false);
        } else {
            // without a fresh exact inverse Hessian (not due, or not positive definite): the rank-2 update (NO.f90:957-963)
            if (!refreshed) {
                if constexpr (BF_DEFER > 0) direction_bfgs_deferred();
                else direction_bfgs();
            }
            h_valid = 1;
        }
        phid = uni(phid);
        pp = uni(pp);
        a = uni(a);
        return begin_linesearch();
    }

    // inner solver returned; with AUG run the outer update (NO.f90:2155-2157), else finish
    __device__ __forceinline__ int inner_finished()
    {
        if constexpr (AUG) {
            // call c(cx,x,M,N): cx of the last evaluation is c(x) (x is the last evaluated point)
            const int m = A.aug_m;
            double *cxs = cx_ptr();
            double c2 = 0.0;
            for (int j = 0; j < m; ++j) c2 = c2 + cxs[j] * cxs[j];
            cc = uni(c2);
            ++outer_it;
            inner_iters_total += iters;
            iters = 0;
            const double tolsq = uni(A.precision * A.precision); // (pinned to SGPRs: a loop invariant the compiler
                                                                 //  would otherwise carry in a VGPR pair for the whole solve)
            if (c2 < tolsq) { // if(dot_product(cx,cx)<tolsq) exit
                status = FL_STATUS_CONVERGED;
                phase = PH_DONE;
                return 0;
            }
            __syncthreads();
            const int tl = G::tid();
            if (tl < m) // lambda=lambda-miu*cx
                lds[L_LAM + tl] = lds[L_LAM + tl] - miu * cxs[tl];
            __syncthreads();
            miu = uni(miu * A.incr); // miu=miu*incrmt
            if (outer_it >= A.maxit) { // do iIteration=1,maxit exhausted
                status = FL_STATUS_MAXIT;
                phase = PH_DONE;
                return 0;
            }
            // staged launches: few problems are left running -- hand this one to the next launch, which gives it more waves
            if (STAGED && A.pause_below > 0) {
                const int fin = __builtin_amdgcn_readfirstlane(__hip_atomic_load(A.sched, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (A.batch - fin <= A.pause_below) {
                    status = FL_STATUS_PAUSED;
                    phase = PH_DONE;
                    return 0;
                }
            }
            // fresh inner solve from the current x: every solver starts with an evaluation of L, L'
            recent = -1;
            lrec = -1;
            cnt = 0;
            main_it = 0; // BFGS: a new call of the inner solver builds its inverse Hessian from H = a I again
            h_valid = 0;
            ndef = 0;
            h_ident = 0;
            phase = PH_INIT;
            return FL_REQ_F | FL_REQ_G | FL_REQ_NOMOVE;
        } else {
            phase = PH_DONE;
            return 0;
        }
    }

    // ---------------------------------------------------------------- directions
    __device__ __forceinline__ void direction_cg(const double (&g0)[EPT])
    {
        double yk[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) yk[k] = g[k] - g0[k];
        double beta;
        if (A.cg_method == FL_CG_DY) { // p=-g+(g.g)/((g-gold).p)*p, NO.f90:366
            double q[1] = {dot_part<EPT>(yk, p)};
            R.run(q);
            beta = gg / q[0];
        } else { // p=-g+(g.(g-gold))/(gold.gold)*p, NO.f90:387
            double q[2] = {dot_part<EPT>(g, yk), dot_part<EPT>(g0, g0)};
            R.run(q);
            beta = q[0] / q[1];
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -g[k] + beta * p[k];
        double q2[2] = {dot_part<EPT>(g, p), dot_part<EPT>(p, p)};
        R.run(q2);
        phid = q2[0];
        pp = q2[1];
        if (phid > 0.0) { // ascent direction: reset to steepest descent, NO.f90:368-370
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = -g[k];
            phid = -gg;
            pp = gg;
        }
        a = a * phidold / phid;
    }

    // The recursion without any of the traffic machinery: the new pair goes to the ring, every pair is fetched where it
    // is used, one row at a time.  Same arithmetic in the same order as direction_lbfgs (bit for bit).
    __device__ __forceinline__ void direction_lbfgs_plain(const double (&g0)[EPT])
    {
        double *hist = hist_base();
        double *rho_s = lds + L_RHO, *alpha_s = lds + L_ALPHA;
        const int mem = A.mem;
        recent = (recent + 1 == mem) ? 0 : recent + 1;
        if (cnt < mem) ++cnt;
        double r[2];
        {
            double sv[EPT], yv[EPT];
            if constexpr (X0_LDS) load_pad<NW, EPT>(lds + L_X0, sv);
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                sv[k] = x[k] - (X0_LDS ? sv[k] : x0[X0_LDS ? 0 : k]);
                yv[k] = g[k] - g0[k];
            }
            store_pad<NW, EPT>(hist + (size_t)(2 * recent) * NPAD, sv);
            store_pad<NW, EPT>(hist + (size_t)(2 * recent + 1) * NPAD, yv);
            r[0] = dot_part<EPT>(yv, sv);
            r[1] = dot_part<EPT>(yv, yv);
        }
        R.run(r);
        if (G::ltid() == 0) rho_s[recent] = 1.0 / r[0];
        rho_recent = uni(1.0 / r[0]);
        yy_recent = uni(r[1]);
        __syncthreads(); // rho_s; and the ring rows just stored are read back by the threads that wrote them
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = g[k];
        auto slot_of = [&](int j) {
            int s_ = recent - j;
            return s_ < 0 ? s_ + mem : s_;
        };
        for (int j = 0; j < cnt; ++j) { // newest -> oldest
            const int sl = slot_of(j);
            double row[EPT];
            load_pad<NW, EPT>(hist + (size_t)(2 * sl) * NPAD, row);
            double q[1] = {dot_part<EPT>(row, p)};
            R.run(q);
            const double al = rho_s[sl] * q[0];
            if (G::ltid() == 0) alpha_s[sl] = al;
            load_pad<NW, EPT>(hist + (size_t)(2 * sl + 1) * NPAD, row);
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = p[k] - al * row[k];
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = p[k] / rho_recent / yy_recent;
        __syncthreads(); // alpha_s
        for (int j = cnt - 1; j >= 0; --j) { // oldest -> newest
            const int sl = slot_of(j);
            double row[EPT];
            load_pad<NW, EPT>(hist + (size_t)(2 * sl + 1) * NPAD, row);
            double q[1] = {dot_part<EPT>(row, p)};
            R.run(q);
            const double be = rho_s[sl] * q[0];
            const double co = alpha_s[sl] - be;
            load_pad<NW, EPT>(hist + (size_t)(2 * sl) * NPAD, row);
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = p[k] + co * row[k];
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -p[k];
        r[0] = dot_part<EPT>(g, p);
        r[1] = dot_part<EPT>(p, p);
        R.run(r);
        phid = r[0];
        pp = r[1];
        a = 1.0;
    }
    __device__ __forceinline__ void direction_lbfgs(const double (&g0)[EPT])
    {
        if constexpr (AUG_LEAN18) {
            direction_lbfgs_plain(g0);
            return;
        }
        double *hist = hist_base();
        double *rho_s = lds + L_RHO, *alpha_s = lds + L_ALPHA;
        const int mem = A.mem;
        recent = (recent + 1 == mem) ? 0 : recent + 1; // recent=mod(recent+1,mem)
        if (cnt < mem) ++cnt;
        double r[2];
        // the newest pair stays in registers for both loops (its two row reads per loop never leave the CU)
        double sv[EPT], yv[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            sv[k] = x[k] - x0[k];
            yv[k] = g[k] - g0[k];
        }
        constexpr int K = LDS_PAIRS, RP = REG_PAIRS;
        if constexpr (RP > 0) {
            // the pair that leaves the registers joins the LDS ring (g_old was parked in that slot's y row and is
            // already in g0), the others move up, the new pair becomes the newest
            if (cnt > RP) { // (cnt counts the new pair: there were at least RP before it)
                if constexpr (K > 0) {
                    lrec = (lrec + 1 == K) ? 0 : lrec + 1;
                    store_pad<NW, EPT>(lds_pair(lrec), rs_[RP - 1]);
                    store_pad<NW, EPT>(lds_pair(lrec) + NPAD, ry_[RP - 1]);
                }
            }
#pragma unroll
            for (int r_ = RP - 1; r_ >= 1; --r_) {
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    rs_[r_][k] = rs_[r_ - 1][k];
                    ry_[r_][k] = ry_[r_ - 1][k];
                }
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                rs_[0][k] = sv[k];
                ry_[0][k] = yv[k];
            }
        } else if constexpr (K > 0) {
            lrec = (lrec + 1 == K) ? 0 : lrec + 1; // g_old was parked in this slot's y row (already in g0)
            store_pad<NW, EPT>(lds_pair(lrec), sv);
            store_pad<NW, EPT>(lds_pair(lrec) + NPAD, yv);
        }
        if (K + RP == 0 || mem > K + RP) { // a ring that fits on the chip never touches HBM
            store_pad<NW, EPT>(hist + (size_t)(2 * recent) * NPAD, sv);
            store_pad<NW, EPT>(hist + (size_t)(2 * recent + 1) * NPAD, yv);
        }
        r[0] = dot_part<EPT>(yv, sv);
        r[1] = dot_part<EPT>(yv, yv);
        R.run(r);
        if (G::ltid() == 0) rho_s[recent] = 1.0 / r[0]; // rho=1/(y.s): no curvature safeguard (NO.f90:623)
        rho_recent = uni(1.0 / r[0]);
        yy_recent = uni(r[1]);
        __syncthreads();

        // Before(): two-loop recursion, newest -> oldest, then oldest -> newest (NO.f90:586-608).
        // p = g in registers; the ring streams from HBM with the next step's rows in flight.
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = g[k];
        double sA[EPT], yA[EPT], sB[EPT], yB[EPT];
        auto slot_of = [&](int j) { // j-th newest
            int s = recent - j;
            return s < 0 ? s + mem : s;
        };
        // (the ring's rows are NOT streamed past the caches: non-temporal fetches take the headline from 155.3 ms to 181.6 on the
        // way down only, 161.0 on the way up only, 235 on both -- what is read on the way down comes back from the Infinity Cache
        // on the way up and in the next iteration; profiles/r04/c4_nontemporal_ab.txt)
        auto fetch = [&](int j, double (&s_)[EPT], double (&y_)[EPT]) { // j >= RP: the pairs behind the register pairs
            if (K > 0 && j < RP + K) { // in the LDS ring (separate branches keep ds_read / global_load apart)
                const int ls = lrec - (j - RP);
                const double *row = lds_pair(ls < 0 ? ls + K : ls);
                load_pad<NW, EPT>(row, s_);
                load_pad<NW, EPT>(row + NPAD, y_);
            } else {
                const double *row = hist + (size_t)(2 * slot_of(j)) * NPAD;
                load_pad<NW, EPT>(row, s_);
                load_pad<NW, EPT>(row + NPAD, y_);
            }
        };
        auto down = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
            const int sl = slot_of(j);
            double q[1] = {dot_part<EPT>(s_, p)};
            R.run(q);
            const double al = rho_s[sl] * q[0]; // alpha(i)=rho(i)*dot_product(s(:,i),p)
            if (G::ltid() == 0) alpha_s[sl] = al;
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = p[k] - al * y_[k];
        };
        auto upw = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
            const int sl = slot_of(j);
            double q[1] = {dot_part<EPT>(y_, p)};
            R.run(q);
            const double be = rho_s[sl] * q[0]; // phidnew=rho(i)*dot_product(y(:,i),p)
            const double co = alpha_s[sl] - be;
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = p[k] + co * s_[k];
        };
        // Row buffers: the pairs that have to be fetched alternate between B and A, starting with B at j = J0 (the
        // first pair behind the ones the way down has in registers: the new pair sv, yv and the register pairs).  The
        // two oldest pairs are still in the two buffers at the turn-around (last two steps down, first two steps up)
        // and are not fetched again; every other row is in flight one step ahead of its use.  On the way up the pairs j >= L come through the buffers
        // (j = 0 too when no pair is kept in registers), the register pairs last.
        constexpr int J0 = RP > 1 ? RP : 1, L = RP;
        if (cnt > J0) fetch(J0, sB, yB);
        down(0, sv, yv);
#pragma unroll
        for (int r_ = 1; r_ < RP; ++r_)
            if (cnt > r_) down(r_, rs_[r_], ry_[r_]);
        for (int j = J0; j < cnt; j += 2) {
            if (j + 1 < cnt) fetch(j + 1, sA, yA);
            down(j, sB, yB);
            if (j + 1 < cnt) {
                if (j + 2 < cnt) fetch(j + 2, sB, yB);
                down(j + 1, sA, yA);
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = p[k] / rho_recent / yy_recent; // p=p/rho(recent)/(y.y)
        __syncthreads(); // alpha_s written by thread 0 is visible (NW == 1 has no reduction barrier)
        int j = cnt - 1;
        if (RP == 0 && cnt == 1) fetch(0, sA, yA);
        // the pair BEFORE the oldest went through the other buffer on the way down and nothing has overwritten it:
        // the two oldest pairs are both used twice and fetched once
        bool prev_resident = (j - 1 >= J0);
        if (j >= L && (((j - J0) & 1) != 0)) { // the top pair waits in A
            if (j - 1 >= L && !prev_resident) fetch(j - 1, sB, yB);
            prev_resident = false;
            upw(j, sA, yA);
            --j;
        }
        for (; j >= L; j -= 2) { // j: resident / prefetched in B
            if (j - 1 >= L && !prev_resident) fetch(j - 1, sA, yA);
            prev_resident = false;
            upw(j, sB, yB);
            if (j - 1 >= L) {
                if (j - 2 >= L) fetch(j - 2, sB, yB);
                upw(j - 1, sA, yA);
            }
        }
#pragma unroll
        for (int r_ = RP - 1; r_ >= 1; --r_)
            if (cnt > r_) upw(r_, rs_[r_], ry_[r_]);
        if constexpr (RP >= 1) upw(0, rs_[0], ry_[0]);
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -p[k];
        r[0] = dot_part<EPT>(g, p);
        r[1] = dot_part<EPT>(p, p);
        R.run(r);
        phid = r[0]; // phidnew=dot_product(fdnew,p)
        pp = r[1];
        a = 1.0;
    }

    // BFGS without exact Hessian (ExactStep <= 0): rank-2 update of the inverse Hessian in its
    // O(n^2) form, H' = H - rho q s^T - rho s q^T + (rho^2 y.q + rho) s s^T, q = H y, which is
    // algebraically U^T (H U) + rho s s^T with U = I - rho y s^T (NO.f90:1010-1014; first step
    // H = a I, NO.f90:711-715).  Two streaming passes over the column-major H [n][NPAD]:
    //   pass 1: q = H y           (reads 8 n^2 B; "axpy" form, no reductions: q_i sums over j in order)
    //   pass 2: H' written in place while p = -H' g accumulates (reads + writes 16 n^2 B)
    __device__ __forceinline__ void direction_bfgs()
    {
        double *H = hist_base();
        double *bs = lds + L_BF, *bq = bs + NPAD, *bg = bq + NPAD;
        const bool first = !h_valid; // first quasi-Newton matrix from H = a I (NO.f90:711-715)
        double sv[EPT], yv[EPT], q[EPT];
        {
            double g0[EPT];
            load_pad<NW, EPT>(g0_park(), g0); // (the second broadcast array: read before bq is written below)
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                sv[k] = x[k] - x0[k]; // s=x-s
                yv[k] = g[k] - g0[k]; // y=fdnew-y
            }
        }
        park();
        double r1[1] = {dot_part<EPT>(yv, sv)};
        R.run(r1);
        const double rho = uni(1.0 / r1[0]); // rho=1d0/dot_product(y,s)
        __syncthreads();
        store_pad<NW, EPT>(bs, yv); // y broadcast for pass 1
        __syncthreads();
        if (first) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) q[k] = a * yv[k];
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k) q[k] = 0.0;
            for (int j = 0; j < n; j += BF_UNROLL) {
                double h[BF_UNROLL][EPT];
#pragma unroll
                for (int u = 0; u < BF_UNROLL; ++u)
                    if (j + u < n) load_pad_stream<NW, EPT>(H + (size_t)(j + u) * NPAD, h[u]);
#pragma unroll
                for (int u = 0; u < BF_UNROLL; ++u) {
                    if (j + u < n) {
                        const double yj = bs[j + u];
#pragma unroll
                        for (int k = 0; k < EPT; ++k) q[k] = q[k] + h[u][k] * yj;
                    }
                }
            }
        }
        double r2[1] = {dot_part<EPT>(yv, q)};
        R.run(r2);
        const double cs = uni(rho * rho * r2[0] + rho);
        __syncthreads();
        store_pad<NW, EPT>(bs, sv);
        store_pad<NW, EPT>(bq, q);
        store_pad<NW, EPT>(bg, g);
        __syncthreads();
        double rq_[EPT], rs_[EPT], cs_[EPT], acc[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            rq_[k] = rho * q[k];
            rs_[k] = rho * sv[k];
            cs_[k] = cs * sv[k];
            acc[k] = 0.0;
        }
        for (int j = 0; j < n; j += BF_UNROLL) {
            double h[BF_UNROLL][EPT];
#pragma unroll
            for (int u = 0; u < BF_UNROLL; ++u) {
                if (j + u < n) {
                    if (first) {
#pragma unroll
                        for (int k = 0; k < EPT; ++k)
                            h[u][k] = (G::e0(k >> 1) + (k & 1) == j + u) ? a : 0.0;
                    } else {
                        load_pad_stream<NW, EPT>(H + (size_t)(j + u) * NPAD, h[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BF_UNROLL; ++u) {
                if (j + u < n) {
                    const double sj = bs[j + u], qj = bq[j + u], gj = bg[j + u];
#pragma unroll
                    for (int k = 0; k < EPT; ++k) {
                        const double hn = h[u][k] - rq_[k] * sj - rs_[k] * qj + cs_[k] * sj;
                        h[u][k] = hn;
                        acc[k] = acc[k] + hn * gj;
                    }
                    store_pad_stream<NW, EPT>(H + (size_t)(j + u) * NPAD, h[u]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) p[k] = -acc[k]; // p=-matmul(H,fdnew)
        unpark();
        double r3[2] = {dot_part<EPT>(g, p), dot_part<EPT>(p, p)};
        R.run(r3);
        phid = r3[0];
        pp = r3[1];
        a = 1.0;
    }

    // The same update DEFERRED (BF_DEFER > 0; oracle update_form 100 + BF_DEFER): with the pending updates
    //   H_{l+1} = H_l - rho_l q_l s_l^T - rho_l s_l q_l^T + cs_l s_l s_l^T,  q_l = H_l y_l,   l = 0 .. ndef-1
    // kept as vectors, H_cur y and H_cur g are H y and H g -- ONE read pass with two accumulators per row -- plus
    // 4 dot products and two axpy-like corrections per pending update; the new update joins the list, p = -H_new g
    // follows algebraically, and every BF_DEFER-th iteration the list is folded into H element by element in the
    // order the updates occurred.  HBM bytes per iteration: 8 n^2 + 16 n^2 / BF_DEFER instead of 24 n^2.
    __device__ __forceinline__ void direction_bfgs_deferred()
    {
        double *H = hist_base(), *D = deferred_rows();
        double *by = lds + L_BF, *bg = by + NPAD;
        double *drho = lds + L_DEF, *dcs = drho + (BF_DEFER > 0 ? BF_DEFER : 1);
        double sv[EPT], yv[EPT], q[EPT], w[EPT];
        {
            double g0[EPT];
            load_pad<NW, EPT>(g0_park(), g0); // (= bg's array: read before g is broadcast below)
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                sv[k] = x[k] - x0[k];
                yv[k] = g[k] - g0[k];
            }
        }
        park();
        double r1[1] = {dot_part<EPT>(yv, sv)};
        R.run(r1);
        const double rho = uni(1.0 / r1[0]);
        if (!h_valid) { // first quasi-Newton matrix from H = a I (NO.f90:711-715): the identity stays implicit
            ndef = 0;
            h_ident = 1;
            a_id = a;
        }
        if (h_ident) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                q[k] = a_id * yv[k];
                w[k] = a_id * g[k];
            }
        } else {
            __syncthreads();
            store_pad<NW, EPT>(by, yv);
            store_pad<NW, EPT>(bg, g);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < EPT; ++k) q[k] = w[k] = 0.0;
            for (int j = 0; j < n; j += BF_UNROLL) {
                double h[BF_UNROLL][EPT];
#pragma unroll
                for (int u = 0; u < BF_UNROLL; ++u)
                    if (j + u < n) load_pad_stream<NW, EPT>(H + (size_t)(j + u) * NPAD, h[u]);
#pragma unroll
                for (int u = 0; u < BF_UNROLL; ++u) {
                    if (j + u < n) {
                        const double yj = by[j + u], gj = bg[j + u];
#pragma unroll
                        for (int k = 0; k < EPT; ++k) {
                            q[k] = q[k] + h[u][k] * yj;
                            w[k] = w[k] + h[u][k] * gj;
                        }
                    }
                }
            }
        }
        for (int l = 0; l < ndef; ++l) { // corrections of the pending updates, oldest first
            double S[EPT], Q[EPT];
            load_pad<NW, EPT>(D + (size_t)(2 * l) * NPAD, S);
            load_pad<NW, EPT>(D + (size_t)(2 * l + 1) * NPAD, Q);
            double r[4] = {dot_part<EPT>(S, yv), dot_part<EPT>(Q, yv), dot_part<EPT>(S, g), dot_part<EPT>(Q, g)};
            R.run(r);
            const double rl = drho[l], cl = dcs[l];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const double rq = rl * Q[k], rs = rl * S[k], cc_ = cl * S[k];
                q[k] = q[k] - rq * r[0] - rs * r[1] + cc_ * r[0];
                w[k] = w[k] - rq * r[2] - rs * r[3] + cc_ * r[2];
            }
        }
        double r3[3] = {dot_part<EPT>(yv, q), dot_part<EPT>(sv, g), dot_part<EPT>(q, g)};
        R.run(r3);
        const double cs = uni(rho * rho * r3[0] + rho);
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            p[k] = -(w[k] - (rho * q[k]) * r3[1] - (rho * sv[k]) * r3[2] + (cs * sv[k]) * r3[1]); // p=-matmul(H,fdnew)
        store_pad<NW, EPT>(D + (size_t)(2 * ndef) * NPAD, sv);
        store_pad<NW, EPT>(D + (size_t)(2 * ndef + 1) * NPAD, q);
        if (G::ltid() == 0) {
            drho[ndef] = rho;
            dcs[ndef] = cs;
        }
        ++ndef;
        __syncthreads(); // rho_l, cs_l and the rows of this update are visible to the workgroup
        if (ndef == BF_DEFER) {
            // the fold keeps the row factors of all pending updates in registers: g and p wait in the second and
            // third broadcast arrays meanwhile (the fold's staging area is inside the first)
            if constexpr (PARK) {
                store_pad<NW, EPT>(lds + L_BF + NPAD, g);
                store_pad<NW, EPT>(lds + L_BF + 2 * NPAD, p);
            }
            bfgs_fold();
            if constexpr (PARK) {
                load_pad<NW, EPT>(lds + L_BF + NPAD, g);
                load_pad<NW, EPT>(lds + L_BF + 2 * NPAD, p);
            }
        }
        unpark();
        double r4[2] = {dot_part<EPT>(g, p), dot_part<EPT>(p, p)};
        R.run(r4);
        phid = r4[0];
        pp = r4[1];
        a = 1.0;
    }
    // H <- H with the pending updates applied in order.  Each thread sweeps the columns once per 16-byte chunk of
    // its rows (the row factors of all pending updates for EPT rows at once would not fit the registers); the
    // s_l[j], q_l[j] of BF_FOLD_COLS columns at a time are staged in LDS.
    __device__ __forceinline__ void bfgs_fold()
    {
        constexpr int J = BF_DEFER > 0 ? BF_DEFER : 1, CB = BF_FOLD_COLS;
        double *H = hist_base(), *D = deferred_rows();
        double *stage = lds + L_BF; // [2*J][CB]
        const double *drho = lds + L_DEF, *dcs = drho + J;
        static_assert(BF_DEFER == 0 || 2 * J * CB <= NPAD, "staging area inside the first broadcast array");
        for (int c = 0; c < G::NCH; ++c) {
            const int e = G::e0(c);
            double rq[J][2], rs[J][2], cf[J][2];
#pragma unroll
            for (int l = 0; l < J; ++l) {
                const double2 S = *reinterpret_cast<const double2 *>(D + (size_t)(2 * l) * NPAD + e);
                const double2 Q = *reinterpret_cast<const double2 *>(D + (size_t)(2 * l + 1) * NPAD + e);
                const double rl = drho[l], cl = dcs[l];
                rq[l][0] = rl * Q.x; rq[l][1] = rl * Q.y;
                rs[l][0] = rl * S.x; rs[l][1] = rl * S.y;
                cf[l][0] = cl * S.x; cf[l][1] = cl * S.y;
            }
            for (int jb = 0; jb < n; jb += CB) {
                __syncthreads(); // the previous block's readers are done
                for (int i = G::tid(); i < 2 * J * CB; i += G::T) {
                    const int row = i / CB, col = i - row * CB;
                    stage[i] = (jb + col < n) ? D[(size_t)row * NPAD + jb + col] : 0.0;
                }
                __syncthreads();
                const int jend = (n - jb < CB) ? n - jb : CB;
                for (int jj = 0; jj < jend; ++jj) {
                    const int j = jb + jj;
                    double2 *hp = reinterpret_cast<double2 *>(H + (size_t)j * NPAD + e);
                    double ha, hb;
                    if (h_ident) {
                        ha = (e == j) ? a_id : 0.0;
                        hb = (e + 1 == j) ? a_id : 0.0;
                    } else {
#if FL_BFGS_NT >= 2
                        typedef double fl_d2 __attribute__((ext_vector_type(2)));
                        const fl_d2 t = __builtin_nontemporal_load(reinterpret_cast<const fl_d2 *>(hp));
#else
                        const double2 t = *hp;
#endif
                        ha = t.x;
                        hb = t.y;
                    }
#pragma unroll
                    for (int l = 0; l < J; ++l) {
                        const double sj = stage[(2 * l) * CB + jj], qj = stage[(2 * l + 1) * CB + jj];
                        ha = ha - rq[l][0] * sj - rs[l][0] * qj + cf[l][0] * sj;
                        hb = hb - rq[l][1] * sj - rs[l][1] * qj + cf[l][1] * sj;
                    }
#if FL_BFGS_NT >= 3
                    {
                        typedef double fl_d2s __attribute__((ext_vector_type(2)));
                        fl_d2s o;
                        o.x = ha;
                        o.y = hb;
                        __builtin_nontemporal_store(o, reinterpret_cast<fl_d2s *>(hp));
                    }
#else
                    *hp = make_double2(ha, hb);
#endif
                }
            }
        }
        __syncthreads();
        h_ident = 0;
        ndef = 0;
    }

    // ---------------------------------------------------------------- reverse communication
    // The machine parked in HBM between two launches of the step kernel (fl_rci.hip):
    // sc[RCI_SCALARS] doubles, vec[4][NPAD] = p, x0, gold, g, rho[FL_MAX_MEMORY].
    //
    // OWNERSHIP OF THE PARKED STATE (who may touch what, between which barriers).  Derived from the race of round 1
    // (n = 5001 through the legacy symbols: fused and reverse-communication L-BFGS differed by 1e-11; the scalar block
    // of a problem was rewritten by thread 0 while a slower wave was still reading it) -- keep it true when editing
    // rci_step_kernel, rci_step_big_kernel (fl_big.hpp has the same save / load pair) or line_search_step_kernel:
    //   1. One workgroup owns one problem's parked state for the whole launch; no other workgroup reads or writes it.
    //      Launches on the handle's stream are ordered, so a launch sees everything the previous one stored.
    //   2. VECTOR rows (vec, the history ring, the deferred-update rows, H): element e is loaded and stored only by
    //      the thread that owns e (Geo::e0 / BigSolver::e_of).  A thread therefore reads back its own stores in
    //      program order; no barrier is needed, and none may be assumed by code that makes a thread read ANOTHER
    //      thread's element (Rosenbrock's neighbours, the broadcast arrays of the dense phases, the fold's staging
    //      area: each of those has its own __syncthreads pair, written next to it).
    //   3. SCALAR block sc[] and rho[]: read by EVERY thread in load(), written by THREAD 0 ALONE in save() (rho: by
    //      the first FL_MAX_MEMORY threads after a barrier).  Between the two there must be a workgroup barrier that
    //      every wave passes AFTER its last read of sc[]: load() ends with that barrier.  Do not rely on the
    //      reductions' barriers for this -- a step that only takes an objective value (FL_REQ_F pending) performs no
    //      reduction at all before save().
    //   4. The early exit of a finished problem (phase == PH_DONE: the kernel returns right after load()) writes
    //      nothing but request[prob] = 0, by thread 0.
    //   5. LDS copies of parked values (g_old in g0_park(), rho in L_RHO) follow rule 2 / rule 3 respectively: the
    //      g_old row is per-thread private; rho_s[] is written before the barrier at the end of load() and by thread 0
    //      (one new entry) inside direction_lbfgs before its own barrier.
    // line_search_step_kernel keeps the same discipline with its own barrier after reading sc[] (fl_rci.hip).
    static constexpr int RCI_SCALARS = 48;
    __device__ __forceinline__ void save(double *sc, double *vec, double *rho, double fv_c, double pv_c)
    {
        if constexpr (RCI_LAZY) {
            if (ls_begun) { // (uniform) the only steps that change p and x0
                store_pad<NW, EPT>(vec, p);
                store_pad<NW, EPT>(vec + NPAD, x0);
            }
        } else {
            store_pad<NW, EPT>(vec, p);
            store_pad<NW, EPT>(vec + NPAD, x0);
            store_pad<NW, EPT>(vec + 3 * NPAD, g);
            if constexpr (NEEDS_G0) {
                double g0[EPT];
                load_pad<NW, EPT>(g0_park(), g0);
                store_pad<NW, EPT>(vec + 2 * NPAD, g0);
            }
        }
        if constexpr (METHOD == FL_SOLVER_LBFGS) {
            __syncthreads();
            if (threadIdx.x < FL_MAX_MEMORY) rho[threadIdx.x] = lds[L_RHO + threadIdx.x];
        }
        if (threadIdx.x == 0) {
            double *q = sc;
            *q++ = fnew; *q++ = gg; *q++ = pp; *q++ = phid; *q++ = phidold; *q++ = a;
            *q++ = yy_recent; *q++ = rho_recent; *q++ = fv_c; *q++ = pv_c;
            *q++ = ls.c1; *q++ = ls.c2abs; *q++ = ls.incr; *q++ = ls.fx0; *q++ = ls.phid0;
            *q++ = ls.a; *q++ = ls.aold; *q++ = ls.fx; *q++ = ls.fold; *q++ = ls.phidnew; *q++ = ls.phidold;
            *q++ = ls.low; *q++ = ls.up; *q++ = ls.flow; *q++ = ls.fup; *q++ = ls.phidlow; *q++ = ls.phidup;
            *q++ = ls.plma; *q++ = ls.a_eval;
            int *iq = reinterpret_cast<int *>(sc + 32);
            *iq++ = iters; *iq++ = nf; *iq++ = ng; *iq++ = status; *iq++ = phase; *iq++ = pending;
            *iq++ = recent; *iq++ = cnt; *iq++ = ls.st; *iq++ = ls.zret; *iq++ = ls.fused;
            *iq++ = main_it; *iq++ = h_valid; *iq++ = hess_stage;
            if constexpr (AUG) { // outer loop of the augmented Lagrangian (the multipliers themselves: A.lambda, below)
                sc[40] = miu;
                sc[41] = cc;
                int *aq = reinterpret_cast<int *>(sc + 42);
                aq[0] = outer_it;
                aq[1] = inner_iters_total;
            }
        }
        if constexpr (AUG) { // lambda lives in LDS during a launch and in the caller's array between launches
            __syncthreads();
            const int tl = G::tid();
            if (tl < A.aug_m) A.lambda[(size_t)prob * A.aug_m + tl] = lds[L_LAM + tl];
        }
    }
    __device__ __forceinline__ void load(const double *sc, const double *vec, const double *rho, double &fv_c,
                                         double &pv_c)
    {
        load_pad<NW, EPT>(vec, p);
        load_pad<NW, EPT>(vec + NPAD, x0);
        if constexpr (!RCI_LAZY) {
            load_pad<NW, EPT>(vec + 3 * NPAD, g);
            if constexpr (NEEDS_G0) {
                double g0[EPT];
                load_pad<NW, EPT>(vec + 2 * NPAD, g0);
                store_pad<NW, EPT>(g0_park(), g0);
            }
        }
        if constexpr (METHOD == FL_SOLVER_LBFGS) {
            if (threadIdx.x < FL_MAX_MEMORY) lds[L_RHO + threadIdx.x] = rho[threadIdx.x];
            __syncthreads();
        }
        const double *q = sc;
        fnew = *q++; gg = *q++; pp = *q++; phid = *q++; phidold = *q++; a = *q++;
        yy_recent = *q++; rho_recent = *q++; fv_c = *q++; pv_c = *q++;
        ls.c1 = *q++; ls.c2abs = *q++; ls.incr = *q++; ls.fx0 = *q++; ls.phid0 = *q++;
        ls.a = *q++; ls.aold = *q++; ls.fx = *q++; ls.fold = *q++; ls.phidnew = *q++; ls.phidold = *q++;
        ls.low = *q++; ls.up = *q++; ls.flow = *q++; ls.fup = *q++; ls.phidlow = *q++; ls.phidup = *q++;
        ls.plma = *q++; ls.a_eval = *q++;
        const int *iq = reinterpret_cast<const int *>(sc + 32);
        iters = *iq++; nf = *iq++; ng = *iq++; status = *iq++; phase = *iq++; pending = *iq++;
        recent = *iq++; cnt = *iq++; ls.st = *iq++; ls.zret = *iq++; ls.fused = *iq++;
        main_it = *iq++; h_valid = *iq++; hess_stage = *iq++;
        if constexpr (AUG) {
            miu = sc[40];
            cc = sc[41];
            const int *aq = reinterpret_cast<const int *>(sc + 42);
            outer_it = aq[0];
            inner_iters_total = aq[1];
        }
        if constexpr (LEAN) { // the parked scalars arrive as one copy per lane: pin them to scalar registers
            fnew = uni(fnew); gg = uni(gg); pp = uni(pp); phid = uni(phid); phidold = uni(phidold); a = uni(a);
            yy_recent = uni(yy_recent); rho_recent = uni(rho_recent);
            fv_c = uni(fv_c); pv_c = uni(pv_c);
            ls.template uniformize<2>();
            iters = __builtin_amdgcn_readfirstlane(iters); nf = __builtin_amdgcn_readfirstlane(nf);
            ng = __builtin_amdgcn_readfirstlane(ng); status = __builtin_amdgcn_readfirstlane(status);
            phase = __builtin_amdgcn_readfirstlane(phase); pending = __builtin_amdgcn_readfirstlane(pending);
            recent = __builtin_amdgcn_readfirstlane(recent); cnt = __builtin_amdgcn_readfirstlane(cnt);
            main_it = __builtin_amdgcn_readfirstlane(main_it); h_valid = __builtin_amdgcn_readfirstlane(h_valid);
            hess_stage = __builtin_amdgcn_readfirstlane(hess_stage);
        }
        // every wave has read the parked scalars before thread 0 may overwrite them in save(): a step that only
        // takes an objective value has no other barrier
        __syncthreads();
    }

    // ---------------------------------------------------------------- outputs
    __device__ __forceinline__ void finish()
    {
        store_user<NW, EPT>(A.x + (size_t)prob * n, n, x);
        if constexpr (AUG) {
            __syncthreads();
            const int tl = G::tid();
            if (tl < A.aug_m) A.lambda[(size_t)prob * A.aug_m + tl] = lds[L_LAM + tl];
        }
        if constexpr (STAGED) {
            if (A.sched && G::ltid() == 0) {
                if (status == FL_STATUS_PAUSED) {
                    double *ps = A.pstate + (size_t)prob * FL_PSTATE;
                    ps[0] = miu;
                    ps[1] = cc;
                    int *pi = reinterpret_cast<int *>(ps + 2);
                    pi[0] = outer_it; pi[1] = inner_iters_total; pi[2] = nf; pi[3] = ng;
                } else {
                    __hip_atomic_fetch_add(A.sched, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (G::ltid() == 0) {
            if (A.f_out) A.f_out[prob] = fnew;
            if (A.gg_out) A.gg_out[prob] = gg;
            if (A.iters) A.iters[prob] = iters + inner_iters_total;
            if (A.status) A.status[prob] = status;
            if (A.nf) A.nf[prob] = nf;
            if (A.ng) A.ng[prob] = ng;
            if constexpr (AUG) {
                if (A.outer) A.outer[prob] = outer_it;
                if (A.cnorm2) A.cnorm2[prob] = cc;
            }
        }
    }
};

} // namespace fl
