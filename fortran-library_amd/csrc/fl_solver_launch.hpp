// fl_solver_launch.hpp -- the fused solver kernel (fl_solve_kernel) and its launchers by objective / solver for ONE
// geometry <waves, elements per thread>.  Every geometry is instantiated in a translation unit of its own
// (fl_solver_g*.hip: `template hipError_t fl::launch_o<NW, EPT>(...)`), so that the six of them compile in parallel
// (one unit for all took six minutes); fl_solver_kernels.hip holds the dispatch by n and the C ABI.
#pragma once
#include "fl_device.hpp"
#ifndef __HIPCC_RTC__ // (run-time compilation takes the kernels only: the host side of a launch is csrc/fl_user_rtc.cpp)
#include "fl_host.hpp"
#endif

namespace fl {

// occupancy the register allocator is held to (waves per SIMD).  The augmented-Lagrangian L-BFGS / CG kernels at 4
// elements per thread are bound by the latency of their ~45 objective-only trials per gradient (C5): 130 VGPRs gave
// 3 waves, capped at 128 they run 4.
#ifndef FL_AUG18_WPE
#define FL_AUG18_WPE 3
#endif
template <int NW, int EPT, int OBJ, int METHOD, int AUG> constexpr int min_waves_per_simd()
{
#ifdef FL_MIN_WPE
    return FL_MIN_WPE;
#else
    // (the quartic / Rosenbrock instances would spill 10-14 VGPRs under the cap: left alone)
    if (AUG == FL_AUG_USER) return 1; // (the caller's constraints: no occupancy caps -- nothing can spill for them)
    if (AUG && EPT == 4 && NW <= 2 && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD && (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_CG)) return 4;
    if (FL_AUG_LEAN18 && AUG && EPT == 8 && NW == 1 && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD && (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_CG)) return FL_AUG18_WPE; // Solver::AUG_LEAN18 (4 waves: 80-100 spills)
    // SD / CG on them at 8 elements per thread, x0 in LDS (Solver::X0_LDS)
    if (!AUG && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD && EPT == 8 && NW >= 2 && (METHOD == FL_SOLVER_SD || METHOD == FL_SOLVER_CG) && FL_X0_LDS) return 4;
    // L-BFGS on them at 4 elements per thread (n <= 512): 134-135
    if (!AUG && tuned_like<OBJ>() == FL_OBJ_DIAGQUAD && EPT == 4 && METHOD == FL_SOLVER_LBFGS) return 4;
    return 1;
#endif
}

template <int NW, int EPT, int OBJ, int METHOD, int AUG, int EXACT = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(min_waves_per_simd<NW, EPT, OBJ, METHOD, AUG>())))
void fl_solve_kernel(SolveArgs A)
{
    using S = Solver<NW, EPT, OBJ, METHOD, AUG, EXACT>;
    if constexpr (S::STAGED) { // a listed launch (staged augmented Lagrangian) is sized for the most problems it can get
        if (A.list && (int)blockIdx.x >= A.sched[1]) return;
    }
#ifndef FL_LDS_PAD // tuning knob: extra LDS per workgroup caps the workgroups resident per CU
#define FL_LDS_PAD 0
#endif
    __shared__ __attribute__((aligned(16))) double lds[S::LDS_TOTAL + FL_LDS_PAD];
    S s(A, lds);
    s.init();
    int rq = s.start();
    double fv = 0.0, pv = 0.0, gg = 0.0;
    bool have_g = false; // augmented Lagrangian: objective-only trials skip the gradient until it is asked for
    constexpr int SK = S::SPEC_K;
#ifdef FL_PHASE_TIMERS // tuning builds only (tools/phase_timers.py): where a problem's wall time goes, in 10 ns ticks, to A.user
    long long tm[6] = {0, 0, 0, 0, 0, 0}, tc[6] = {0, 0, 0, 0, 0, 0};
#define FL_T0 const long long t0_ = wall_clock64()
#define FL_T1(i_) do { tm[i_] += wall_clock64() - t0_; ++tc[i_]; } while (0)
#else
#define FL_T0
#define FL_T1(i_)
#endif
    while (rq) {
        // which evaluation the request needs (each form is inlined once)
        bool f_only = false, full = false;
        if (!(rq & FL_REQ_SAME)) {
            if (AUG && !(rq & FL_REQ_G)) f_only = true;
            else full = true;
            bool forwarded = false;
            if constexpr (SK > 1) { // an objective-only shrinking loop of the line search: run it as a tight loop to its exit
                if (f_only && s.spec_shrinking()) {
                    FL_T0;
                    fv = s.template fast_forward<SK>(); // (leaves x at the exit trial's point)
                    FL_T1(0);
                    have_g = false;
                    f_only = false;
                    forwarded = true;
                }
            }
            if constexpr (S::GROW_K > 1) { // the growing loop of StrongWolfe (f and f' per trial): GROW_K trials per pass
                if (full && !(rq & FL_REQ_NOMOVE) && s.grow_loop_pending()) {
                    FL_T0;
                    s.template fast_forward_grow<S::GROW_K>(fv, pv); // (leaves x, g at the exit trial's point)
                    FL_T1(1);
                    have_g = true;
                    full = false;
                    forwarded = true;
                }
            }
            if (!forwarded && !(rq & FL_REQ_NOMOVE)) s.move(s.request_point());
        } else if (AUG && (rq & FL_REQ_G) && !have_g) { // gradient at the point whose objective is already known
            full = true;
        }
        if (AUG && f_only) {
            FL_T0;
            s.template evaluate<false>(fv, pv, gg);
            have_g = false;
            FL_T1(2);
        }
        if (full) {
            FL_T0;
            s.template evaluate<true>(fv, pv, gg);
            have_g = true;
            FL_T1(3);
        }
        {
            FL_T0;
#ifdef FL_PHASE_TIMERS
            const int it0 = s.iters + s.inner_iters_total;
#endif
            // an objective that is not a number ends the problem (Solver::not_finite).  Written so that the loop keeps its one
            // way out: a `break` here cost the dense BFGS kernel 130 more spilled SGPRs
            // an objective that is not a number ends the problem (Solver::not_finite).  Tested BEFORE advance() -- which then
            // takes one step of the machine with the NaN: a step has no loops -- and acted upon behind it, so that the loop keeps
            // its one way out and no value stays live across advance() for it (a `break` before advance() cost the dense BFGS
            // kernel 130 more spilled SGPRs, handing advance() a clean value the n <= 256 L-BFGS kernel 2 spilled VGPRs)
            const bool nanv = s.must_stop(fv); // (... or a zoom that never narrows: Solver::must_stop)
            rq = s.advance(fv, pv, gg);
            if (nanv) {
                s.stop_not_finite();
                rq = 0;
            }
#ifdef FL_PHASE_TIMERS
            if (s.iters + s.inner_iters_total != it0) FL_T1(5); // a line search ended: convergence tests + the new direction
            else FL_T1(4);
#endif
        }
    }
    s.finish();
#ifdef FL_PHASE_TIMERS
    if (A.user && threadIdx.x == 0) {
        long long *o = (long long *)A.user + (size_t)blockIdx.x * 12;
        for (int i = 0; i < 6; ++i) { o[i] = tm[i]; o[6 + i] = tc[i]; }
    }
#endif
}

// A master wave and REP - 1 helper waves per problem (Solver::fast_forward_wide_cs): the master runs the machine exactly as
// fl_solve_kernel<1, EPT, ...> does, the helpers take their share of the objective-only shrink loop of the line search --
// trial-parallel -- and otherwise wait at the workgroup barrier.  For the augmented-Lagrangian kernels whose geometry is one
// wave (n <= 512), picked by the host for batches that under-fill the chip.  Results are bit for bit fl_solve_kernel's.
// Barrier protocol: a helper sits in `for (;;) { barrier; read cmd; ... }`, so EVERY barrier the master executes outside the
// shared loop is matched by one turn of that loop; cmd (LDS, written by the master's lane 0 before the barrier that publishes
// it) = 1: a shrink loop starts behind this barrier -- both sides then execute the same passes, the same exits, hence the
// same barriers -- = 2: the problem is finished (a terminated wave no longer counts at s_barrier), = 0: nothing.
// Registers: as the allocator likes (199 VGPRs for BASELINE config 5's kernel = two waves per SIMD).  Held to the unhelped
// kernel's three waves per SIMD (168) it spills 7 VGPRs and is slower wherever helpers are used at all (batch 256: 30.7
// against 28.2 ms with three helpers, 43.6 against 39.0 with one; profiles/r04/geometry_by_batch.txt, last two tables).
template <int REP, int NW, int EPT, int OBJ, int METHOD, int AUG>
__global__ __launch_bounds__(REP * NW * 64) void fl_solve_rep_kernel(SolveArgs A)
{
    using S = Solver<NW, EPT, OBJ, METHOD, AUG, 0>;
    static_assert(NW == 1 && AUG && S::SPEC_K > 1, "helpers share out the speculative objective-only trials of a one-wave machine");
    static_assert(S::Obj::LDS_DOUBLES == 0, "the helpers keep no LDS image of the objective");
    constexpr int SK = S::SPEC_K;
    constexpr int LT = (S::LDS_TOTAL + 1) & ~1; // the master's LDS
    constexpr int HL = (S::L_XS + 1) & ~1;      // a helper's: the machine's small arrays (lambda, c(x) buffers of its trials)
    constexpr int PUB = (S::PUB_DOUBLES + 2 + 1) & ~1; // what the master publishes per loop, then the command word
    __shared__ __attribute__((aligned(16))) double lds[LT + (REP - 1) * HL + PUB + 4 * SK * REP + REP + 2];
    if (A.list && (int)blockIdx.x >= A.sched[1]) return; // (a listed launch is sized for the most problems it can get)
    const int rep = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *pub = lds + LT + (REP - 1) * HL, *xch = pub + PUB;
    volatile int *cmd = reinterpret_cast<volatile int *>(pub + S::PUB_DOUBLES);
    const bool lane0 = (threadIdx.x & 63) == 0, master = rep == 0;
    // ONE machine object and ONE copy of the shared loop for both roles
    S s(A, master ? lds : lds + LT + (rep - 1) * HL);
    if (master && lane0) *cmd = 0;
    s.init();
    int rq = master ? s.start() : 0;
    double fv = 0.0, pv = 0.0, gg = 0.0;
    bool have_g = false;
    for (;;) { // one turn = one request of the master's machine (one call site of advance(): it is inlined once)
        int c = 0; // 1: a shrink loop is shared out behind the next barrier, 2: the problem is finished
        bool f_only = false, full = false;
        if (master) {
            if (!rq) {
                c = 2;
            } else if (!(rq & FL_REQ_SAME)) {
                if (!(rq & FL_REQ_G)) f_only = true;
                else full = true;
                if (f_only && s.spec_shrinking()) c = 1;
                else if (!(rq & FL_REQ_NOMOVE)) s.move(s.request_point());
            } else if ((rq & FL_REQ_G) && !have_g) {
                full = true;
            }
            if (c) {
                if (c == 1) s.publish_search(pub);
                if (lane0) *cmd = c;
                __syncthreads();
            }
        } else {
            do { // every barrier the master executes on its way is matched by one turn
                __syncthreads();
                c = __builtin_amdgcn_readfirstlane(*cmd);
            } while (c == 0);
            if (c == 1) s.helper_take(pub);
        }
        if (c == 2) break;
        if (c == 1) {
            const double fx = s.template fast_forward_wide<SK, REP>(xch, rep, lds + S::L_X0, master);
            if (master) {
                if (lane0) *cmd = 0; // (behind the loop's last barrier; the helpers read it behind the master's next one)
                fv = fx;
                have_g = false;
                f_only = false;
            }
        }
        if (master) {
            if (f_only) {
                s.template evaluate<false>(fv, pv, gg);
                have_g = false;
            }
            if (full) {
                s.template evaluate<true>(fv, pv, gg);
                have_g = true;
            }
            // (the objective is not a number, or the zoom never narrows: the problem ends here, in the form -- and so with the
            // counters -- of fl_solve_kernel's loop; the helpers are released at the top of the next turn)
            const bool stop = s.must_stop(fv);
            rq = s.advance(fv, pv, gg);
            if (stop) {
                s.stop_not_finite();
                rq = 0;
            }
        }
    }
    if (master) s.finish();
}
#ifndef __HIPCC_RTC__ // ---- the host side of the launches (not part of a run-time compilation)
template <int REP, int NW, int EPT, int OBJ, int METHOD>
static hipError_t launch_rep_k(const SolveArgs &A, hipStream_t st)
{
    hipLaunchKernelGGL((fl_solve_rep_kernel<REP, NW, EPT, OBJ, METHOD, 1>), dim3(A.list ? A.pause_grid : A.batch), dim3(REP * NW * 64), 0, st, A);
    return hipGetLastError();
}
// (fl_solver_g*r.hip) objective: FL_OBJ_DIAGQUAD | FL_OBJ_QUARTIC, method: FL_SOLVER_LBFGS | FL_SOLVER_CG
template <int NW, int EPT> hipError_t launch_rep(int rep, int obj, int method, const SolveArgs &A, hipStream_t st)
{
#define FL_REP(R_, O_)                                                                                     \
    return method == FL_SOLVER_CG ? launch_rep_k<R_, NW, EPT, O_, FL_SOLVER_CG>(A, st) : launch_rep_k<R_, NW, EPT, O_, FL_SOLVER_LBFGS>(A, st)
    if (rep == 2) {
        if (obj == FL_OBJ_QUARTIC) FL_REP(2, FL_OBJ_QUARTIC);
        FL_REP(2, FL_OBJ_DIAGQUAD);
    }
    if (obj == FL_OBJ_QUARTIC) FL_REP(4, FL_OBJ_QUARTIC);
    FL_REP(4, FL_OBJ_DIAGQUAD);
#undef FL_REP
}

// the reference's optional arguments -> the kernel's arguments (defaults and clamps of NO.f90:419-434); shared by the
// built-in entry points (fl_solver_kernels.hip) and a caller-compiled objective (include/fl_user_objective.hpp)
static inline void fill_solve_args(SolveArgs &A, int method, int batch, int n, double *x, const double *d, const double *b,
                                   const fl_options *opt, void *ws, double *f, double *gg, int32_t *iters, int32_t *status,
                                   int32_t *nf, int32_t *ng)
{
    A.n = n;
    A.batch = batch;
    A.mem = opt->memory > 1 ? opt->memory : 1; // mem=max(1,Memory)
    A.exact_step = (method == FL_SOLVER_BFGS) ? opt->exact_step : 0;
    A.maxit = opt->max_iteration;
    A.strong = opt->strong != 0;
    A.fused = opt->fused_f_fd != 0;
    A.cg_method = opt->cg_method;
    A.tol = opt->precision * opt->precision;                 // NO.f90:427
    A.minstep = opt->min_step_length * opt->min_step_length; // NO.f90:429
    A.c1 = opt->wolfe_c1;
    A.c2 = opt->wolfe_c2;
    if (opt->clamp) { // NO.f90:431-434
        A.c1 = opt->wolfe_c1 > 1e-15 ? opt->wolfe_c1 : 1e-15;
        const double lo = A.c1 + 1e-15;
        const double c2 = opt->wolfe_c2 > lo ? opt->wolfe_c2 : lo;
        A.c2 = c2 < 1.0 - 1e-15 ? c2 : 1.0 - 1e-15;
    }
    A.incr = opt->increment;
    A.x = x;
    A.d = d;
    A.b = b;
    A.hist = static_cast<double *>(ws);
    A.f_out = f;
    A.gg_out = gg;
    A.iters = iters;
    A.status = status;
    A.nf = nf;
    A.ng = ng;
    A.aug_m = 0;
    A.miu0 = 1.0;
    A.precision = opt->precision;
    A.lambda = nullptr;
    A.outer = nullptr;
    A.cnorm2 = nullptr;
    A.user = nullptr;
    A.list = nullptr;
    A.sched = nullptr;
    A.pstate = nullptr;
    A.pause_below = 0;
    A.resume = 0;
    A.pause_grid = 0;
#ifdef FL_PHASE_TIMERS
    if (const char *e = getenv("FL_PHASE_BUFFER")) A.user = (const void *)strtoull(e, nullptr, 16); // [batch][12] int64, device
#endif
}

template <int NW, int EPT, int OBJ, int METHOD, int AUG, int EXACT = 0>
static hipError_t launch_k(const SolveArgs &A, hipStream_t st)
{
    hipLaunchKernelGGL((fl_solve_kernel<NW, EPT, OBJ, METHOD, AUG, EXACT>), dim3(A.list ? A.pause_grid : A.batch), dim3(NW * 64), 0, st, A);
    return hipGetLastError();
}
template <int NW, int EPT, int OBJ> static hipError_t launch_m(int method, int aug, const SolveArgs &A, hipStream_t st)
{
    if (aug) { // augmented Lagrangian around L-BFGS, CG (NO.f90:2150-2185) or quasi-Newton BFGS (2131-2148, ExactStep <= 0)
        if (method == FL_SOLVER_CG) return launch_k<NW, EPT, OBJ, FL_SOLVER_CG, 1>(A, st);
        // NewtonRaphson / exact-Hessian BFGS around the Hessian of L (Ldd).  Every n of the register path since round 4: at
        // 512 threads (n > 2048) the Cholesky kernels, the deferred updates and the constraint terms together fill the 256
        // VGPRs exactly for the quadratic and Rosenbrock; the quartic's two kernels spill (22 / 130 registers, outside the
        // streaming passes over the matrix -- tests/test_kernel_resources.py names them)
        if (method == FL_SOLVER_NEWTON) return launch_k<NW, EPT, OBJ, FL_SOLVER_NEWTON, 1>(A, st); // fdd=Ldd (2074-2130)
        if (method == FL_SOLVER_BFGS && A.exact_step > 0) return launch_k<NW, EPT, OBJ, FL_SOLVER_BFGS, 1, 1>(A, st);
        if (method == FL_SOLVER_BFGS) return launch_k<NW, EPT, OBJ, FL_SOLVER_BFGS, 1, 0>(A, st);
        return launch_k<NW, EPT, OBJ, FL_SOLVER_LBFGS, 1>(A, st);
    }
    switch (method) {
    case FL_SOLVER_SD: return launch_k<NW, EPT, OBJ, FL_SOLVER_SD, 0>(A, st);
    case FL_SOLVER_CG: return launch_k<NW, EPT, OBJ, FL_SOLVER_CG, 0>(A, st);
    case FL_SOLVER_BFGS: // the exact-Hessian refresh (Cholesky kernels) is compiled into its own instantiation
        if (A.exact_step > 0) return launch_k<NW, EPT, OBJ, FL_SOLVER_BFGS, 0, 1>(A, st);
        return launch_k<NW, EPT, OBJ, FL_SOLVER_BFGS, 0, 0>(A, st);
    case FL_SOLVER_NEWTON: return launch_k<NW, EPT, OBJ, FL_SOLVER_NEWTON, 0>(A, st);
    default: return launch_k<NW, EPT, OBJ, FL_SOLVER_LBFGS, 0>(A, st);
    }
}
template <int NW, int EPT> hipError_t launch_o(int obj, int method, int aug, const SolveArgs &A, hipStream_t st)
{
    switch (obj) {
    case FL_OBJ_QUARTIC: return launch_m<NW, EPT, FL_OBJ_QUARTIC>(method, aug, A, st);
    case FL_OBJ_ROSENBROCK: return launch_m<NW, EPT, FL_OBJ_ROSENBROCK>(method, aug, A, st);
    default: return launch_m<NW, EPT, FL_OBJ_DIAGQUAD>(method, aug, A, st);
    }
}

// SD / CG without constraints only: the geometries that exist for them alone (fl_solver_g116.hip)
template <int NW, int EPT, int OBJ> static hipError_t launch_vec_m(int method, const SolveArgs &A, hipStream_t st)
{
    if (method == FL_SOLVER_SD) return launch_k<NW, EPT, OBJ, FL_SOLVER_SD, 0>(A, st);
    return launch_k<NW, EPT, OBJ, FL_SOLVER_CG, 0>(A, st);
}
template <int NW, int EPT> hipError_t launch_vec(int obj, int method, const SolveArgs &A, hipStream_t st)
{
    switch (obj) {
    case FL_OBJ_QUARTIC: return launch_vec_m<NW, EPT, FL_OBJ_QUARTIC>(method, A, st);
    case FL_OBJ_ROSENBROCK: return launch_vec_m<NW, EPT, FL_OBJ_ROSENBROCK>(method, A, st);
    default: return launch_vec_m<NW, EPT, FL_OBJ_DIAGQUAD>(method, A, st);
    }
}

// NewtonRaphson only (with or without the augmented Lagrangian around it): fl_solver_g24.hip
template <int NW, int EPT> hipError_t launch_newton(int obj, int aug, const SolveArgs &A, hipStream_t st)
{
#define FL_NEWTON(O_)                                                                      \
    return aug ? launch_k<NW, EPT, O_, FL_SOLVER_NEWTON, 1>(A, st) : launch_k<NW, EPT, O_, FL_SOLVER_NEWTON, 0>(A, st)
    switch (obj) {
    case FL_OBJ_QUARTIC: FL_NEWTON(FL_OBJ_QUARTIC);
    case FL_OBJ_ROSENBROCK: FL_NEWTON(FL_OBJ_ROSENBROCK);
    default: FL_NEWTON(FL_OBJ_DIAGQUAD);
    }
#undef FL_NEWTON
}
#endif // __HIPCC_RTC__

} // namespace fl
