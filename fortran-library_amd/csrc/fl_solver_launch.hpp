// fl_solver_launch.hpp -- the fused solver kernel (fl_solve_kernel) and its launchers by objective / solver for ONE
// geometry <waves, elements per thread>.  Every geometry is instantiated in a translation unit of its own
// (fl_solver_g*.hip: `template hipError_t fl::launch_o<NW, EPT>(...)`), so that the six of them compile in parallel
// (one unit for all took six minutes); fl_solver_kernels.hip holds the dispatch by n and the C ABI.
#pragma once
#include "fl_device.hpp"
#include "fl_host.hpp"

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
    while (rq) {
        // which evaluation the request needs (each form is inlined once)
        bool f_only = false, full = false;
        if (!(rq & FL_REQ_SAME)) {
            if (AUG && !(rq & FL_REQ_G)) f_only = true;
            else full = true;
            bool forwarded = false;
            if constexpr (SK > 1) { // an objective-only shrinking loop of the line search: run it as a tight loop to its exit
                if (f_only && s.spec_shrinking()) {
                    fv = s.template fast_forward<SK>(); // (leaves x at the exit trial's point)
                    have_g = false;
                    f_only = false;
                    forwarded = true;
                }
            }
            if constexpr (S::GROW_K > 1) { // the growing loop of StrongWolfe (f and f' per trial): GROW_K trials per pass
                if (full && !(rq & FL_REQ_NOMOVE) && s.grow_loop_pending()) {
                    s.template fast_forward_grow<S::GROW_K>(fv, pv); // (leaves x, g at the exit trial's point)
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
            s.template evaluate<false>(fv, pv, gg);
            have_g = false;
        }
        if (full) {
            s.template evaluate<true>(fv, pv, gg);
            have_g = true;
        }
        rq = s.advance(fv, pv, gg);
    }
    s.finish();
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
}

template <int NW, int EPT, int OBJ, int METHOD, int AUG, int EXACT = 0>
static hipError_t launch_k(const SolveArgs &A, hipStream_t st)
{
    hipLaunchKernelGGL((fl_solve_kernel<NW, EPT, OBJ, METHOD, AUG, EXACT>), dim3(A.batch), dim3(NW * 64), 0, st, A);
    return hipGetLastError();
}
template <int NW, int EPT, int OBJ> static hipError_t launch_m(int method, int aug, const SolveArgs &A, hipStream_t st)
{
    if (aug) { // augmented Lagrangian around L-BFGS, CG (NO.f90:2150-2185) or quasi-Newton BFGS (2131-2148, ExactStep <= 0)
        if (method == FL_SOLVER_CG) return launch_k<NW, EPT, OBJ, FL_SOLVER_CG, 1>(A, st);
        // NewtonRaphson / exact-Hessian BFGS around the Hessian of L (Ldd): up to n = 2048 (solve() refuses beyond: at 512
        // threads the Cholesky kernels, the deferred updates and the constraint terms together do not fit 256 VGPRs)
        if constexpr (NW < 8) {
            if (method == FL_SOLVER_NEWTON) return launch_k<NW, EPT, OBJ, FL_SOLVER_NEWTON, 1>(A, st); // fdd=Ldd (2074-2130)
            if (method == FL_SOLVER_BFGS && A.exact_step > 0) return launch_k<NW, EPT, OBJ, FL_SOLVER_BFGS, 1, 1>(A, st);
        }
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

} // namespace fl
