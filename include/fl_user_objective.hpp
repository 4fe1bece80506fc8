// fl_user_objective.hpp -- YOUR objective inside the fused solver kernels (C++ / HIP, header only).
//
// The reference takes the objective as callbacks: subroutine f(fx,x,dim), fd(g,x,dim), integer function f_fd(fx,g,x,dim)
// (NonlinearOptimization.f90:33-38).  On the GPU the equivalent of "pass your own f and fd" with no loss of speed is to
// COMPILE the objective into the solver kernel: one workgroup owns one problem from its initial guess to convergence, the
// iterate, gradient and search direction never leave registers, and a line-search trial costs one workgroup reduction
// (DESIGN.md 4.1).  The reverse-communication form (fl_rci_*, include/fl_nlopt.h) takes ANY objective -- a torch model,
// host callbacks -- but streams the machine through HBM on every evaluation and runs without the kernels' scheduling;
// this header gives a HIP objective the fused kernel itself: the same geometry, summation order, wave priority, on-chip
// L-BFGS pairs and lazy g.g as the built-in objectives (an objective that restates FL_OBJ_DIAGQUAD reproduces the
// built-in bit for bit: tests/test_gpu_user_objective.py).
//
// Usage (one translation unit of yours, compiled with hipcc --offload-arch=gfx950 -ffp-contract=off, linked with libFL.so):
//
//     #include <hip/hip_runtime.h>
//     template <int NW, int EPT> struct MyObjective;      // see the interface below
//     #define FL_USER_OBJECTIVE MyObjective
//     // #define FL_USER_TUNE_LIKE FL_OBJ_DIAGQUAD        // optional: element-wise, <= 2 data vectors in registers
//     #include "fl_user_objective.hpp"
//     ...
//     int threads, ept;  fl_reduction_geometry(n, &threads, &ept);            // e.g. n = 1024 -> 128 threads x 8
//     size_t wsb = fl_workspace_bytes(FL_SOLVER_LBFGS, batch, n, opt.memory); // L-BFGS ring / BFGS inverse Hessian
//     int rc = fl::user::solve<2, 8>(FL_SOLVER_LBFGS, batch, n, x_dev, data0_dev, data1_dev, params_dev, &opt,
//                                    ws_dev, wsb, f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, stream);
//
// <NW, EPT> = (threads / 64, ept) MUST be the geometry fl_reduction_geometry reports for n (the workspace layout and
// the kernels' fixed summation order depend on it); solve() checks.  Solvers: FL_SOLVER_SD, FL_SOLVER_CG,
// FL_SOLVER_LBFGS, FL_SOLVER_BFGS (quasi-Newton updates: exact_step is taken as 0; a Hessian would need hess_column).
// Outputs and options are those of fl_lbfgs_batched (include/fl_nlopt.h).
//
// The objective: a class template over the geometry.  Thread t of the NW*64 threads of the workgroup holds EPT elements
// of the problem's vectors in 16-byte chunks dealt round-robin: element e0(c) + j, e0(c) = (c * NW*64 + t) * 2, for
// chunk c < EPT/2, j < 2 -- as x[2c + j].  Elements beyond n are padding: they carry x = 0 and MUST produce g = 0 and
// zero terms of the sums.
//
//     template <int NW, int EPT> struct MyObjective {
//         static constexpr int LDS_DOUBLES = 0;     // doubles of LDS scratch your eval() wants (0 if none)
//         // once per problem (and again after a dense phase): keep what you need per element in members
//         __device__ void init(const fl::SolveArgs &A, int prob, double *lds);
//             // A.n; A.d, A.b, A.user: the three data pointers handed to solve(); fl::load_user<NW, EPT>(row, n, v)
//             // loads the thread's elements of a [batch][n] array's row (zero padded)
//         // the thread's part of the gradient and of up to two sums:  f = combine(sum_all s0, sum_all s1)
//         __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int n, double *lds);
//         __device__ static double combine(double s0, double s1);
//     };
//
// The sums over the workgroup are taken by the kernel in its fixed order (csrc/fl_reduce.hpp), so results are
// reproducible bit for bit from run to run and replayable on the CPU (oracle FLO_SUM_TREE order).  Use
// -ffp-contract=off if you want to compare with a CPU restatement bit for bit; eval() may use barriers
// (__syncthreads) -- every thread of the workgroup calls it the same number of times.
#pragma once
#ifndef FL_USER_OBJECTIVE
#error "define FL_USER_OBJECTIVE to your objective's class template before including fl_user_objective.hpp"
#endif
#if __has_include("fl/fl_solver_launch.hpp") // installed layout: prefix/include/fl/ (make install)
#include "fl/fl_solver_launch.hpp"
#else // the source tree
#include "../fortran-library_amd/csrc/fl_solver_launch.hpp"
#endif

namespace fl {
namespace user {

template <int NW, int EPT>
int solve(int solver, int batch, int n, double *x_dev, const double *data0_dev, const double *data1_dev, const void *params_dev,
          const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev, double *gg_dev,
          int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, hipStream_t stream)
{
    if (!x_dev || !opt || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (opt->cg_method != FL_CG_DY && opt->cg_method != FL_CG_PR) return FL_ERR_INVALID_ARGUMENT;
    if (solver != FL_SOLVER_SD && solver != FL_SOLVER_CG && solver != FL_SOLVER_LBFGS && solver != FL_SOLVER_BFGS)
        return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK || n > 4096) return FL_ERR_UNSUPPORTED_SIZE;
    if (threads != NW * 64 || ept != EPT) return FL_ERR_INVALID_ARGUMENT; // not the geometry of this n
    const int mem = opt->memory > 1 ? opt->memory : 1;
    if (solver == FL_SOLVER_LBFGS && mem > FL_MAX_MEMORY) return FL_ERR_UNSUPPORTED_SIZE;
    fl_options o = *opt;
    o.exact_step = 0;
    if (solver == FL_SOLVER_LBFGS || solver == FL_SOLVER_BFGS) {
        if (!workspace_dev || workspace_bytes < fl_workspace_bytes_for(solver, batch, n, &o)) return FL_ERR_WORKSPACE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    SolveArgs A;
    fill_solve_args(A, solver, batch, n, x_dev, data0_dev, data1_dev, &o, workspace_dev, f_dev, gg_dev, iters_dev, status_dev,
                    nf_dev, ng_dev);
    A.user = params_dev;
    hipError_t e;
    switch (solver) {
    case FL_SOLVER_SD: e = launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_SD, 0>(A, stream); break;
    case FL_SOLVER_CG: e = launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_CG, 0>(A, stream); break;
    case FL_SOLVER_BFGS: e = launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_BFGS, 0, 0>(A, stream); break;
    default: e = launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_LBFGS, 0>(A, stream); break;
    }
    return launch_status(e);
}

// AugmentedLagrangian (NO.f90:2005-2241) around the objective: with FL_USER_CONSTRAINTS defined (a class template with init /
// partial / add_gradient, csrc/fl_device.hpp) the caller's own constraints, m <= 8; without it the library's m block spheres.
// solver: FL_SOLVER_LBFGS | FL_SOLVER_CG (the inner solver); arguments as fl_augmented_lagrangian_batched.
template <int NW, int EPT>
int solve_auglag(int solver, int batch, int n, int m, double *x_dev, const double *data0_dev, const double *data1_dev, const void *params_dev,
                 double *lambda_dev, double miu0, const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                 double *cnorm2_dev, int32_t *iters_dev, int32_t *outer_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                 hipStream_t stream)
{
#ifdef FL_USER_CONSTRAINTS
    constexpr int A_ = FL_AUG_USER;
    if (m < 1 || m > FL_USER_MAX_CONSTRAINTS) return FL_ERR_INVALID_ARGUMENT;
#else
    constexpr int A_ = 1;
    if (m < 1 || m > FL_MAX_CONSTRAINTS || n % m != 0) return FL_ERR_INVALID_ARGUMENT;
#endif
    if (!x_dev || !opt || !lambda_dev || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (solver != FL_SOLVER_CG && solver != FL_SOLVER_LBFGS) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK || n > 4096) return FL_ERR_UNSUPPORTED_SIZE;
    if (threads != NW * 64 || ept != EPT) return FL_ERR_INVALID_ARGUMENT; // not the geometry of this n
    if (solver == FL_SOLVER_LBFGS && (!workspace_dev || workspace_bytes < fl_workspace_bytes_for(solver, batch, n, opt))) return FL_ERR_WORKSPACE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    SolveArgs A;
    fill_solve_args(A, solver, batch, n, x_dev, data0_dev, data1_dev, opt, workspace_dev, f_dev, nullptr, iters_dev, status_dev, nf_dev, ng_dev);
    A.user = params_dev;
    A.aug_m = m;
    A.miu0 = miu0;
    A.lambda = lambda_dev;
    A.outer = outer_dev;
    A.cnorm2 = cnorm2_dev;
    const hipError_t e = solver == FL_SOLVER_CG ? launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_CG, A_>(A, stream)
                                                : launch_k<NW, EPT, FL_OBJ_USER, FL_SOLVER_LBFGS, A_>(A, stream);
    return launch_status(e);
}

} // namespace user
} // namespace fl
