// fl_user_stream_objective.hpp -- YOUR objective inside the fused solver kernel for n > 4096 (C++ / HIP, header only).
//
// Beyond n = 4096 the solver's vectors live in HBM and one workgroup of 1024 threads owns one problem
// (fl_big.hpp).  The register path's functor (fl_user_objective.hpp: x and g as register arrays) has
// no meaning there; this path asks the objective ONE ELEMENT PAIR AT A TIME and does the loads, the stores of g, the g.p / g.g
// terms and the fixed-order sums around it (the STREAMING functor).  Reference interface: callbacks f, fd
// (NonlinearOptimization.f90:33-38).  Source text instead of a translation unit of yours: fl_user_compile with n > 4096
// (include/fl_nlopt.h) takes the same class.
//
// Include this header TWICE -- before your class (it needs fl::SolveArgs) and, with its name in FL_USER_STREAM_OBJECTIVE, after:
//
//     #include <hip/hip_runtime.h>
//     #include "fl_user_stream_objective.hpp"
//     struct MyObjective {                                   // a plain class
//         static constexpr bool NEIGHBOURS = false;          // true: pair() reads x of OTHER elements through `x` (the trial
//                                                            // point is then stored in a pass of its own, barriers around)
//         __device__ void init(const fl::SolveArgs &A, int prob);   // once per problem: A.n, A.d, A.b, A.user
//         __device__ void pair(int e, int n, const double *x, double xa, double xb,     // elements e (even), e + 1
//                              double &ta, double &tb, double &ua, double &ub, double &ga, double &gb);
//             // ta, tb: the elements' terms of the first sum s0; ua, ub: of the second sum s1; ga, gb: df/dx_e, df/dx_{e+1}
//             // an element >= n is padding (x = 0): do not read your arrays there; what you return for it is replaced by zeros
//         __device__ static double combine(double s0, double s1);   // f from the two sums
//     };
//     #define FL_USER_STREAM_OBJECTIVE MyObjective
//     #include "fl_user_stream_objective.hpp"
//     ...
//     size_t wsb = fl_workspace_bytes_for(FL_SOLVER_LBFGS, batch, n, &opt);
//     int rc = fl::user::solve_stream(FL_SOLVER_LBFGS, batch, n, x_dev, data0_dev, data1_dev, params_dev, &opt, ws_dev, wsb,
//                                     f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, stream);
//
// Solvers: FL_SOLVER_SD, FL_SOLVER_CG, FL_SOLVER_LBFGS (any n > 4096), FL_SOLVER_BFGS (quasi-Newton updates, n <= 16384).
// Summation order: thread t adds the terms of its pairs (c * 1024 + t) * 2, c = 0, 1, ..., then the workgroup's fixed tree
// (fl_reduction_geometry reports 1024 threads x 2 * slots): bit for bit reproducible, replayable by the oracle.
// One workgroup per problem here (the cooperative form -- several workgroups per problem for few huge problems -- is chosen
// by the library's own entries: fl_*_batched and fl_user_solve).  Compile with hipcc --offload-arch=gfx950 -ffp-contract=off,
// link with libFL.so.
#ifndef FL_USER_STREAM_OBJECTIVE_TYPES
#define FL_USER_STREAM_OBJECTIVE_TYPES
#if __has_include("fl/fl_solver_launch.hpp") // installed layout: prefix/include/fl/ (make install)
#include "fl/fl_solver_launch.hpp"
#else // the source tree
#include "../fortran-library_amd/csrc/fl_solver_launch.hpp"
#endif
#endif

#if defined(FL_USER_STREAM_OBJECTIVE) && !defined(FL_USER_STREAM_OBJECTIVE_SOLVE)
#define FL_USER_STREAM_OBJECTIVE_SOLVE
#if __has_include("fl/fl_big.hpp")
#include "fl/fl_big.hpp"
#else
#include "../fortran-library_amd/csrc/fl_big.hpp"
#endif

namespace fl {
namespace user {

inline int solve_stream(int solver, int batch, int n, double *x_dev, const double *data0_dev, const double *data1_dev, const void *params_dev,
                        const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev, double *gg_dev,
                        int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, hipStream_t stream)
{
    if (!x_dev || !opt || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (opt->cg_method != FL_CG_DY && opt->cg_method != FL_CG_PR) return FL_ERR_INVALID_ARGUMENT;
    if (solver != FL_SOLVER_SD && solver != FL_SOLVER_CG && solver != FL_SOLVER_LBFGS && solver != FL_SOLVER_BFGS)
        return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (n <= 4096 || fl_reduction_geometry_for(solver, n, &threads, &ept) != FL_OK || threads != 1024) return FL_ERR_UNSUPPORTED_SIZE;
    if (solver == FL_SOLVER_BFGS && n > BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>::BF_MAX_N) return FL_ERR_UNSUPPORTED_SIZE;
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
    // the machine's four rows per problem: behind the history in the workspace; SD / CG have none: the stream-ordered allocator
    const size_t npad = (size_t)threads * ept, rows_bytes = (size_t)batch * 4 * npad * sizeof(double);
    double *rows = nullptr;
    const bool own_rows = (solver == FL_SOLVER_SD || solver == FL_SOLVER_CG);
    if (solver == FL_SOLVER_LBFGS) rows = static_cast<double *>(workspace_dev) + (size_t)batch * 2 * (size_t)A.mem * npad;
    else if (solver == FL_SOLVER_BFGS)
        rows = static_cast<double *>(workspace_dev) + (size_t)batch * BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>::bfgs_rows(n) * npad;
    else if (hipMallocAsync((void **)&rows, rows_bytes, stream) != hipSuccess) {
        (void)hipGetLastError();
        return FL_ERR_WORKSPACE;
    }
#define FL_STREAM_K(M) hipLaunchKernelGGL((fl_big_solve_kernel<FL_OBJ_USER, M>), dim3(batch), dim3(1024), 0, stream, A, rows, 1, (double *)nullptr, (unsigned *)nullptr)
    switch (solver) {
    case FL_SOLVER_SD: FL_STREAM_K(FL_SOLVER_SD); break;
    case FL_SOLVER_CG: FL_STREAM_K(FL_SOLVER_CG); break;
    case FL_SOLVER_BFGS: FL_STREAM_K(FL_SOLVER_BFGS); break;
    default: FL_STREAM_K(FL_SOLVER_LBFGS); break;
    }
#undef FL_STREAM_K
    hipError_t e = hipGetLastError();
    if (own_rows) {
        const hipError_t ef = hipFreeAsync(rows, stream);
        if (e == hipSuccess) e = ef;
    }
    return launch_status(e);
}

} // namespace user
} // namespace fl
#endif
