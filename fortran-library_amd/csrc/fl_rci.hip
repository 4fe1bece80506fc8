// fl_rci.hip -- reverse-communication ("ask / tell") form of the batched solvers, and on
// top of it the reference's legacy single-problem entry points with HOST callbacks.
//
// The reference calls user code for every evaluation: subroutine f(fx,x,dim), fd(g,x,dim),
// integer function f_fd(fx,g,x,dim) (NO.f90:33-38).  A GPU solver cannot call back into the
// host from a kernel, so the machine of fl_device.hpp is parked in HBM between launches:
//     fl_rci_step():  [take the caller's f, g at the requested points] -> advance every
//                     problem's machine to its next request -> write the next trial points
// The caller (host code with callbacks, or a torch / HIP objective for the whole batch)
// evaluates exactly what request[] asks for and calls fl_rci_step again.  All vector
// arithmetic (x0 + a p, dot products, two-loop recursion, BFGS update) runs in the kernel;
// there is no CPU solver behind these entry points.
//
// Legacy symbols (cpp/NonlinearOptimization.hpp:278-393 binds them; gfortran and ifort
// manglings): steepestdescent, conjugategradient(_basic), lbfgs, bfgs -- batch of one,
// callbacks evaluated on the host, x copied device <-> host per evaluation.
#include "fl_device.hpp"
#include "fl_host.hpp"
#include "fl_big.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <new>
#include <algorithm>
#include <vector>

namespace fl {

// flags of a step
#define FL_RCI_BOTH 1 // the caller evaluated f AND grad f at every requested point, whatever the request bits said

// One step of every (active) problem.  slot = blockIdx.x indexes the caller's arrays of this step (f, g, request, and the
// points xio: the evaluated point in, the next requested point out); prob = active ? active[slot] : slot is the problem,
// i.e. where its parked machine, its history and its final answer x[prob][:] live.  Without an active list the two
// coincide and xio = x (the classic fl_rci_step).
template <int NW, int EPT, int METHOD, int AUG = 0>
__global__ __launch_bounds__(NW * 64) void rci_step_kernel(SolveArgs A, int first, int flags, const int32_t *active, double *xio,
                                                           double *sc_all, double *vec_all, double *rho_all, const double *f_dev,
                                                           const double *g_dev, const double *c_dev, const double *cd_dev,
                                                           int32_t *request)
{
    using S = Solver<NW, EPT, FL_OBJ_EXTERNAL, METHOD, AUG>;
    __shared__ __attribute__((aligned(16))) double lds[S::LDS_TOTAL];
    const int slot = blockIdx.x, n = A.n;
    const int prob = active ? active[slot] : slot;
    double *sc = sc_all + (size_t)prob * S::RCI_SCALARS;
    double *vec = vec_all + (size_t)prob * 4 * S::NPAD;
    double *rho = rho_all + (size_t)prob * FL_MAX_MEMORY;
    // a finished problem costs nothing but this look at its parked phase (every thread reads the same word; thread 0
    // alone rewrites the scalar block, and only after load()'s barrier: see the ownership rules above Solver::save)
    if (!first && reinterpret_cast<const int *>(sc + 32)[4] == S::PH_DONE) {
        if (threadIdx.x == 0) request[slot] = 0;
        return;
    }
    S s(A, lds);
    s.prob = prob;
    double *xrow = xio + (size_t)slot * n;
    s.init(xrow); // x = the point the caller has just evaluated (or the initial guess)
    s.rci_vec = vec;
    int rq;
    double fv = 0.0, pv = 0.0;
    if (first) {
        rq = s.start();
        s.ls_begun = true; // (the parked rows p, x0 start defined: zero direction, the initial guess)
#pragma unroll
        for (int k = 0; k < EPT; ++k) s.x0[k] = s.x[k];
    } else {
        s.load(sc, vec, rho, fv, pv);
        double ggv = s.gg;
        const bool both = (flags & FL_RCI_BOTH) != 0;
        const bool have_f = ((s.pending & FL_REQ_F) || both) && f_dev, have_g = ((s.pending & FL_REQ_G) || both) && g_dev;
        if constexpr (AUG) {
            // the caller's f, grad f, c, cd -> the augmented Lagrangian and its gradient (c comes with every request,
            // cd with every gradient request: the request bits FL_REQ_C / FL_REQ_CD say so)
            if (have_g) load_user<NW, EPT>(g_dev + (size_t)slot * n, n, s.g);
            s.take_external_aug(have_f ? f_dev[slot] : 0.0, have_f, have_g, c_dev + (size_t)slot * A.aug_m,
                                cd_dev + (size_t)slot * A.aug_m * n, fv, pv, ggv);
        } else {
            if (have_f) fv = f_dev[slot];
            if (have_g) {
                load_user<NW, EPT>(g_dev + (size_t)slot * n, n, s.g);
                double q[2] = {dot_part<EPT>(s.g, s.p), dot_part<EPT>(s.g, s.g)};
                s.R.run(q);
                pv = q[0];
                ggv = q[1];
            }
        }
        // FL_RCI_BOTH: a request for the other quantity AT THE SAME POINT (StrongWolfe: f, then f' once Armijo holds,
        // NO.f90:1483-1485) is answered on the spot instead of costing the caller another round
        // (SD / CG / L-BFGS; the dense solvers' kernels have no registers to spare for a loop around the machine)
        if constexpr (S::RCI_LAZY) {
            bool again;
            do {
                rq = s.advance(fv, pv, ggv);
                again = both && rq != 0 && (rq & FL_REQ_SAME);
            } while (again);
        } else {
            rq = s.advance(fv, pv, ggv);
        }
    }
    if (rq == 0) {
        s.finish();
    } else if (!(rq & (FL_REQ_SAME | FL_REQ_NOMOVE))) {
        s.move(s.request_point());
        store_user<NW, EPT>(xrow, n, s.x);
    }
    s.save(sc, vec, rho, fv, pv);
    if (threadIdx.x == 0) {
        int out = rq;
        if (AUG && rq != 0) out |= FL_REQ_C | ((rq & FL_REQ_G) ? FL_REQ_CD : 0);
        request[slot] = out;
    }
}

// n > 4096 (fl_big.hpp): the same step with the machine's vectors in HBM, one workgroup of 1024 threads per problem
// groups > 1: the COOPERATIVE form -- `groups` workgroups share one problem (BigSolver::set_cooperative): blockIdx.x =
// problem * groups + group.  For few problems of very large n (the reference's callers typically solve ONE problem of
// any dim): a single n = 2^20 problem then occupies the whole chip instead of one CU.
// AUG = 1: the augmented Lagrangian around the solver with the caller's c [batch][m] and cd [batch][m][n] (round 4: until then the
// constrained form stopped at n = 4096; one workgroup per problem -- lambda has one copy, so no cooperative form)
template <int METHOD, int AUG = 0>
__global__ __launch_bounds__(1024) void rci_step_big_kernel(SolveArgs A, int first, double *sc_all, double *vec_all,
                                                            double *rho_all, const double *f_dev, const double *g_dev,
                                                            int32_t *request, int groups, double *coop_part,
                                                            unsigned *coop_counter, int parity, unsigned *coop_host_flag,
                                                            const double *c_dev = nullptr, const double *cd_dev = nullptr)
{
    using S = BigSolver<FL_OBJ_EXTERNAL, METHOD, AUG>;
    __shared__ __attribute__((aligned(16))) double lds[S::LDS_TOTAL];
    const int prob = blockIdx.x / groups, group = blockIdx.x - prob * groups, n = A.n;
    S s(A, lds, vec_all, prob);
    if (groups > 1)
        s.set_cooperative(groups, group, coop_part + (size_t)prob * 2 * groups * Reducer<S::NW>::NVMAX, coop_counter + 2 * prob, coop_host_flag);
    // cooperative form: the scalars (and the rho ring) are parked in two copies used alternately -- this step reads copy
    // `parity` and its first workgroup writes the other one, so no workgroup can see a half-written block and no barrier
    // is needed between load() and save() (a step that only takes an objective value has none of its own)
    const size_t two = groups > 1 ? 2 : 1, in = groups > 1 ? (size_t)parity : 0, out = groups > 1 ? (size_t)(1 - parity) : 0;
    const double *sc_in = sc_all + ((size_t)prob * two + in) * S::RCI_SCALARS, *rho_in = rho_all + ((size_t)prob * two + in) * FL_MAX_MEMORY;
    double *sc = sc_all + ((size_t)prob * two + out) * S::RCI_SCALARS, *rho = rho_all + ((size_t)prob * two + out) * FL_MAX_MEMORY;
    s.init();
    int rq;
    double fv = 0.0, pv = 0.0;
    if (first) {
        s.clear_rows();
        rq = s.start();
    } else {
        s.load(sc_in, rho_in, fv, pv);
        if (s.phase == S::PH_DONE) {
            if (group == 0) s.save(sc, rho, fv, pv); // (the other copy must say so too)
            if (threadIdx.x == 0 && group == 0) request[prob] = 0;
            return;
        }
        double ggv = s.gg;
        if constexpr (AUG) {
            const bool have_f = (s.pending & FL_REQ_F) && f_dev, have_g = (s.pending & FL_REQ_G) && g_dev;
            s.take_external_aug(have_f ? f_dev[prob] : 0.0, have_f, have_g, g_dev + (size_t)prob * n, c_dev + (size_t)prob * A.aug_m,
                                cd_dev + (size_t)prob * A.aug_m * n, fv, pv, ggv);
        } else {
            if ((s.pending & FL_REQ_F) && f_dev) fv = f_dev[prob];
            if ((s.pending & FL_REQ_G) && g_dev) s.take_gradient(g_dev + (size_t)prob * n, pv, ggv);
        }
        rq = s.advance(fv, pv, ggv);
    }
    if (rq == 0) s.finish();
    else if (!(rq & (FL_REQ_SAME | FL_REQ_NOMOVE))) s.move(s.request_point());
    s.save(sc, rho, fv, pv);
    if (threadIdx.x == 0 && group == 0) {
        int out = rq;
        if (AUG && rq != 0) out |= FL_REQ_C | ((rq & FL_REQ_G) ? FL_REQ_CD : 0);
        request[prob] = out;
    }
}

// ------------------------------------------------------------------ the line searchers on their own
// Wolfe / StrongWolfe (+ _fdwithf) are public procedures of the reference module (NO.f90:1286, 1373, 1462, 1582):
// one search along a given p.  Same machine (fl_linesearch.hpp), parked in HBM between evaluations; the trial
// point x = x0 + a p and phi'(a) = g.p are formed here in the geometry fl_reduction_geometry(n) reports.
template <int NW>
__global__ __launch_bounds__(NW * 64) void line_search_step_kernel(int n, int nslot, int first, int strong, int fused,
                                                                   double c1, double c2, double incr, double a0,
                                                                   double fx0, double phid0, const double *x0,
                                                                   const double *p, double *x, const double *g,
                                                                   const double *f_in, double *sc, int32_t *request)
{
    constexpr int T = NW * 64;
    __shared__ double slots[2 * Reducer<NW>::NVMAX * NW];
    Reducer<NW> R{slots, 0};
    LineSearch ls;
    const int tid = threadIdx.x;
    int rq;
    double fv = 0.0, pv = 0.0;
    if (first) {
        rq = ls.begin(strong, fused, c1, c2, incr, a0, fx0, phid0);
    } else {
        const double *q = sc;
        ls.c1 = *q++; ls.c2abs = *q++; ls.incr = *q++; ls.fx0 = *q++; ls.phid0 = *q++;
        ls.a = *q++; ls.aold = *q++; ls.fx = *q++; ls.fold = *q++; ls.phidnew = *q++; ls.phidold = *q++;
        ls.low = *q++; ls.up = *q++; ls.flow = *q++; ls.fup = *q++; ls.phidlow = *q++; ls.phidup = *q++;
        ls.plma = *q++; ls.a_eval = *q++; fv = *q++; pv = *q++;
        const int *iq = reinterpret_cast<const int *>(sc + 24);
        ls.st = iq[0]; ls.zret = iq[1]; ls.fused = iq[2];
        const int pending = iq[3];
        __syncthreads(); // all waves have read the parked state before thread 0 rewrites it
        if (pending & FL_REQ_F) fv = *f_in;
        if (pending & FL_REQ_G) {
            double r[1] = {0.0};
            for (int c = 0; c < nslot; ++c) {
                const int e = (c * T + tid) << 1;
                const double ta = (e < n) ? g[e] * p[e] : 0.0, tb = (e + 1 < n) ? g[e + 1] * p[e + 1] : 0.0;
                r[0] = (c == 0) ? ta : r[0] + ta;
                r[0] = r[0] + tb;
            }
            R.run(r);
            pv = r[0];
        }
        rq = ls.step(fv, pv);
    }
    if (rq != 0 && !(rq & FL_REQ_SAME)) {
        const double at = ls.a_eval;
        for (int c = 0; c < nslot; ++c) {
            const int e = (c * T + tid) << 1;
            if (e < n) x[e] = x0[e] + at * p[e];
            if (e + 1 < n) x[e + 1] = x0[e + 1] + at * p[e + 1];
        }
    }
    if (tid == 0) {
        double *q = sc;
        *q++ = ls.c1; *q++ = ls.c2abs; *q++ = ls.incr; *q++ = ls.fx0; *q++ = ls.phid0;
        *q++ = ls.a; *q++ = ls.aold; *q++ = ls.fx; *q++ = ls.fold; *q++ = ls.phidnew; *q++ = ls.phidold;
        *q++ = ls.low; *q++ = ls.up; *q++ = ls.flow; *q++ = ls.fup; *q++ = ls.phidlow; *q++ = ls.phidup;
        *q++ = ls.plma; *q++ = ls.a_eval; *q++ = fv; *q++ = pv;
        int *iq = reinterpret_cast<int *>(sc + 24);
        iq[0] = ls.st; iq[1] = ls.zret; iq[2] = ls.fused; iq[3] = rq;
        *request = rq;
    }
}

// H_src [batch][n][n] dense column-major -> the handle's padded buffer (problem k at dst + k * stride, ld) for the problems
// whose request carries FL_REQ_H (all of them when request == nullptr)
__global__ __launch_bounds__(256) void put_hessians_kernel(int n, const double *H_src, double *dst, size_t stride, int ld,
                                                           const int32_t *request)
{
    const int k = blockIdx.y, col = blockIdx.x;
    if (request && !(request[k] & FL_REQ_H)) return;
    const double *s = H_src + ((size_t)k * n + col) * n;
    double *d = dst + (size_t)k * stride + (size_t)col * ld;
    for (int r = threadIdx.x; r < n; r += 256) d[r] = s[r];
}
// the augmented Lagrangian's penalty parameter per problem, from the parked machines
__global__ void get_miu_kernel(int batch, const double *sc_all, int scalars, double miu_first, int first, double *miu)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < batch) miu[k] = first ? miu_first : sc_all[(size_t)k * scalars + 40];
}
// Central differences with MKL djacobi's step rule (fl_host.hpp: central_difference_jacobian), for a batch:
//   fd_points:  xp = x with coordinate j at x_j (1 + eps) | x_j + eps,  xm likewise with (1 - eps) | - eps
//   fd_column:  H(:, j) of every problem = (gp - gm) * (0.5 / h_j),  h_j = eps x_j | eps  (H [batch][n][n] column-major)
__global__ __launch_bounds__(256) void fd_points_kernel(int batch, int n, int j, double eps, const double *x, double *xp, double *xm)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)batch * n) return;
    const int i = (int)(e % n);
    const double v = x[e];
    double a = v, b = v;
    if (i == j) {
        if (fabs(v) > eps) {
            a = v * (1.0 + eps);
            b = v * (1.0 - eps);
        } else {
            a = v + eps;
            b = v - eps;
        }
    }
    xp[e] = a;
    xm[e] = b;
}
__global__ __launch_bounds__(256) void fd_column_kernel(int batch, int n, int j, double eps, const double *x, const double *gp,
                                                        const double *gm, double *H)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)batch * n) return;
    const size_t k = e / n;
    const int i = (int)(e % n);
    const double xj = x[k * n + j];
    const double h = fabs(xj) > eps ? eps * xj : eps;
    H[(k * n + j) * n + i] = (gp[e] - gm[e]) * (0.5 / h);
}

struct Rci {
    int solver, batch, n, nw, ept, first;
    int aug; // augmented Lagrangian around the inner solver: c, cd come with the evaluations
    int32_t *outer;
    double *cnorm2;
    SolveArgs A;
    double *sc, *vec, *rho, *ws, *f_out, *gg_out;
    int32_t *iters, *status, *nf, *ng;
    hipStream_t stream;
    int coop_groups;        // vectors-in-HBM path: workgroups per problem (BigSolver's cooperative form), 1 = none
    int parity;             // ... which copy of the parked scalars the next step reads
    double *coop_part;      // [batch][2][groups][NVMAX]
    unsigned *coop_counter; // [batch][2]: arrivals, "gave up" flag
    unsigned *coop_flag_host, *coop_flag_dev; // one pinned word the kernels set when a cooperative barrier gave up (+ its device address)
};

template <int NW, int EPT>
static void launch_rci(Rci *h, const double *f, const double *g, const double *c, const double *cd, int32_t *req, int flags,
                       const int32_t *active, int nactive, double *xio)
{
    dim3 grid(active ? nactive : h->batch), block(NW * 64);
#define FL_RCI(M)                                                                                                     \
    hipLaunchKernelGGL((rci_step_kernel<NW, EPT, M, 0>), grid, block, 0, h->stream, h->A, h->first, flags, active, xio, h->sc, \
                       h->vec, h->rho, f, g, c, cd, req)
#define FL_RCI_AUG(M)                                                                                                 \
    hipLaunchKernelGGL((rci_step_kernel<NW, EPT, M, 1>), grid, block, 0, h->stream, h->A, h->first, flags, active, xio, h->sc, \
                       h->vec, h->rho, f, g, c, cd, req)
    if (h->aug) { // AugmentedLagrangian around L-BFGS (NO.f90:2150-2167), ConjugateGradient (2168-2185) or quasi-Newton
                  // BFGS (2131-2148 with ExactStep <= 0: every outer round rebuilds H from a I)
        if (h->solver == FL_SOLVER_CG) FL_RCI_AUG(FL_SOLVER_CG);
        else if (h->solver == FL_SOLVER_BFGS) FL_RCI_AUG(FL_SOLVER_BFGS);
        else if (h->solver == FL_SOLVER_NEWTON) {
            // NewtonRaphson around the caller's Hessian of L (NO.f90:2074-2130)
            FL_RCI_AUG(FL_SOLVER_NEWTON);
        } else FL_RCI_AUG(FL_SOLVER_LBFGS);
        return;
    }
    switch (h->solver) {
    case FL_SOLVER_SD: FL_RCI(FL_SOLVER_SD); break;
    case FL_SOLVER_CG: FL_RCI(FL_SOLVER_CG); break;
    case FL_SOLVER_BFGS: FL_RCI(FL_SOLVER_BFGS); break;
    case FL_SOLVER_NEWTON: FL_RCI(FL_SOLVER_NEWTON); break;
    default: FL_RCI(FL_SOLVER_LBFGS); break;
    }
#undef FL_RCI
#undef FL_RCI_AUG
}

static void launch_rci_big(Rci *h, const double *f, const double *g, int32_t *req, const double *c = nullptr, const double *cd = nullptr)
{
    const int G = h->coop_groups > 1 ? h->coop_groups : 1;
    if (G > 1) // this launch's barriers count from 0 (word 0 of each problem's pair; word 1 is the "gave up waiting" flag: kept)
        (void)hipMemset2DAsync(h->coop_counter, 2 * sizeof(unsigned), 0, sizeof(unsigned), h->batch, h->stream);
    dim3 grid(h->batch * G), block(1024);
#define FL_RCI(M)                                                                                                 \
    hipLaunchKernelGGL((rci_step_big_kernel<M>), grid, block, 0, h->stream, h->A, h->first, h->sc, h->vec, h->rho, f, \
                       g, req, G, h->coop_part, h->coop_counter, h->parity, h->coop_flag_dev)
#define FL_RCI_AUG(M)                                                                                                  \
    hipLaunchKernelGGL((rci_step_big_kernel<M, 1>), grid, block, 0, h->stream, h->A, h->first, h->sc, h->vec, h->rho, f, \
                       g, req, 1, h->coop_part, h->coop_counter, h->parity, h->coop_flag_dev, c, cd)
    if (h->aug) { // AugmentedLagrangian around L-BFGS, ConjugateGradient or quasi-Newton BFGS (NO.f90:2131-2185)
        switch (h->solver) {
        case FL_SOLVER_CG: FL_RCI_AUG(FL_SOLVER_CG); break;
        case FL_SOLVER_BFGS: FL_RCI_AUG(FL_SOLVER_BFGS); break;
        default: FL_RCI_AUG(FL_SOLVER_LBFGS); break;
        }
        return;
    }
    switch (h->solver) {
    case FL_SOLVER_SD: FL_RCI(FL_SOLVER_SD); break;
    case FL_SOLVER_CG: FL_RCI(FL_SOLVER_CG); break;
    case FL_SOLVER_BFGS: FL_RCI(FL_SOLVER_BFGS); break;
    default: FL_RCI(FL_SOLVER_LBFGS); break;
    }
#undef FL_RCI
#undef FL_RCI_AUG
    if (G > 1) h->parity ^= 1;
}

} // namespace fl

extern "C" {

struct fl_rci {
    fl::Rci r;
};

int fl_rci_destroy(fl_rci *h)
{
    if (!h) return FL_OK;
    if (h->r.coop_part) (void)hipFree(h->r.coop_part);
    if (h->r.coop_counter) (void)hipFree(h->r.coop_counter);
    if (h->r.coop_flag_host) (void)hipHostFree(h->r.coop_flag_host);
    void *bufs[] = {h->r.sc, h->r.vec, h->r.rho, h->r.ws, h->r.f_out, h->r.gg_out, h->r.iters, h->r.status,
                    h->r.nf, h->r.ng, h->r.outer, h->r.cnorm2};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete h;
    return FL_OK;
}

int fl_rci_create(fl_rci **out, int solver, int batch, int n, const fl_options *opt, void *stream)
{
    if (!out || !opt || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (solver < FL_SOLVER_SD || solver > FL_SOLVER_NEWTON) return FL_ERR_INVALID_ARGUMENT;
    if (opt->cg_method != FL_CG_DY && opt->cg_method != FL_CG_PR) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(n, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    // beyond the register path (1024 threads): SD / CG / L-BFGS, and BFGS with quasi-Newton updates only up to
    // n = 16384; NewtonRaphson and the exact-Hessian refresh (dense Cholesky) need n <= 4096
    if (threads == 1024 && (solver == FL_SOLVER_NEWTON ||
                            (solver == FL_SOLVER_BFGS && (opt->exact_step > 0 ||
                                                          n > fl::BigSolver<FL_OBJ_EXTERNAL, FL_SOLVER_BFGS>::BF_MAX_N))))
        return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    fl_rci *h = new (std::nothrow) fl_rci();
    if (!h) return FL_ERR_WORKSPACE;
    std::memset(&h->r, 0, sizeof h->r);
    fl::Rci &r = h->r;
    r.solver = solver;
    r.batch = batch;
    r.n = n;
    r.nw = threads / 64;
    r.ept = ept;
    r.first = 1;
    r.stream = static_cast<hipStream_t>(stream);
    fl::SolveArgs &A = r.A;
    A.n = n;
    A.batch = batch;
    A.mem = opt->memory > 1 ? opt->memory : 1;
    if (solver == FL_SOLVER_LBFGS && A.mem > FL_MAX_MEMORY) {
        delete h;
        return FL_ERR_UNSUPPORTED_SIZE;
    }
    A.maxit = opt->max_iteration;
    A.strong = opt->strong != 0;
    A.fused = opt->fused_f_fd != 0;
    A.cg_method = opt->cg_method;
    A.tol = opt->precision * opt->precision;
    A.minstep = opt->min_step_length * opt->min_step_length;
    A.c1 = opt->wolfe_c1;
    A.c2 = opt->wolfe_c2;
    if (opt->clamp) {
        A.c1 = opt->wolfe_c1 > 1e-15 ? opt->wolfe_c1 : 1e-15;
        const double lo = A.c1 + 1e-15;
        const double c2 = opt->wolfe_c2 > lo ? opt->wolfe_c2 : lo;
        A.c2 = c2 < 1.0 - 1e-15 ? c2 : 1.0 - 1e-15;
    }
    A.incr = opt->increment;
    A.exact_step = (solver == FL_SOLVER_BFGS) ? opt->exact_step : 0; // > 0: the caller answers FL_REQ_H requests
    A.miu0 = 1.0;
    A.precision = opt->precision;
    const size_t npad = (size_t)threads * ept, B = (size_t)batch;
    const size_t wsb = fl_workspace_bytes_for(solver, batch, n, opt);
    bool ok = hipMalloc((void **)&r.sc, 2 * B * 48 * sizeof(double)) == hipSuccess && // (two copies: the cooperative form alternates)
              hipMalloc((void **)&r.vec, B * 4 * npad * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&r.rho, 2 * B * FL_MAX_MEMORY * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&r.f_out, B * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&r.gg_out, B * sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&r.iters, B * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void **)&r.status, B * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void **)&r.nf, B * sizeof(int32_t)) == hipSuccess &&
              hipMalloc((void **)&r.ng, B * sizeof(int32_t)) == hipSuccess;
    if (ok && wsb) ok = hipMalloc((void **)&r.ws, wsb) == hipSuccess;
    r.coop_groups = 1;
    if (ok && threads == 1024 && solver != FL_SOLVER_BFGS) {
        // Few problems of very large n: several workgroups per problem, as many as keep every workgroup RESIDENT (their
        // barriers spin: one 1024-thread workgroup per CU is assumed, 256 CUs) and give each at least two slots.
        // FL_COOP_GROUPS in the environment overrides (tests; 1 switches the cooperative form off).
        using BS = fl::BigSolver<FL_OBJ_EXTERNAL, FL_SOLVER_LBFGS>;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (cus > BS::COOP_MAX_GROUPS) cus = BS::COOP_MAX_GROUPS;
        // the barriers of the cooperative form SPIN: all batch x G workgroups must be resident together.  The grid is sized
        // from what the runtime says fits (workgroups of this very kernel per CU x CUs), not from an assumption
        int per_cu = 0;
        {
            hipError_t eo = hipErrorUnknown;
            switch (solver) {
            case FL_SOLVER_SD: eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fl::rci_step_big_kernel<FL_SOLVER_SD>, 1024, 0); break;
            case FL_SOLVER_CG: eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fl::rci_step_big_kernel<FL_SOLVER_CG>, 1024, 0); break;
            default: eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fl::rci_step_big_kernel<FL_SOLVER_LBFGS>, 1024, 0); break;
            }
            if (eo != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                per_cu = 0; // (no answer: no cooperative form)
            }
            if (per_cu > 1) per_cu = 1; // one workgroup per CU: a sibling on the same CU would share its LDS pipe for nothing
        }
        cus *= per_cu;
        int want = cus / batch;
        const int nslot = BS::slots_for(n);
        if (want > nslot / 2) want = nslot / 2;
        if (const char *e = std::getenv("FL_COOP_GROUPS")) want = std::atoi(e);
        if (want > cus / batch) want = cus / batch;
        if (want > 1) {
            r.coop_groups = BS::coop_groups(n, want);
            ok = hipMalloc((void **)&r.coop_part, B * 2 * r.coop_groups * fl::Reducer<16>::NVMAX * sizeof(double)) == hipSuccess &&
                 hipMalloc((void **)&r.coop_counter, 2 * B * sizeof(unsigned)) == hipSuccess &&
                 hipMemsetAsync(r.coop_counter, 0, 2 * B * sizeof(unsigned), r.stream) == hipSuccess &&
                 hipHostMalloc((void **)&r.coop_flag_host, sizeof(unsigned), hipHostMallocMapped) == hipSuccess;
            if (ok) {
                *r.coop_flag_host = 0u;
                ok = hipHostGetDevicePointer((void **)&r.coop_flag_dev, r.coop_flag_host, 0) == hipSuccess;
            }
        }
    }
    if (!ok) {
        fl_rci_destroy(h);
        return FL_ERR_WORKSPACE;
    }
    A.hist = r.ws;
    A.f_out = r.f_out;
    A.gg_out = r.gg_out;
    A.iters = r.iters;
    A.status = r.status;
    A.nf = r.nf;
    A.ng = r.ng;
    *out = h;
    return FL_OK;
}

static int rci_step_any(fl_rci *h, double *x_dev, const double *f_dev, const double *g_dev, const double *c_dev,
                        const double *cd_dev, int32_t *request_dev, int flags = 0, const int32_t *active_dev = nullptr,
                        int nactive = 0, double *xc_dev = nullptr)
{
    if (!h || !x_dev || !request_dev) return FL_ERR_INVALID_ARGUMENT;
    if (h->r.aug && !h->r.first && (!c_dev || !cd_dev)) return FL_ERR_INVALID_ARGUMENT;
    fl::Rci *r = &h->r;
    // f_dev / g_dev may be NULL only where no problem can have asked for them: on the first step, and with a solver that
    // takes Hessians from the caller (a step that only delivers them, FL_REQ_H); otherwise a NULL array is a caller's bug,
    // not something to step the machines past
    const bool hessians = r->solver == FL_SOLVER_NEWTON || (r->solver == FL_SOLVER_BFGS && r->A.exact_step > 0);
    if (!r->first && !hessians && (!f_dev || !g_dev)) return FL_ERR_INVALID_ARGUMENT;
    r->A.x = x_dev;
    const int nw = r->nw, ept = r->ept;
    double *xio = xc_dev ? xc_dev : x_dev;
    if (nw == 16) {
        if (active_dev || flags) return FL_ERR_UNSUPPORTED_SIZE; // (the vectors-in-HBM path steps whole batches)
        // a cooperative barrier of an earlier step gave up (its siblings were not resident): nothing since can be trusted
        if (r->coop_flag_host && *(volatile unsigned *)r->coop_flag_host != 0u) return FL_ERR_LAUNCH;
        fl::launch_rci_big(r, f_dev, g_dev, request_dev, c_dev, cd_dev);
    }
    else if (nw == 1 && ept == 2) fl::launch_rci<1, 2>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    else if (nw == 1 && ept == 4) fl::launch_rci<1, 4>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    else if (nw == 1 && ept == 8) fl::launch_rci<1, 8>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    else if (nw == 2 && ept == 8) fl::launch_rci<2, 8>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    else if (nw == 4 && ept == 8) fl::launch_rci<4, 8>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    else fl::launch_rci<8, 8>(r, f_dev, g_dev, c_dev, cd_dev, request_dev, flags, active_dev, nactive, xio);
    r->first = 0;
    return fl::launch_status();
}

// The step for a caller that compacts: only the n_active problems listed in active_dev are stepped, and the arrays of
// the step -- xc_dev [n_active][n] (in: the points just evaluated; out: the next requested points), f_dev, g_dev,
// request_dev -- are indexed by the position in that list.  x_dev [batch][n] receives a problem's minimiser when it
// finishes.  Between two calls the caller may drop finished problems from the list, moving the rows of xc_dev along.
int fl_rci_step_compact(fl_rci *h, double *x_dev, const int32_t *active_dev, int n_active, double *xc_dev, const double *f_dev,
                        const double *g_dev, int32_t *request_dev, int flags)
{
    if (!h || !active_dev || !xc_dev || n_active <= 0 || n_active > h->r.batch) return FL_ERR_INVALID_ARGUMENT;
    if (h->r.aug || (h->r.solver != FL_SOLVER_SD && h->r.solver != FL_SOLVER_CG && h->r.solver != FL_SOLVER_LBFGS))
        return FL_ERR_INVALID_ARGUMENT; // (the dense solvers park their iterate in x_dev[prob]: whole-batch steps only)
    if (flags & ~FL_RCI_BOTH) return FL_ERR_INVALID_ARGUMENT;
    return rci_step_any(h, x_dev, f_dev, g_dev, nullptr, nullptr, request_dev, flags, active_dev, n_active, xc_dev);
}

int fl_rci_step_flags(fl_rci *h, double *x_dev, const double *f_dev, const double *g_dev, int32_t *request_dev, int flags)
{
    if (h && h->r.aug) return FL_ERR_INVALID_ARGUMENT;
    if (flags & ~FL_RCI_BOTH) return FL_ERR_INVALID_ARGUMENT;
    if (flags && h && h->r.solver != FL_SOLVER_SD && h->r.solver != FL_SOLVER_CG && h->r.solver != FL_SOLVER_LBFGS)
        return FL_ERR_INVALID_ARGUMENT; // FL_RCI_BOTH: the vector solvers (the dense ones step request by request)
    return rci_step_any(h, x_dev, f_dev, g_dev, nullptr, nullptr, request_dev, flags);
}

int fl_rci_step(fl_rci *h, double *x_dev, const double *f_dev, const double *g_dev, int32_t *request_dev)
{
    if (h && h->r.aug) return FL_ERR_INVALID_ARGUMENT; // an augmented-Lagrangian handle steps with fl_rci_step_auglag
    return rci_step_any(h, x_dev, f_dev, g_dev, nullptr, nullptr, request_dev);
}

// AugmentedLagrangian (NO.f90:2005-2241) for a batch with the CALLER's objective and constraints: create a handle
// around the inner solver (FL_SOLVER_LBFGS | FL_SOLVER_CG | FL_SOLVER_BFGS with exact_step <= 0), then step it like fl_rci_step with, in addition,
// c_dev [batch][m] = c(x) and cd_dev [batch][m][n] = the constraint Jacobian (row j = grad c_j; Fortran cdx(N,M)).
int fl_rci_create_auglag(fl_rci **out, int solver, int batch, int n, int m, double *lambda_dev, double miu0,
                         const fl_options *opt, void *stream)
{
    if (solver != FL_SOLVER_LBFGS && solver != FL_SOLVER_CG && solver != FL_SOLVER_BFGS && solver != FL_SOLVER_NEWTON)
        return FL_ERR_INVALID_ARGUMENT;
    if (!opt || m < 1 || m > FL_MAX_CONSTRAINTS || !lambda_dev) return FL_ERR_INVALID_ARGUMENT;
    // NewtonRaphson, and BFGS with exact_step > 0, ask for the Hessian of L (FL_REQ_H: NO.f90:2229-2241, Ldd): the register path
    // only (n <= 4096).  L-BFGS, CG and quasi-Newton BFGS go on beyond it with the vectors in HBM (round 4; BFGS to n = 16384)
    if (n > 4096 && (solver == FL_SOLVER_NEWTON || (solver == FL_SOLVER_BFGS && opt->exact_step > 0))) return FL_ERR_UNSUPPORTED_SIZE;
    const int rc = fl_rci_create(out, solver, batch, n, opt, stream);
    if (rc != FL_OK) return rc;
    fl::Rci &r = (*out)->r;
    r.coop_groups = 1; // (one copy of lambda per problem: one workgroup per problem)
    const size_t B = (size_t)batch;
    if (hipMalloc((void **)&r.outer, B * sizeof(int32_t)) != hipSuccess ||
        hipMalloc((void **)&r.cnorm2, B * sizeof(double)) != hipSuccess) {
        fl_rci_destroy(*out);
        *out = nullptr;
        return FL_ERR_WORKSPACE;
    }
    r.aug = 1;
    r.A.aug_m = m;
    r.A.miu0 = miu0;
    r.A.lambda = lambda_dev; // in: lambda0, out: the multipliers (kept up to date from step to step)
    r.A.outer = r.outer;
    r.A.cnorm2 = r.cnorm2;
    return FL_OK;
}

int fl_rci_step_auglag(fl_rci *h, double *x_dev, const double *f_dev, const double *g_dev, const double *c_dev,
                       const double *cd_dev, int32_t *request_dev)
{
    if (!h || !h->r.aug) return FL_ERR_INVALID_ARGUMENT;
    return rci_step_any(h, x_dev, f_dev, g_dev, c_dev, cd_dev, request_dev);
}

// after all requests are 0: c.c at exit and the number of outer iterations per problem (each may be NULL); the other
// outputs through fl_rci_results (f = the augmented Lagrangian at exit, iters = inner iterations of all outer rounds)
int fl_rci_results_auglag(fl_rci *h, double *cnorm2_dev, int32_t *outer_dev)
{
    if (!h || !h->r.aug) return FL_ERR_INVALID_ARGUMENT;
    const size_t B = (size_t)h->r.batch;
    bool ok = true;
    if (cnorm2_dev) ok &= hipMemcpyAsync(cnorm2_dev, h->r.cnorm2, B * 8, hipMemcpyDeviceToDevice, h->r.stream) == hipSuccess;
    if (outer_dev) ok &= hipMemcpyAsync(outer_dev, h->r.outer, B * 4, hipMemcpyDeviceToDevice, h->r.stream) == hipSuccess;
    return ok ? FL_OK : FL_ERR_LAUNCH;
}

// workgroups that share one problem in this handle's step kernel (> 1: the cooperative form of the vectors-in-HBM path;
// the summation order -- hence the bits -- depends on it: oracle flo_set_sum_groups)
int fl_rci_cooperative_groups(fl_rci *h) { return h ? (h->r.coop_groups > 1 ? h->r.coop_groups : 1) : 0; }

int fl_rci_hessian_buffer(fl_rci *h, double **hessian_dev, int *ld)
{
    if (!h || !hessian_dev || !ld) return FL_ERR_INVALID_ARGUMENT;
    const size_t npad = (size_t)h->r.nw * 64 * h->r.ept, mat = (size_t)h->r.n * npad;
    if (h->r.solver == FL_SOLVER_NEWTON) *hessian_dev = h->r.ws;                                  // [batch][n][ld]
    else if (h->r.solver == FL_SOLVER_BFGS && h->r.A.exact_step > 0) *hessian_dev = h->r.ws + mat; // U of (H,U,W)
    else return FL_ERR_INVALID_ARGUMENT;
    *ld = (int)npad;
    return FL_OK;
}

// H_dev [batch][n][n] (dense, column-major -- a Hessian is symmetric, so row-major is the same) -> the handle's buffer, for
// the problems whose request asks for it (request_dev == NULL: all)
int fl_rci_put_hessians(fl_rci *h, const double *H_dev, const int32_t *request_dev)
{
    double *dst = nullptr;
    int ld = 0;
    if (!h || !H_dev) return FL_ERR_INVALID_ARGUMENT;
    const int rc = fl_rci_hessian_buffer(h, &dst, &ld);
    if (rc != FL_OK) return rc;
    const size_t mat = (size_t)h->r.n * ld, stride = (h->r.solver == FL_SOLVER_NEWTON) ? mat : 3 * mat;
    for (int b0 = 0; b0 < h->r.batch; b0 += FL_GRID_YZ_MAX) {
        const int nb = h->r.batch - b0 < FL_GRID_YZ_MAX ? h->r.batch - b0 : FL_GRID_YZ_MAX;
        hipLaunchKernelGGL(fl::put_hessians_kernel, dim3(h->r.n, nb), dim3(256), 0, h->r.stream, h->r.n,
                           H_dev + (size_t)b0 * h->r.n * h->r.n, dst + (size_t)b0 * stride, stride, ld,
                           request_dev ? request_dev + b0 : nullptr);
    }
    return fl::launch_status();
}

// miu_dev [batch] <- the penalty parameter of every problem's current outer round (an augmented-Lagrangian handle): with
// lambda_dev (kept up to date by the steps) what a caller needs to form the Hessian of L as the reference's Ldd does,
//   Ldd = f'' + sum_j c_j'' (miu c_j - lambda_j) + cd cd^T                                   (NO.f90:2229-2241)
int fl_rci_auglag_miu(fl_rci *h, double *miu_dev)
{
    if (!h || !h->r.aug || !miu_dev) return FL_ERR_INVALID_ARGUMENT;
    const double m0 = h->r.A.miu0 > 1.0 ? h->r.A.miu0 : 1.0;
    hipLaunchKernelGGL(fl::get_miu_kernel, dim3((h->r.batch + 255) / 256), dim3(256), 0, h->r.stream, h->r.batch, h->r.sc, 48, m0,
                       h->r.first, miu_dev);
    return fl::launch_status();
}

// Central differences of a gradient for a batch, with MKL djacobi's step rule (what the reference does for f'' when no
// fdd is passed, NO.f90:676, 981, 1067): for j = 0 .. n-1:  fl_fd_points(j) -> evaluate the gradients gp at xp and gm
// at xm -> fl_fd_column(j) writes column j of every problem's H [batch][n][n].  2n gradient evaluations of the batch.
int fl_fd_points(int batch, int n, int j, double eps, const double *x_dev, double *xp_dev, double *xm_dev, void *stream)
{
    if (batch <= 0 || n <= 0 || j < 0 || j >= n || !(eps > 0.0) || !x_dev || !xp_dev || !xm_dev) return FL_ERR_INVALID_ARGUMENT;
    const size_t tot = (size_t)batch * n;
    hipLaunchKernelGGL(fl::fd_points_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), batch, n,
                       j, eps, x_dev, xp_dev, xm_dev);
    return fl::launch_status();
}
int fl_fd_column(int batch, int n, int j, double eps, const double *x_dev, const double *gp_dev, const double *gm_dev, double *H_dev,
                 void *stream)
{
    if (batch <= 0 || n <= 0 || j < 0 || j >= n || !(eps > 0.0) || !x_dev || !gp_dev || !gm_dev || !H_dev) return FL_ERR_INVALID_ARGUMENT;
    const size_t tot = (size_t)batch * n;
    hipLaunchKernelGGL(fl::fd_column_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), batch, n,
                       j, eps, x_dev, gp_dev, gm_dev, H_dev);
    return fl::launch_status();
}

int fl_rci_results(fl_rci *h, double *f_dev, double *gg_dev, int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev,
                   int32_t *ng_dev)
{
    if (!h) return FL_ERR_INVALID_ARGUMENT;
    const size_t B = (size_t)h->r.batch;
    hipStream_t st = h->r.stream;
    bool ok = true;
    if (h->r.coop_groups > 1) { // did a cooperative step ever give up waiting for its siblings?  (then nothing can be trusted)
        std::vector<unsigned> words(2 * B);
        if (hipMemcpyAsync(words.data(), h->r.coop_counter, 2 * B * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return FL_ERR_LAUNCH;
        for (size_t k = 0; k < B; ++k)
            if (words[2 * k + 1] != 0) return FL_ERR_LAUNCH;
    }
    if (f_dev) ok &= hipMemcpyAsync(f_dev, h->r.f_out, B * 8, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (gg_dev) ok &= hipMemcpyAsync(gg_dev, h->r.gg_out, B * 8, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (iters_dev) ok &= hipMemcpyAsync(iters_dev, h->r.iters, B * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (status_dev) ok &= hipMemcpyAsync(status_dev, h->r.status, B * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (nf_dev) ok &= hipMemcpyAsync(nf_dev, h->r.nf, B * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    if (ng_dev) ok &= hipMemcpyAsync(ng_dev, h->r.ng, B * 4, hipMemcpyDeviceToDevice, st) == hipSuccess;
    return ok ? FL_OK : FL_ERR_LAUNCH;
}

// ------------------------------------------------------------------ legacy entry points
typedef void (*f_cb)(double &, const double *, const int &);
typedef void (*fd_cb)(double *, const double *, const int &);
typedef int (*ffd_cb)(double &, double *, const double *, const int &);
typedef int (*fdd_cb)(double *, const double *, const int &);

// One problem, host callbacks: the machine steps on the GPU, f / f' are evaluated by the caller's
// code on the host exactly when the reference would call them (request bits).  Returns the status.
struct HostObjective { // the caller's code, evaluated on the host when the machine asks for it
    std::function<void(double *, const double *, int)> fdd; // Hessian, column-major n x n (empty = absent)
    std::function<void(double &, const double *, int)> f;
    std::function<void(double *, const double *, int)> fd;
    std::function<void(double &, double *, const double *, int)> f_fd; // empty = f_fd absent
};
static HostObjective host_objective(f_cb f, fd_cb fd, ffd_cb f_fd)
{
    HostObjective ob;
    ob.f = [f](double &fx, const double *x, int n) { f(fx, x, n); };
    ob.fd = [fd](double *g, const double *x, int n) { fd(g, x, n); };
    if (f_fd) ob.f_fd = [f_fd](double &fx, double *g, const double *x, int n) { (void)f_fd(fx, g, x, n); };
    return ob;
}

// fdd absent: the reference differentiates f' numerically with MKL's djacobi(fd_j,dim,dim,H,x,1d-8) (NO.f90:676,
// 981, 1067, 1258).  Same central differences with djacobi's own step rule (fl_host.hpp: central_difference_jacobian,
// pinned bit for bit to the real MKL routine by tests/golden/mkl_djacobi.npz), 2n gradient calls per Hessian.
static std::function<void(double *, const double *, int)>
central_difference_hessian(std::function<void(double *, const double *, int)> grad)
{
    return [grad](double *H, const double *x, int n) {
        std::vector<double> xp(x, x + n), gp(n), gm(n);
        fl::central_difference_jacobian([&](const double *xx, double *g) { grad(g, xx, n); }, n, n, H, xp.data(), 1e-8, gp.data(),
                                    gm.data());
    };
}
static std::function<void(double *, const double *, int)> central_difference_hessian(fd_cb fd)
{
    return central_difference_hessian(
        std::function<void(double *, const double *, int)>([fd](double *g, const double *x, int n) { fd(g, x, n); }));
}

static int legacy_solve(int solver, const char *name, const HostObjective &ob, double *x, int n, const fl_options &o,
                        int warn)
{
    fl_rci *h = nullptr;
    int rc = fl_rci_create(&h, solver, 1, n, &o, nullptr);
    if (rc != FL_OK) {
        std::fprintf(stderr, "FortranLibrary(MI355X) %s: cannot run on the device (error %d); x is unchanged\n", name, rc);
        return rc;
    }
    // The mailbox between the step kernel and the host callbacks: x, f'(x), f and the request live in ONE block of
    // pinned host memory that is mapped into the device's address space.  The kernel reads the evaluations from it and
    // writes the next trial point and request into it directly; the host only synchronises the stream.  (Round 1 moved
    // them with three or four pageable hipMemcpy per trial -- most of the time of a one-problem solve.)
    double *mail = nullptr, *mail_dev = nullptr;
    std::vector<double> hess;
    int32_t status = FL_STATUS_MAXIT;
    const size_t mail_doubles = 2 * (size_t)n + 2;
    bool ok = hipHostMalloc((void **)&mail, sizeof(double) * mail_doubles, hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void **)&mail_dev, mail, 0) == hipSuccess;
    double *xm = mail, *gm = mail + n, *fm = mail + 2 * n;           // host views
    volatile int32_t *rqm = reinterpret_cast<volatile int32_t *>(mail + 2 * n + 1);
    double *xd = mail_dev, *gd = mail_dev + n, *fdv = mail_dev + 2 * n; // the same block as the kernel sees it
    int32_t *rqd = reinterpret_cast<int32_t *>(mail_dev + 2 * n + 1);
    if (ok) {
        std::copy(x, x + n, xm);
        *fm = 0.0;
        *rqm = 0;
    }
    while (ok) {
        if (fl_rci_step(h, xd, fdv, gd, rqd) != FL_OK || hipStreamSynchronize(nullptr) != hipSuccess) { ok = false; break; }
        const int32_t rq = *rqm;
        if (rq == 0) break;
        if (rq & FL_REQ_H) { // info=fdd(H,x,dim): evaluate on the host, copy into the handle's padded buffer
            double *Hd = nullptr;
            int ld = 0;
            if (!ob.fdd || fl_rci_hessian_buffer(h, &Hd, &ld) != FL_OK) { ok = false; break; }
            if (hess.empty()) hess.resize((size_t)n * n);
            ob.fdd(hess.data(), xm, n);
            if (hipMemcpy2D(Hd, sizeof(double) * ld, hess.data(), sizeof(double) * n, sizeof(double) * n, n,
                            hipMemcpyHostToDevice) != hipSuccess) { ok = false; break; }
            continue;
        }
        const bool wf = rq & FL_REQ_F, wg = rq & FL_REQ_G;
        if (wf && wg && ob.f_fd) {
            ob.f_fd(*fm, gm, xm, n); // the integer return value is ignored like the reference does (NO.f90:437)
        } else {
            if (wf) ob.f(*fm, xm, n);
            if (wg) ob.fd(gm, xm, n);
        }
    }
    double gg = 0.0;
    if (ok) {
        std::copy(xm, xm + n, x); // the minimiser (the kernel's finish() wrote it into the mailbox)
        ok = hipMemcpy(&status, h->r.status, sizeof status, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(&gg, h->r.gg_out, sizeof gg, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) %s: HIP error during the solve\n", name);
    if (ok && warn) { // the reference's warnings (e.g. NO.f90:580-583, 616-619)
        if (status == FL_STATUS_MAXIT) {
            std::printf(" Failed %s: max iteration exceeded!\n Euclidean norm of gradient = %24.16E\n", name, sqrt(gg));
        } else if (status == FL_STATUS_STEP_CONVERGED) {
            std::printf(" %s warning: step length has converged, but gradient norm has not met accuracy goal\n"
                        " Euclidean norm of gradient = %24.16E\n", name, sqrt(gg));
        }
    }
    if (mail) (void)hipHostFree(mail);
    fl_rci_destroy(h);
    return ok ? status : FL_ERR_LAUNCH;
}

// One line search with host callbacks (the public Wolfe / StrongWolfe procedures).  On exit, like the reference:
// a = the accepted step, x = x + a p, fx = f(x), fdx = f'(x).
static void legacy_line_search(int strong, int fused, const double *c1, const double *c2, f_cb f, fd_cb fd, ffd_cb f_fd,
                               double *x, double *a, const double *p, double *fx, const double *phid0, double *fdx,
                               const int *dim, const double *Increment)
{
    const int n = *dim;
    int threads = 0, ept = 0;
    if (n <= 0 || fl_reduction_geometry(n, &threads, &ept) != FL_OK) {
        std::fprintf(stderr, "FortranLibrary(MI355X) line search: unsupported dimension %d; x is unchanged\n", n);
        return;
    }
    const double incr = Increment ? *Increment : 1.05; // fail-safe max(1+1d-15, Increment) inside the machine
    // x0, p and the parked machine on the device; the trial point, f'(x), f and the request in a pinned, device-mapped
    // mailbox (see legacy_solve)
    double *x0d = nullptr, *pd = nullptr, *sc = nullptr, *mail = nullptr, *mail_dev = nullptr;
    const size_t vb = sizeof(double) * (size_t)n;
    bool ok = hipMalloc((void **)&x0d, vb) == hipSuccess && hipMalloc((void **)&pd, vb) == hipSuccess &&
              hipMalloc((void **)&sc, 32 * sizeof(double)) == hipSuccess &&
              hipHostMalloc((void **)&mail, sizeof(double) * (2 * (size_t)n + 2), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void **)&mail_dev, mail, 0) == hipSuccess;
    double *xm = mail, *gm = mail + n, *fm = mail + 2 * n;
    volatile int32_t *rqm = reinterpret_cast<volatile int32_t *>(mail + 2 * n + 1);
    double *xd = mail_dev, *gd = mail_dev + n, *fd_dev = mail_dev + 2 * n;
    int32_t *rqd = reinterpret_cast<int32_t *>(mail_dev + 2 * n + 1);
    ok = ok && hipMemcpy(x0d, x, vb, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(pd, p, vb, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        std::copy(x, x + n, xm);
        std::copy(fdx, fdx + n, gm); // (what the caller's array holds stays if no gradient is ever asked for)
        *fm = *fx;
        *rqm = 0;
    }
    int first = 1;
    const int nw = threads / 64, nslot = ept / 2;
    while (ok) {
#define FL_LS(NW_)                                                                                                \
    hipLaunchKernelGGL((fl::line_search_step_kernel<NW_>), dim3(1), dim3(NW_ * 64), 0, nullptr, n, nslot, first,       \
                       strong, fused, *c1, *c2, incr, *a, *fx, *phid0, x0d, pd, xd, gd, fd_dev, sc, rqd)
        switch (nw) {
        case 1: FL_LS(1); break;
        case 2: FL_LS(2); break;
        case 4: FL_LS(4); break;
        case 8: FL_LS(8); break;
        default: FL_LS(16); break;
        }
#undef FL_LS
        first = 0;
        if (hipStreamSynchronize(nullptr) != hipSuccess) { ok = false; break; }
        const int32_t rq = *rqm;
        if (rq == 0) break;
        const bool wf = rq & FL_REQ_F, wg = rq & FL_REQ_G;
        if (wf && wg && f_fd) {
            (void)f_fd(*fm, gm, xm, n);
        } else {
            if (wf) f(*fm, xm, n);
            if (wg) fd(gm, xm, n);
        }
    }
    if (ok) { // on exit, like the reference: x = x + a p and fdx = f'(x) there (the last evaluation)
        std::copy(xm, xm + n, x);
        std::copy(gm, gm + n, fdx);
    }
    if (ok) {
        double st[8];
        ok = hipMemcpy(st, sc, sizeof st, hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) {
            *a = st[5];  // ls.a
            *fx = st[7]; // ls.fx
        }
    }
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) line search: HIP error\n");
    void *bufs[] = {x0d, pd, sc};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (mail) (void)hipHostFree(mail);
}

// optional dummy arguments arrive as NULL when absent (Fortran callers); C++ always passes all
static void legacy_options(fl_options &o, int solver, const int32_t *Strong, const int *MaxIteration,
                           const double *Precision, const double *MinStepLength, const double *WolfeConst1,
                           const double *WolfeConst2, const double *Increment, ffd_cb f_fd)
{
    fl_default_options(&o, solver);
    if (Strong) o.strong = *Strong != 0;
    if (MaxIteration) o.max_iteration = *MaxIteration;
    if (Precision) o.precision = *Precision;
    if (MinStepLength) o.min_step_length = *MinStepLength;
    if (WolfeConst1) o.wolfe_c1 = *WolfeConst1;
    if (WolfeConst2) o.wolfe_c2 = *WolfeConst2;
    if (Increment) o.increment = *Increment;
    o.fused_f_fd = f_fd != nullptr;
}
static int warn_of(const int32_t *Warning) { return Warning ? *Warning != 0 : 1; }

static int cg_method_of(const char *Method, int len)
{
    if (!Method) return FL_CG_DY; // Method absent: 'DY' (NO.f90:214-215)
    // character*2::type = Method: the first two characters decide (NO.f90:207, 214)
    char t[2] = {len > 0 ? Method[0] : ' ', len > 1 ? Method[1] : ' '};
    if (t[0] == 'D' && t[1] == 'Y') return FL_CG_DY;
    if (t[0] == 'P' && t[1] == 'R') return FL_CG_PR;
    return -1;
}

#define FL_LEGACY_COMMON                                                                                          \
    const int32_t *Strong, const int32_t *Warning, const int *MaxIteration, const double *Precision,              \
        const double *MinStepLength, const double *WolfeConst1, const double *WolfeConst2, const double *Increment

// subroutine SteepestDescent(f,fd,x,dim,f_fd,Strong,...)  NO.f90:55 ; hpp:279-292
void __nonlinearoptimization_MOD_steepestdescent(f_cb f, fd_cb fd, double *x, const int *dim, ffd_cb f_fd,
                                                 FL_LEGACY_COMMON)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_SD, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, f_fd);
    legacy_solve(FL_SOLVER_SD, "steepest descent", host_objective(f, fd, f_fd), x, *dim, o, warn_of(Warning));
}

// subroutine ConjugateGradient(f,fd,x,dim,Method,f_fd,Strong,...)  NO.f90:193 ; hpp:309-324
void __nonlinearoptimization_MOD_conjugategradient(f_cb f, fd_cb fd, double *x, const int *dim, const char *Method,
                                                   ffd_cb f_fd, FL_LEGACY_COMMON, int len_Method)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_CG, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, f_fd);
    const int m = cg_method_of(Method, len_Method);
    if (m < 0) { // reference: write + stop (NO.f90:345); a library must not kill its caller: report and return
        std::printf(" Program abort: unsupported conjugate gradient method %.*s\n", len_Method, Method);
        return;
    }
    o.cg_method = m;
    legacy_solve(FL_SOLVER_CG, "conjugate gradient", host_objective(f, fd, f_fd), x, *dim, o, warn_of(Warning));
}

// subroutine ConjugateGradient_basic(...)  NO.f90:2249-2346 ; hpp:294-307: every argument required,
// no f_fd, and NO fail-safe clamps on the Wolfe constants
void __nonlinearoptimization_MOD_conjugategradient_basic(f_cb f, fd_cb fd, double *x, const int *dim,
                                                         const char *Method, FL_LEGACY_COMMON, int len_Method)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_CG, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, nullptr);
    o.clamp = 0;
    const int m = cg_method_of(Method, len_Method);
    if (m < 0) {
        std::printf(" Program abort: unsupported conjugate gradient method %.*s\n", len_Method, Method);
        return;
    }
    o.cg_method = m;
    legacy_solve(FL_SOLVER_CG, "conjugate gradient", host_objective(f, fd, nullptr), x, *dim, o, warn_of(Warning));
}

// subroutine LBFGS(f,fd,x,dim,Memory,f_fd,Strong,...)  NO.f90:398 (no C++ wrapper in the reference header)
void __nonlinearoptimization_MOD_lbfgs(f_cb f, fd_cb fd, double *x, const int *dim, const int *Memory, ffd_cb f_fd,
                                       FL_LEGACY_COMMON)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_LBFGS, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, f_fd);
    if (Memory) o.memory = *Memory;
    legacy_solve(FL_SOLVER_LBFGS, "L-BFGS", host_objective(f, fd, f_fd), x, *dim, o, warn_of(Warning));
}

// subroutine BFGS(f,fd,x,dim,fdd,ExactStep,f_fd,Strong,...)  NO.f90:632 ; hpp:326-342.
// ExactStep>0 asks for an exact inverse Hessian every ExactStep iterations (NO.f90:674-682, 949-956): the
// machine requests it (FL_REQ_H), fdd -- or central differences of fd when fdd is absent -- answers on the
// host, Cholesky + inverse run on the device.
void __nonlinearoptimization_MOD_bfgs(f_cb f, fd_cb fd, double *x, const int *dim, fdd_cb fdd, const int *ExactStep,
                                      ffd_cb f_fd, FL_LEGACY_COMMON)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_BFGS, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, f_fd);
    int freq = ExactStep ? *ExactStep : 20;
    if (freq > 0 && *dim > 4096) { // the exact inverse Hessian is a dense Cholesky: register path only
        if (warn_of(Warning))
            std::printf(" BFGS (MI355X): dim > 4096, the exact Hessian refresh is skipped (quasi-Newton updates only)\n");
        freq = 0;
    }
    o.exact_step = freq;
    HostObjective ob = host_objective(f, fd, f_fd);
    if (freq > 0 && fdd) {
        ob.fdd = [fdd](double *H, const double *xx, int n) { (void)fdd(H, xx, n); };
    } else if (freq > 0) { // fdd absent: numerical Hessian in place of MKL djacobi (NO.f90:676, 981)
        ob.fdd = central_difference_hessian(fd);
    }
    legacy_solve(FL_SOLVER_BFGS, "BFGS", ob, x, *dim, o, warn_of(Warning));
}

// subroutine NewtonRaphson(f,fd,x,dim,fdd,f_fd,Strong,...)  NO.f90:1026 ; hpp:344-358.
void __nonlinearoptimization_MOD_newtonraphson(f_cb f, fd_cb fd, double *x, const int *dim, fdd_cb fdd, ffd_cb f_fd,
                                               FL_LEGACY_COMMON)
{
    fl_options o;
    legacy_options(o, FL_SOLVER_NEWTON, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2,
                   Increment, f_fd);
    HostObjective ob = host_objective(f, fd, f_fd);
    if (fdd) ob.fdd = [fdd](double *H, const double *xx, int n) { (void)fdd(H, xx, n); };
    else ob.fdd = central_difference_hessian(fd); // NO.f90:1066-1067
    legacy_solve(FL_SOLVER_NEWTON, "Newton-Raphson", ob, x, *dim, o, warn_of(Warning));
}
void nonlinearoptimization_mp_newtonraphson_(f_cb f, fd_cb fd, double *x, const int *dim, fdd_cb fdd, ffd_cb f_fd,
                                             FL_LEGACY_COMMON)
{
    __nonlinearoptimization_MOD_newtonraphson(f, fd, x, dim, fdd, f_fd, Strong, Warning, MaxIteration, Precision,
                                              MinStepLength, WolfeConst1, WolfeConst2, Increment);
}

// subroutine AugmentedLagrangian(f,fd,c,cd,x,N,M,UnconstrainedSolver,lambda0,miu0,fdd,cdd,ExactStep,Memory,Method,
// f_fd,Strong,...)  NO.f90:2005-2241 ; hpp:367-392.  The outer loop and the wrappers L, Ld, L_Ld (NO.f90:2193-2228:
// compositions of the caller's f, fd, c, cd -- part of the objective evaluation, O(N M) on the host like the
// callbacks themselves) follow the reference line by line; every inner solve runs on the GPU through the same
// reverse-communication path as the entry points above.  Solvers: 'LBFGS', 'ConjugateGradient', 'BFGS'
// (quasi-Newton branch, see bfgs above); 'NewtonRaphson' is not on the device path.
typedef void (*c_cb)(double *, const double *, const int &, const int &);
typedef void (*cd_cb)(double *, const double *, const int &, const int &);
typedef int (*cdd_cb)(double *, const double *, const int &, const int &);

void __nonlinearoptimization_MOD_augmentedlagrangian(f_cb f, fd_cb fd, c_cb c, cd_cb cd, double *x, const int *N,
                                                     const int *M, const char *UnconstrainedSolver,
                                                     const double *lambda0, const double *miu0, fdd_cb fdd, cdd_cb cdd,
                                                     const int *ExactStep, const int *Memory, const char *Method,
                                                     ffd_cb f_fd, FL_LEGACY_COMMON, int len_solver, int len_method)
{
    const int n = *N, m = *M;
    std::string solver = UnconstrainedSolver ? std::string(UnconstrainedSolver, (size_t)len_solver) : "BFGS";
    while (!solver.empty() && solver.back() == ' ') solver.pop_back();
    const int warn = warn_of(Warning);
    int sv;
    if (solver == "LBFGS") sv = FL_SOLVER_LBFGS;
    else if (solver == "ConjugateGradient") sv = FL_SOLVER_CG;
    else if (solver == "BFGS") sv = FL_SOLVER_BFGS;
    else if (solver == "NewtonRaphson") sv = FL_SOLVER_NEWTON;
    else { // NO.f90:2186
        std::printf(" Program abort: unsupported unconstrained solver %s\n", solver.c_str());
        return;
    }
    fl_options o;
    legacy_options(o, sv, Strong, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment, nullptr);
    if (!WolfeConst2) o.wolfe_c2 = (sv == FL_SOLVER_CG) ? 0.45 : 0.9; // NO.f90:2053-2062
    if (Memory) o.memory = *Memory;
    o.fused_f_fd = 1; // the inner solver always receives f_fd=L_Ld (NO.f90:2134, 2153, 2171)
    o.exact_step = (sv == FL_SOLVER_BFGS) ? (ExactStep ? *ExactStep : 20) : 0; // freq, NO.f90:2044-2045
    if (sv == FL_SOLVER_CG) {
        const int mth = cg_method_of(Method, Method ? len_method : 0);
        if (mth < 0) {
            std::printf(" Program abort: unsupported conjugate gradient method %.*s\n", len_method, Method);
            return;
        }
        o.cg_method = mth;
    }
    const int maxit = MaxIteration ? *MaxIteration : 1000;
    const double tol = Precision ? *Precision : 1e-15, incrmt = Increment ? *Increment : 1.05;
    std::vector<double> lambda(m, 0.0), cx(m), cdx((size_t)n * m), v(m);
    if (lambda0) lambda.assign(lambda0, lambda0 + m);
    double miu = miu0 ? (*miu0 > 1.0 ? *miu0 : 1.0) : 1.0; // miu=max(1d0,miu0)
    auto seqdot = [](int k, const double *a, const double *b) {
        double t = 0.0;
        for (int i = 0; i < k; ++i) t = t + a[i] * b[i];
        return t;
    };
    auto lx = [&](double &Lx) { Lx = Lx - seqdot(m, lambda.data(), cx.data()) + miu / 2.0 * seqdot(m, cx.data(), cx.data()); };
    auto ldx = [&](double *Ldx) { // Ldx=Ldx+matmul(cdx,miu*cx-lambda)
        for (int j = 0; j < m; ++j) v[j] = miu * cx[j] - lambda[j];
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            for (int j = 0; j < m; ++j) t = t + cdx[(size_t)j * n + i] * v[j];
            Ldx[i] = Ldx[i] + t;
        }
    };
    HostObjective ob;
    ob.f = [&](double &Lx, const double *xx, int nn) { // L, NO.f90:2193-2199
        f(Lx, xx, nn);
        c(cx.data(), xx, m, nn);
        lx(Lx);
    };
    ob.fd = [&](double *Ldx, const double *xx, int nn) { // Ld, NO.f90:2200-2206
        fd(Ldx, xx, nn);
        c(cx.data(), xx, m, nn);
        cd(cdx.data(), xx, m, nn);
        ldx(Ldx);
    };
    ob.f_fd = [&](double &Lx, double *Ldx, const double *xx, int nn) { // L_Ld / L_Ld_fdwithf, NO.f90:2207-2228
        if (f_fd) {
            (void)f_fd(Lx, Ldx, xx, nn);
            c(cx.data(), xx, m, nn);
            lx(Lx);
            cd(cdx.data(), xx, m, nn);
        } else {
            f(Lx, xx, nn);
            c(cx.data(), xx, m, nn);
            lx(Lx);
            fd(Ldx, xx, nn);
            cd(cdx.data(), xx, m, nn);
        }
        ldx(Ldx);
    };
    // Ldd (NO.f90:2229-2240), handed to NewtonRaphson / BFGS when the caller gave fdd AND cdd; as written there:
    // Lddx = f'' + sum_k c''_k (miu c_k - lambda_k) + matmul(cdx, transpose(cdx)) -- no miu on the last term.
    // Without them the inner solver differentiates Ld numerically (the reference: MKL djacobi; here central differences).
    std::vector<double> cddx, vk(m);
    if (sv == FL_SOLVER_NEWTON || (sv == FL_SOLVER_BFGS && o.exact_step > 0)) {
        if (fdd && cdd) {
            cddx.resize((size_t)n * n * m);
            ob.fdd = [&, fdd, cdd](double *Lddx, const double *xx, int nn) {
                (void)fdd(Lddx, xx, nn);
                (void)cdd(cddx.data(), xx, m, nn); // cddx(N,N,M), column-major
                c(cx.data(), xx, m, nn);
                cd(cdx.data(), xx, m, nn);
                for (int k = 0; k < m; ++k) cx[k] = miu * cx[k] - lambda[k]; // cx=miu*cx-lambda
                for (int i = 0; i < nn; ++i)
                    for (int j = 0; j < nn; ++j) {
                        double t = 0.0; // Lddxtemp(j,i)=sum_k cddx(i,j,k)*cx(k)
                        for (int k = 0; k < m; ++k) t = t + cddx[(size_t)k * nn * nn + (size_t)j * nn + i] * cx[k];
                        double cc = 0.0; // matmul(cdx,transpose(cdx))(j,i)
                        for (int k = 0; k < m; ++k) cc = cc + cdx[(size_t)k * nn + j] * cdx[(size_t)k * nn + i];
                        Lddx[(size_t)i * nn + j] = Lddx[(size_t)i * nn + j] + t + cc;
                    }
            };
        } else {
            ob.fdd = central_difference_hessian(ob.fd);
        }
    }
    const char *names[] = {"steepest descent", "conjugate gradient", "L-BFGS", "BFGS", "Newton-Raphson"};
    int it = 1;
    for (; it <= maxit; ++it) {
        if (legacy_solve(sv, names[sv], ob, x, n, o, warn) < 0) return;
        c(cx.data(), x, m, n);
        if (seqdot(m, cx.data(), cx.data()) < tol * tol) break; // if(dot_product(cx,cx)<tolsq) exit
        for (int j = 0; j < m; ++j) lambda[j] = lambda[j] - miu * cx[j];
        miu = miu * incrmt;
    }
    if (it > maxit && warn)
        std::printf(" Failed augmented Lagrangian: max iteration exceeded!\n Euclidean norm of constraint violation = %24.16E\n",
                    sqrt(seqdot(m, cx.data(), cx.data())));
}
void nonlinearoptimization_mp_augmentedlagrangian_(f_cb f, fd_cb fd, c_cb c, cd_cb cd, double *x, const int *N,
                                                   const int *M, const char *UnconstrainedSolver,
                                                   const double *lambda0, const double *miu0, fdd_cb fdd, cdd_cb cdd,
                                                   const int *ExactStep, const int *Memory, const char *Method,
                                                   ffd_cb f_fd, FL_LEGACY_COMMON, int len_solver, int len_method)
{
    __nonlinearoptimization_MOD_augmentedlagrangian(f, fd, c, cd, x, N, M, UnconstrainedSolver, lambda0, miu0, fdd, cdd,
                                                    ExactStep, Memory, Method, f_fd, Strong, Warning, MaxIteration,
                                                    Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment,
                                                    len_solver, len_method);
}

// ifort manglings of the same procedures (cpp/NonlinearOptimization.hpp:11-123)
void nonlinearoptimization_mp_steepestdescent_(f_cb f, fd_cb fd, double *x, const int *dim, ffd_cb f_fd,
                                               FL_LEGACY_COMMON)
{
    __nonlinearoptimization_MOD_steepestdescent(f, fd, x, dim, f_fd, Strong, Warning, MaxIteration, Precision,
                                                MinStepLength, WolfeConst1, WolfeConst2, Increment);
}
void nonlinearoptimization_mp_conjugategradient_(f_cb f, fd_cb fd, double *x, const int *dim, const char *Method,
                                                 ffd_cb f_fd, FL_LEGACY_COMMON, int len_Method)
{
    __nonlinearoptimization_MOD_conjugategradient(f, fd, x, dim, Method, f_fd, Strong, Warning, MaxIteration,
                                                  Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment,
                                                  len_Method);
}
void nonlinearoptimization_mp_conjugategradient_basic_(f_cb f, fd_cb fd, double *x, const int *dim,
                                                       const char *Method, FL_LEGACY_COMMON, int len_Method)
{
    __nonlinearoptimization_MOD_conjugategradient_basic(f, fd, x, dim, Method, Strong, Warning, MaxIteration,
                                                        Precision, MinStepLength, WolfeConst1, WolfeConst2,
                                                        Increment, len_Method);
}
void nonlinearoptimization_mp_lbfgs_(f_cb f, fd_cb fd, double *x, const int *dim, const int *Memory, ffd_cb f_fd,
                                     FL_LEGACY_COMMON)
{
    __nonlinearoptimization_MOD_lbfgs(f, fd, x, dim, Memory, f_fd, Strong, Warning, MaxIteration, Precision,
                                      MinStepLength, WolfeConst1, WolfeConst2, Increment);
}
void nonlinearoptimization_mp_bfgs_(f_cb f, fd_cb fd, double *x, const int *dim, fdd_cb fdd, const int *ExactStep,
                                    ffd_cb f_fd, FL_LEGACY_COMMON)
{
    __nonlinearoptimization_MOD_bfgs(f, fd, x, dim, fdd, ExactStep, f_fd, Strong, Warning, MaxIteration, Precision,
                                     MinStepLength, WolfeConst1, WolfeConst2, Increment);
}

// subroutine Wolfe / Wolfe_fdwithf / StrongWolfe / StrongWolfe_fdwithf (c1,c2,f,fd[,f_fd],x,a,p,fx,phid0,fdx,dim,Increment)
// NO.f90:1286, 1373, 1462, 1582.  Wolfe_fdwithf has the body of Wolfe: it never calls f_fd.
void __nonlinearoptimization_MOD_wolfe(const double *c1, const double *c2, f_cb f, fd_cb fd, double *x, double *a,
                                       const double *p, double *fx, const double *phid0, double *fdx, const int *dim,
                                       const double *Increment)
{
    legacy_line_search(0, 0, c1, c2, f, fd, nullptr, x, a, p, fx, phid0, fdx, dim, Increment);
}
void __nonlinearoptimization_MOD_wolfe_fdwithf(const double *c1, const double *c2, f_cb f, fd_cb fd, ffd_cb f_fd,
                                               double *x, double *a, const double *p, double *fx, const double *phid0,
                                               double *fdx, const int *dim, const double *Increment)
{
    (void)f_fd;
    legacy_line_search(0, 0, c1, c2, f, fd, nullptr, x, a, p, fx, phid0, fdx, dim, Increment);
}
void __nonlinearoptimization_MOD_strongwolfe(const double *c1, const double *c2, f_cb f, fd_cb fd, double *x, double *a,
                                             const double *p, double *fx, const double *phid0, double *fdx,
                                             const int *dim, const double *Increment)
{
    legacy_line_search(1, 0, c1, c2, f, fd, nullptr, x, a, p, fx, phid0, fdx, dim, Increment);
}
void __nonlinearoptimization_MOD_strongwolfe_fdwithf(const double *c1, const double *c2, f_cb f, fd_cb fd, ffd_cb f_fd,
                                                     double *x, double *a, const double *p, double *fx,
                                                     const double *phid0, double *fdx, const int *dim,
                                                     const double *Increment)
{
    legacy_line_search(1, f_fd != nullptr, c1, c2, f, fd, f_fd, x, a, p, fx, phid0, fdx, dim, Increment);
}
void nonlinearoptimization_mp_wolfe_(const double *c1, const double *c2, f_cb f, fd_cb fd, double *x, double *a,
                                     const double *p, double *fx, const double *phid0, double *fdx, const int *dim,
                                     const double *Increment)
{
    __nonlinearoptimization_MOD_wolfe(c1, c2, f, fd, x, a, p, fx, phid0, fdx, dim, Increment);
}
void nonlinearoptimization_mp_wolfe_fdwithf_(const double *c1, const double *c2, f_cb f, fd_cb fd, ffd_cb f_fd, double *x,
                                             double *a, const double *p, double *fx, const double *phid0, double *fdx,
                                             const int *dim, const double *Increment)
{
    __nonlinearoptimization_MOD_wolfe_fdwithf(c1, c2, f, fd, f_fd, x, a, p, fx, phid0, fdx, dim, Increment);
}
void nonlinearoptimization_mp_strongwolfe_(const double *c1, const double *c2, f_cb f, fd_cb fd, double *x, double *a,
                                           const double *p, double *fx, const double *phid0, double *fdx,
                                           const int *dim, const double *Increment)
{
    __nonlinearoptimization_MOD_strongwolfe(c1, c2, f, fd, x, a, p, fx, phid0, fdx, dim, Increment);
}
void nonlinearoptimization_mp_strongwolfe_fdwithf_(const double *c1, const double *c2, f_cb f, fd_cb fd, ffd_cb f_fd,
                                                   double *x, double *a, const double *p, double *fx,
                                                   const double *phid0, double *fdx, const int *dim,
                                                   const double *Increment)
{
    __nonlinearoptimization_MOD_strongwolfe_fdwithf(c1, c2, f, fd, f_fd, x, a, p, fx, phid0, fdx, dim, Increment);
}

// subroutine LagrangianMultiplier(fd,fdd,c,cd,cdd,x,lambda,N,M,Warning,MaxIteration,Precision)  NO.f90:1950-1993:
// Newton iteration on the KKT system of L = f - lambda.c.  The caller's callbacks and the assembly of -L', L''
// (NO.f90:1972-1981: compositions of their outputs) run on the host; the (N+M)-dimensional symmetric indefinite
// solve My_dsysv (NO.f90:1984) runs on the GPU (fl_dsysv_batched).
void __nonlinearoptimization_MOD_lagrangianmultiplier(fd_cb fd, fdd_cb fdd, c_cb c, cd_cb cd, cdd_cb cdd, double *x,
                                                      double *lambda, const int *N, const int *M,
                                                      const int32_t *Warning, const int *MaxIteration,
                                                      const double *Precision)
{
    const int n = *N, m = *M, dim = n + m;
    const int warn = warn_of(Warning), maxit = MaxIteration ? *MaxIteration : 1000;
    const double tol = Precision ? *Precision * *Precision : 1e-30;
    int threads = 0, ept = 0;
    if (dim > 4096 || fl_reduction_geometry(dim, &threads, &ept) != FL_OK) {
        std::fprintf(stderr, "FortranLibrary(MI355X) LagrangianMultiplier: N+M = %d beyond the dense solver (4096)\n", dim);
        return;
    }
    const size_t ld = (size_t)threads * ept;
    std::vector<double> mLd(dim), cx(m), cdx((size_t)n * m), cddx((size_t)n * n * m), H((size_t)n * n), Ldd((size_t)dim * ld);
    double *Ad = nullptr, *bd = nullptr;
    int32_t *infod = nullptr;
    bool ok = hipMalloc((void **)&Ad, sizeof(double) * dim * ld) == hipSuccess &&
              hipMalloc((void **)&bd, sizeof(double) * dim) == hipSuccess &&
              hipMalloc((void **)&infod, sizeof(int32_t)) == hipSuccess;
    auto minus_gradient = [&]() { // minusLd(1:N)=matmul(cdx,lambda)-f'(x); minusLd(N+1:dim)=cx; returns its square norm
        fd(mLd.data(), x, n);
        c(cx.data(), x, m, n);
        cd(cdx.data(), x, m, n);
        for (int i = 0; i < n; ++i) {
            double t = 0.0;
            for (int k = 0; k < m; ++k) t = t + cdx[(size_t)k * n + i] * lambda[k];
            mLd[i] = t - mLd[i];
        }
        for (int k = 0; k < m; ++k) mLd[n + k] = cx[k];
        double nrm = 0.0;
        for (int i = 0; i < dim; ++i) nrm = nrm + mLd[i] * mLd[i];
        return nrm;
    };
    int it = 1;
    for (; ok && it <= maxit; ++it) {
        if (minus_gradient() < tol) break; // Converged
        (void)fdd(H.data(), x, n);
        (void)cdd(cddx.data(), x, m, n);
        std::fill(Ldd.begin(), Ldd.end(), 0.0);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) { // Ldd(i,1:N)=Ldd(i,1:N)-matmul(cddx(i,:,:),lambda)
                double t = 0.0;
                for (int k = 0; k < m; ++k) t = t + cddx[(size_t)k * n * n + (size_t)j * n + i] * lambda[k];
                Ldd[(size_t)j * ld + i] = H[(size_t)j * n + i] - t;
            }
        for (int k = 0; k < m; ++k)
            for (int j = 0; j < n; ++j) Ldd[(size_t)j * ld + n + k] = -cdx[(size_t)k * n + j]; // -transpose(cdx)
        int32_t info = 0;
        ok = hipMemcpy(Ad, Ldd.data(), sizeof(double) * dim * ld, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(bd, mLd.data(), sizeof(double) * dim, hipMemcpyHostToDevice) == hipSuccess &&
             fl_dsysv_batched(1, dim, Ad, bd, infod, nullptr) == FL_OK &&
             hipMemcpy(&info, infod, sizeof info, hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(mLd.data(), bd, sizeof(double) * dim, hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok || info != 0) break;
        for (int i = 0; i < n; ++i) x[i] = x[i] + mLd[i];
        for (int k = 0; k < m; ++k) lambda[k] = lambda[k] + mLd[n + k];
    }
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) LagrangianMultiplier: HIP error\n");
    if (ok && it > maxit && warn) {
        std::printf(" Failed Lagrangian multiplier: max iteration exceeded!\n");
        std::printf(" Euclidean norm of Lagrangian gradient = %24.16E\n", sqrt(minus_gradient()));
    }
    if (Ad) (void)hipFree(Ad);
    if (bd) (void)hipFree(bd);
    if (infod) (void)hipFree(infod);
}
void nonlinearoptimization_mp_lagrangianmultiplier_(fd_cb fd, fdd_cb fdd, c_cb c, cd_cb cd, cdd_cb cdd, double *x,
                                                    double *lambda, const int *N, const int *M, const int32_t *Warning,
                                                    const int *MaxIteration, const double *Precision)
{
    __nonlinearoptimization_MOD_lagrangianmultiplier(fd, fdd, c, cd, cdd, x, lambda, N, M, Warning, MaxIteration,
                                                     Precision);
}

} // extern "C"
