// fl_solver_kernels.hip -- batched SteepestDescent / ConjugateGradient / L-BFGS
// for MI355X (gfx950): ONE workgroup owns ONE problem for its whole solve.
//
// Reference semantics: /root/reference/source/NonlinearOptimization.f90
//   SteepestDescent 55-188, ConjugateGradient 193-394 (DY 352-372, PR 373-393),
//   LBFGS 398-625 (pre-iteration 472-510, Before 586-608, After 609-624),
//   line searchers 1286-1698 (fl_linesearch.hpp).
//
// Layout / mapping (CDNA4-first, not a translation of the Fortran loops):
//   * x, g, p, xold, gold (and the objective's data) live in REGISTERS for the
//     whole solve: thread t of the T = 64*NW threads holds EPT elements, dealt in
//     16-byte chunks round-robin (element (c*T + t)*2 + j), so every global access
//     is a perfectly coalesced dwordx4 stream.
//   * a line-search trial (x = x0 + a p, f, grad f, g.p, g.g) touches no HBM at
//     all; only the L-BFGS (s,y) ring streams from HBM: 4*m vector reads + 2
//     vector writes per iteration and problem -- the two-loop recursion is the
//     only HBM-bound phase, and its loads for step j+1 are in flight while
//     step j reduces.
//   * dot products: per-thread partial over the EPT elements, 64-lane xor
//     butterfly (DPP + v_permlane swaps, fl_reduce.hpp), then the NW wave partials
//     summed left to right through LDS.  The order is fixed, so results are reproducible
//     bit for bit (tests replay it on the CPU).
//   * all scalars of the line-search machine are workgroup-uniform: every branch
//     is taken by all threads, barriers are safe inside the state machine.
//   * problems finish after different numbers of trials; the grid is one
//     workgroup per problem and the hardware dispatcher back-fills CUs as
//     workgroups retire (no lock-step batch, no host round trips).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fl_nlopt.h"
#include "fl_linesearch.hpp"
#include "fl_reduce.hpp"

#ifdef FL_MIN_WPE
#define FL_OCC_ATTR __attribute__((amdgpu_waves_per_eu(FL_MIN_WPE)))
#else
#define FL_OCC_ATTR
#endif

namespace fl {

struct SolveArgs {
    int n, batch, mem, maxit, strong, fused, cg_method;
    double tol, minstep, c1, c2, incr;
    double *x;
    const double *d, *b;
    double *hist;
    double *f_out, *gg_out;
    int *iters, *status, *nf, *ng;
};

template <int EPT> struct Vec {
    double v[EPT];
};

template <int NW, int EPT> struct Geo {
    static constexpr int T = NW * 64;
    static constexpr int NPAD = T * EPT;
    static constexpr int NCH = EPT / 2;
    __device__ __forceinline__ static int e0(int c) { return ((c * T + (int)threadIdx.x) << 1); }
};

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
// history rows are padded to NPAD and always 16-byte aligned
template <int NW, int EPT> __device__ __forceinline__ void load_hist(const double *row, double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
        const double2 t = *reinterpret_cast<const double2 *>(row + G::e0(c));
        v[2 * c] = t.x;
        v[2 * c + 1] = t.y;
    }
}
template <int NW, int EPT> __device__ __forceinline__ void store_hist(double *row, const double (&v)[EPT])
{
    using G = Geo<NW, EPT>;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c)
        *reinterpret_cast<double2 *>(row + G::e0(c)) = make_double2(v[2 * c], v[2 * c + 1]);
}

template <int EPT> __device__ __forceinline__ double dot_part(const double (&a)[EPT], const double (&b)[EPT])
{
    double acc = a[0] * b[0];
#pragma unroll
    for (int k = 1; k < EPT; ++k) acc = acc + a[k] * b[k];
    return acc;
}

// ------------------------------------------------------------ objectives
// Each objective evaluates, for the thread's EPT elements, the gradient and up
// to two partial sums (f = combine(S0, S1)).  Padded elements (e >= n) carry
// x = 0 and must produce g = 0 and zero terms.
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
};

template <int NW, int EPT> struct Objective<FL_OBJ_DIAGQUAD, NW, EPT> { // f=0.5*sum(d*x*x)-sum(b*x), g=d*x-b
    static constexpr int LDS_DOUBLES = 0;
    double d[EPT], b[EPT];
    __device__ __forceinline__ void init(const SolveArgs &A, int prob, double *)
    {
        load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, d);
        load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, b);
    }
    __device__ __forceinline__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int,
                                         double *)
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const double dx = d[k] * x[k];
            const double t0 = dx * x[k], t1 = b[k] * x[k];
            g[k] = dx - b[k];
            s0 = (k == 0) ? t0 : s0 + t0;
            s1 = (k == 0) ? t1 : s1 + t1;
        }
    }
    __device__ __forceinline__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
};

template <int NW, int EPT> struct Objective<FL_OBJ_ROSENBROCK, NW, EPT> {
    // chained Rosenbrock; x is staged through LDS (one halo element each side) so
    // that every thread can read its chunk's neighbours x[e-1], x[e+2]
    using G = Geo<NW, EPT>;
    static constexpr int LDS_DOUBLES = G::NPAD + 2;
    __device__ __forceinline__ void init(const SolveArgs &, int, double *xs)
    {
        if (threadIdx.x == 0) {
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
};

// ------------------------------------------------------------ the solver
template <int NW, int EPT, int OBJ, int METHOD>
__global__ __launch_bounds__(NW * 64) FL_OCC_ATTR void fl_solve_kernel(SolveArgs A)
{
    using G = Geo<NW, EPT>;
    using Obj = Objective<OBJ, NW, EPT>;
    constexpr int NPAD = G::NPAD;
    __shared__ double red_slots[2 * 4 * NW];
    __shared__ double rho_s[FL_MAX_MEMORY], alpha_s[FL_MAX_MEMORY];
    __shared__ double xs[Obj::LDS_DOUBLES > 0 ? Obj::LDS_DOUBLES : 1];
    // gold (the gradient before the line search) is parked in LDS for the duration of the search:
    // every thread writes and later reads back only its own 16-byte chunks (no barrier needed)
    __shared__ __attribute__((aligned(16))) double g0s[METHOD == FL_SOLVER_SD ? 2 : NPAD];

    const int prob = blockIdx.x;
    const int n = A.n;
    Reducer<NW> R4{red_slots, 0};

    double x[EPT], g[EPT], p[EPT], x0[EPT];
    Obj obj;
    obj.init(A, prob, xs);
    load_user<NW, EPT>(A.x + (size_t)prob * n, n, x);
#pragma unroll
    for (int k = 0; k < EPT; ++k) p[k] = 0.0;

    // evaluate f, g at x; returns f, g.p, g.g in one workgroup reduction
    auto evaluate = [&](double &f, double &gp, double &gg) {
        double r[4];
        obj.eval(x, g, r[0], r[1], n, xs);
        r[2] = dot_part<EPT>(g, p);
        r[3] = dot_part<EPT>(g, g);
        R4.run(r);
        f = LineSearch::uni(Obj::combine(r[0], r[1]));
        gp = LineSearch::uni(r[2]);
        gg = LineSearch::uni(r[3]);
    };

    int nf = 1, ng = 1, iters = 0, status = FL_STATUS_CONVERGED;
    double fnew, gp, gg, pp, phid, a;
    evaluate(fnew, gp, gg); // initial f(x), f'(x): NO.f90:87-91 / 230-234 / 436-440
#pragma unroll
    for (int k = 0; k < EPT; ++k) p[k] = -g[k];
    phid = -gg; // p=-fdnew; phidnew=-dot_product(fdnew,fdnew)
    pp = gg;
    bool finished = gg < A.tol; // if(-phidnew<tol) return
    a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(gg);

    // L-BFGS ring
    double *hist = nullptr;
    int recent = -1, cnt = 0;
    double yy_recent = 0.0;
    const int mem = A.mem;
    if constexpr (METHOD == FL_SOLVER_LBFGS) hist = A.hist + (size_t)prob * (size_t)(2 * mem) * NPAD;
    // line searches allowed: L-BFGS 1 + (mem-1) pre-iterations + maxit; SD/CG maxit
    const int max_ls = (METHOD == FL_SOLVER_LBFGS) ? mem + A.maxit : A.maxit;
    if (!finished) status = FL_STATUS_MAXIT;

    while (!finished && iters < max_ls) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { // xold=x; fdold=fdnew (x0 doubles as the line search's x0)
            x0[k] = x[k];
        }
        if constexpr (METHOD != FL_SOLVER_SD) {
#pragma unroll
            for (int c = 0; c < G::NCH; ++c)
                *reinterpret_cast<double2 *>(g0s + G::e0(c)) = make_double2(g[2 * c], g[2 * c + 1]);
        }
        const double phidold = phid;
        // which searcher: the *_fdwithf variants only in main loops with f_fd present
        // (L-BFGS' first search and pre-iterations never use f_fd: NO.f90:448-460, 486-498)
        int fused = A.fused;
        if constexpr (METHOD == FL_SOLVER_LBFGS) fused = fused && (iters >= mem);
        const int strong = (METHOD == FL_SOLVER_CG && A.cg_method == FL_CG_PR) ? 1 : A.strong;

        LineSearch ls;
        int rq = ls.begin(strong, fused, A.c1, A.c2, A.incr, a, fnew, phid);
        ls.uniformize();
        double fv = fnew, pv = phid;
        while (rq) {
            if (!(rq & FL_REQ_SAME)) {
                const double at = ls.a_eval;
#pragma unroll
                for (int k = 0; k < EPT; ++k) x[k] = x0[k] + at * p[k];
                evaluate(fv, pv, gg);
            }
            nf += (rq & FL_REQ_F) ? 1 : 0;
            ng += (rq & FL_REQ_G) ? 1 : 0;
            rq = __builtin_amdgcn_readfirstlane(ls.step(fv, pv));
            ls.uniformize();
        }
        a = ls.a;
        fnew = ls.fx;
        ++iters;
        double g0[EPT];
        if constexpr (METHOD != FL_SOLVER_SD) {
#pragma unroll
            for (int c = 0; c < G::NCH; ++c) {
                const double2 t = *reinterpret_cast<const double2 *>(g0s + G::e0(c));
                g0[2 * c] = t.x;
                g0[2 * c + 1] = t.y;
            }
        }

        // convergence tests on the new gradient (After / DY / PR, NO.f90:171-183, 353-364, 610-621)
        if (gg < A.tol) {
            status = FL_STATUS_CONVERGED;
            finished = true;
            break;
        }
        if (pp * a * a < A.minstep) {
            status = FL_STATUS_STEP_CONVERGED;
            finished = true;
            break;
        }

        if constexpr (METHOD == FL_SOLVER_SD) { // NO.f90:185-186
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = -g[k];
            phid = -gg;
            pp = gg;
            a = a * phidold / phid;
        } else if constexpr (METHOD == FL_SOLVER_CG) {
            double yk[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) yk[k] = g[k] - g0[k];
            double beta;
            if (A.cg_method == FL_CG_DY) { // p=-g+(g.g)/((g-gold).p)*p, NO.f90:366
                double q[1] = {dot_part<EPT>(yk, p)};
                R4.run(q);
                beta = gg / q[0];
            } else { // p=-g+(g.(g-gold))/(gold.gold)*p, NO.f90:387
                double q[2] = {dot_part<EPT>(g, yk), dot_part<EPT>(g0, g0)};
                R4.run(q);
                beta = q[0] / q[1];
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = -g[k] + beta * p[k];
            double q2[2] = {dot_part<EPT>(g, p), dot_part<EPT>(p, p)};
            R4.run(q2);
            phid = q2[0];
            pp = q2[1];
            if (phid > 0.0) { // ascent direction: reset to steepest descent, NO.f90:368-370
#pragma unroll
                for (int k = 0; k < EPT; ++k) p[k] = -g[k];
                phid = -gg;
                pp = gg;
            }
            a = a * phidold / phid;
        } else { // L-BFGS: store the newest pair, then the two-loop recursion
            recent = (recent + 1 == mem) ? 0 : recent + 1; // recent=mod(recent+1,mem)
            if (cnt < mem) ++cnt;
            double r[2], sv[EPT], yv[EPT];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                sv[k] = x[k] - x0[k];
                yv[k] = g[k] - g0[k];
            }
            store_hist<NW, EPT>(hist + (size_t)(2 * recent) * NPAD, sv);
            store_hist<NW, EPT>(hist + (size_t)(2 * recent + 1) * NPAD, yv);
            r[0] = dot_part<EPT>(yv, sv);
            r[1] = dot_part<EPT>(yv, yv);
            R4.run(r);
            if (threadIdx.x == 0) rho_s[recent] = 1.0 / r[0]; // rho=1/(y.s): no curvature safeguard
            const double rho_recent = 1.0 / r[0];
            yy_recent = r[1];
            // make the stores visible to this workgroup's later loads of the same rows
            __syncthreads();

            // Before(): two-loop recursion, newest -> oldest, then oldest -> newest.
            // p = g in registers; the ring streams from HBM, next step's rows in flight.
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = g[k];
            double sA[EPT], yA[EPT], sB[EPT], yB[EPT];
            auto slot_of = [&](int j) { // j-th newest
                int s = recent - j;
                return s < 0 ? s + mem : s;
            };
            auto fetch = [&](int j, double (&s_)[EPT], double (&y_)[EPT]) {
                const double *row = hist + (size_t)(2 * slot_of(j)) * NPAD;
                load_hist<NW, EPT>(row, s_);
                load_hist<NW, EPT>(row + NPAD, y_);
            };
            auto down = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
                const int sl = slot_of(j);
                double q[1] = {dot_part<EPT>(s_, p)};
                R4.run(q);
                const double al = rho_s[sl] * q[0]; // alpha(i)=rho(i)*dot_product(s(:,i),p)
                if (threadIdx.x == 0) alpha_s[sl] = al;
#pragma unroll
                for (int k = 0; k < EPT; ++k) p[k] = p[k] - al * y_[k];
            };
            auto upw = [&](int j, const double (&s_)[EPT], const double (&y_)[EPT]) {
                const int sl = slot_of(j);
                double q[1] = {dot_part<EPT>(y_, p)};
                R4.run(q);
                const double be = rho_s[sl] * q[0]; // phidnew=rho(i)*dot_product(y(:,i),p)
                const double co = alpha_s[sl] - be;
#pragma unroll
                for (int k = 0; k < EPT; ++k) p[k] = p[k] + co * s_[k];
            };
            fetch(0, sA, yA);
            for (int j = 0; j < cnt; j += 2) {
                if (j + 1 < cnt) fetch(j + 1, sB, yB);
                down(j, sA, yA);
                if (j + 1 < cnt) {
                    if (j + 2 < cnt) fetch(j + 2, sA, yA);
                    down(j + 1, sB, yB);
                }
            }
            // oldest pair first on the way back: start its loads before the scaling
            fetch(cnt - 1, sA, yA);
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = p[k] / rho_recent / yy_recent; // p=p/rho(recent)/(y.y)
            __syncthreads(); // alpha_s written by thread 0 is visible (NW == 1 has no reduction barrier)
            for (int j = cnt - 1; j >= 0; j -= 2) {
                if (j - 1 >= 0) fetch(j - 1, sB, yB);
                upw(j, sA, yA);
                if (j - 1 >= 0) {
                    if (j - 2 >= 0) fetch(j - 2, sA, yA);
                    upw(j - 1, sB, yB);
                }
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k) p[k] = -p[k];
            r[0] = dot_part<EPT>(g, p);
            r[1] = dot_part<EPT>(p, p);
            R4.run(r);
            phid = r[0]; // phidnew=dot_product(fdnew,p)
            pp = r[1];
            a = 1.0;
        }
        phid = LineSearch::uni(phid);
        pp = LineSearch::uni(pp);
        a = LineSearch::uni(a);
    }

    store_user<NW, EPT>(A.x + (size_t)prob * n, n, x);
    if (threadIdx.x == 0) {
        if (A.f_out) A.f_out[prob] = fnew;
        if (A.gg_out) A.gg_out[prob] = gg;
        if (A.iters) A.iters[prob] = iters;
        if (A.status) A.status[prob] = status;
        if (A.nf) A.nf[prob] = nf;
        if (A.ng) A.ng[prob] = ng;
    }
}

// ------------------------------------------------------------ host dispatch
struct GeoSel {
    int nw, ept;
};
static bool select_geometry(int n, GeoSel &g)
{
    if (n <= 0) return false;
    if (n <= 128) g = {1, 2};
    else if (n <= 256) g = {1, 4};
    else if (n <= 512) g = {2, 4};
    else if (n <= 1024) g = {4, 4};
    else if (n <= 2048) g = {4, 8};
    else if (n <= 4096) g = {8, 8};
    else return false;
    return true;
}

template <int NW, int EPT, int OBJ> static hipError_t launch_m(int method, const SolveArgs &A, hipStream_t st)
{
    dim3 grid(A.batch), block(NW * 64);
    switch (method) {
    case FL_SOLVER_SD:
        hipLaunchKernelGGL((fl_solve_kernel<NW, EPT, OBJ, FL_SOLVER_SD>), grid, block, 0, st, A);
        break;
    case FL_SOLVER_CG:
        hipLaunchKernelGGL((fl_solve_kernel<NW, EPT, OBJ, FL_SOLVER_CG>), grid, block, 0, st, A);
        break;
    default:
        hipLaunchKernelGGL((fl_solve_kernel<NW, EPT, OBJ, FL_SOLVER_LBFGS>), grid, block, 0, st, A);
        break;
    }
    return hipGetLastError();
}
template <int NW, int EPT> static hipError_t launch_o(int obj, int method, const SolveArgs &A, hipStream_t st)
{
    switch (obj) {
    case FL_OBJ_QUARTIC: return launch_m<NW, EPT, FL_OBJ_QUARTIC>(method, A, st);
    case FL_OBJ_ROSENBROCK: return launch_m<NW, EPT, FL_OBJ_ROSENBROCK>(method, A, st);
    default: return launch_m<NW, EPT, FL_OBJ_DIAGQUAD>(method, A, st);
    }
}
static hipError_t launch(const GeoSel &g, int obj, int method, const SolveArgs &A, hipStream_t st)
{
    if (g.nw == 1 && g.ept == 2) return launch_o<1, 2>(obj, method, A, st);
    if (g.nw == 1 && g.ept == 4) return launch_o<1, 4>(obj, method, A, st);
    if (g.nw == 2 && g.ept == 4) return launch_o<2, 4>(obj, method, A, st);
    if (g.nw == 4 && g.ept == 4) return launch_o<4, 4>(obj, method, A, st);
    if (g.nw == 4 && g.ept == 8) return launch_o<4, 8>(obj, method, A, st);
    return launch_o<8, 8>(obj, method, A, st);
}

static int solve(int method, int objective, int batch, int n, double *x, const double *d, const double *b,
                 const fl_options *opt, void *ws, size_t ws_bytes, double *f, double *gg, int32_t *iters,
                 int32_t *status, int32_t *nf, int32_t *ng, void *stream)
{
    if (!x || !opt || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (objective < FL_OBJ_QUARTIC || objective > FL_OBJ_DIAGQUAD) return FL_ERR_INVALID_ARGUMENT;
    if (objective == FL_OBJ_DIAGQUAD && (!d || !b)) return FL_ERR_INVALID_ARGUMENT;
    if (opt->cg_method != FL_CG_DY && opt->cg_method != FL_CG_PR) return FL_ERR_INVALID_ARGUMENT;
    GeoSel g;
    if (!select_geometry(n, g)) return FL_ERR_UNSUPPORTED_SIZE;
    SolveArgs A;
    A.n = n;
    A.batch = batch;
    A.mem = opt->memory > 1 ? opt->memory : 1; // mem=max(1,Memory)
    if (method == FL_SOLVER_LBFGS) {
        if (A.mem > FL_MAX_MEMORY) return FL_ERR_UNSUPPORTED_SIZE;
        if (!ws || ws_bytes < fl_workspace_bytes(method, batch, n, A.mem)) return FL_ERR_WORKSPACE;
    }
    A.maxit = opt->max_iteration;
    A.strong = opt->strong != 0;
    A.fused = opt->fused_f_fd != 0;
    A.cg_method = opt->cg_method;
    A.tol = opt->precision * opt->precision;                  // NO.f90:427
    A.minstep = opt->min_step_length * opt->min_step_length;  // NO.f90:429
    A.c1 = opt->wolfe_c1;
    A.c2 = opt->wolfe_c2;
    if (opt->clamp) { // NO.f90:431-434
        A.c1 = opt->wolfe_c1 > 1e-15 ? opt->wolfe_c1 : 1e-15;
        double lo = A.c1 + 1e-15;
        double c2 = opt->wolfe_c2 > lo ? opt->wolfe_c2 : lo;
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
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipError_t e = launch(g, objective, method, A, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? FL_OK : FL_ERR_NO_DEVICE;
}

} // namespace fl

extern "C" {

int fl_version(void) { return 100; }

void fl_default_options(fl_options *o, int solver)
{
    if (!o) return;
    o->strong = 1;
    o->max_iteration = 1000;
    o->precision = 1e-15;
    o->min_step_length = 1e-15;
    o->wolfe_c1 = 1e-4;
    o->wolfe_c2 = (solver == FL_SOLVER_CG) ? 0.45 : 0.9;
    o->increment = 1.05;
    o->memory = 10;
    o->cg_method = FL_CG_DY;
    o->fused_f_fd = 0;
    o->clamp = 1;
}

int fl_reduction_geometry(int n, int *threads, int *ept)
{
    fl::GeoSel g;
    if (!fl::select_geometry(n, g)) return FL_ERR_UNSUPPORTED_SIZE;
    if (threads) *threads = g.nw * 64;
    if (ept) *ept = g.ept;
    return FL_OK;
}

size_t fl_workspace_bytes(int solver, int batch, int n, int memory)
{
    fl::GeoSel g;
    if (solver != FL_SOLVER_LBFGS || batch <= 0 || !fl::select_geometry(n, g)) return 0;
    const size_t mem = memory > 1 ? (size_t)memory : 1;
    return (size_t)batch * 2 * mem * (size_t)(g.nw * 64 * g.ept) * sizeof(double);
}

int fl_lbfgs_batched(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                     const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                     double *gg_dev, int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                     void *stream)
{
    return fl::solve(FL_SOLVER_LBFGS, objective, batch, n, x_dev, d_dev, b_dev, opt, workspace_dev, workspace_bytes,
                     f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, stream);
}

int fl_conjugate_gradient_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                  const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                  int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                  void *stream)
{
    return fl::solve(FL_SOLVER_CG, objective, batch, n, x_dev, d_dev, b_dev, opt, nullptr, 0, f_dev, gg_dev,
                     iters_dev, status_dev, nf_dev, ng_dev, stream);
}

int fl_steepest_descent_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                void *stream)
{
    return fl::solve(FL_SOLVER_SD, objective, batch, n, x_dev, d_dev, b_dev, opt, nullptr, 0, f_dev, gg_dev,
                     iters_dev, status_dev, nf_dev, ng_dev, stream);
}

} // extern "C"
