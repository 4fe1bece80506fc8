// fl_big.hpp -- the solver machine for problems that do not fit the register path (n > 4096).
//
// Same algorithm, same request protocol and the same fixed summation order as Solver (fl_device.hpp), restated
// over vectors that live in HBM: ONE workgroup of 1024 threads owns one problem; thread t handles the element
// pairs (c*1024 + t)*2 + {0,1}, c = 0..nslot-1 -- the register path's layout with EPT = 2*nslot -- and sums its
// terms in that order, then the 16 waves left to right, so the oracle replays it with threads = 1024,
// ept = 2*nslot (fl_reduction_geometry reports both).  SteepestDescent (NO.f90:55-188), ConjugateGradient
// (193-394), L-BFGS (398-625), and BFGS with quasi-Newton updates only (632-1022, ExactStep <= 0) up to n = 16384;
// NewtonRaphson and the exact-Hessian refresh (dense Cholesky) stay on the register path.
// Every element is only ever touched by its own thread, so passes need no barrier besides the reductions --
// except Rosenbrock's neighbour reads, which follow a __syncthreads() after x has been written.
// The L-BFGS recursion is fused pass-wise: the axpy of step j and the dot product of step j+1 share one pass.
#pragma once
#include "fl_device.hpp"

namespace fl {

// ---- the caller's own objective on this path (OBJ = FL_OBJ_USER): the STREAMING functor.  The register path's functor
// (include/fl_user_objective.hpp: x and g as register arrays of the thread) has no meaning where the vectors live in HBM;
// here the objective is asked one ELEMENT PAIR at a time, in the order the built-in ones are evaluated, and the solver does
// the loads, the stores of g, the g.p / g.g terms and the fixed-order sums around it.  A plain class (no template
// parameters), named by the macro FL_USER_STREAM_OBJECTIVE before this header is included (fl_user_compile does that for
// n > 4096):
//   static constexpr bool NEIGHBOURS;   true: pair() reads x of OTHER elements through its `x` argument (the row is complete
//                                       and visible then: the trial point is stored in a pass of its own, a barrier follows)
//   __device__ void init(const fl::SolveArgs &A, int prob);          once per problem (A.d / A.b / A.user: the caller's data)
//   __device__ void pair(int e, int n, const double *x, double xa, double xb,   elements e (even) and e + 1 of problem `prob`
//                        double &ta, double &tb, double &ua, double &ub, double &ga, double &gb);
//        ta, tb: the elements' terms of the first sum s0; ua, ub: of the second sum s1; ga, gb: df/dx_e, df/dx_{e+1}.
//        An element >= n is padding (its x is 0): whatever pair() returns for it is replaced by zeros.
//   __device__ static double combine(double s0, double s1);          f from the two sums
// The terms are added thread by thread in slot order, then over the workgroup in the fixed tree order -- for a functor that
// restates a built-in objective, the built-in kernel's bits (tests/test_gpu_user_rtc.py).
#ifdef FL_USER_STREAM_OBJECTIVE
using BigUserObjective = FL_USER_STREAM_OBJECTIVE;
#else
struct BigUserObjective { // (placeholder: BigSolver<FL_OBJ_USER, .> is instantiated with a caller's class only)
    static constexpr bool NEIGHBOURS = false;
    __device__ void init(const SolveArgs &, int) {}
    __device__ void pair(int, int, const double *, double, double, double &ta, double &tb, double &ua, double &ub, double &ga, double &gb)
    {
        ta = tb = ua = ub = ga = gb = 0.0;
    }
    __device__ static double combine(double s0, double) { return s0; }
};
#endif
struct BigNoObjective {};

// AUG = 1 (reverse communication only, OBJ = FL_OBJ_EXTERNAL): the AugmentedLagrangian outer loop (NO.f90:2150-2185) around the
// solver with the CALLER's f, f', c, c' -- Solver<..., AUG>::take_external_aug and inner_finished restated over vectors in HBM.
template <int OBJ, int METHOD, int AUG = 0> struct BigSolver {
    static_assert(!AUG || OBJ == FL_OBJ_EXTERNAL, "beyond n = 4096 the augmented Lagrangian runs by reverse communication");
    static_assert(METHOD == FL_SOLVER_SD || METHOD == FL_SOLVER_CG || METHOD == FL_SOLVER_LBFGS ||
                      METHOD == FL_SOLVER_BFGS,
                  "vectors-in-HBM path: SD, CG, L-BFGS, BFGS (quasi-Newton updates only)");
    static constexpr int NW = 16, T = 1024;
    static constexpr bool NEEDS_G0 = (METHOD != FL_SOLVER_SD);
    static constexpr int ROWS = 4; // p, x0, g0, g per problem
    // LDS carve (doubles)
    static constexpr int L_RED = 0;
    static constexpr int L_RHO = L_RED + 2 * Reducer<NW>::NVMAX * NW;
    static constexpr int L_ALPHA = L_RHO + FL_MAX_MEMORY;
    // BFGS (n <= BF_MAX_N): inverse Hessian [n][npad] in HBM, rank-2 updates deferred exactly like
    // Solver::direction_bfgs_deferred (pending updates as vectors, folded every BF_DEFER-th iteration; oracle
    // update_form 100 + BF_DEFER); L_STAGE: s_l[j], q_l[j] of BF_FOLD_COLS columns while folding
    static constexpr int BF_DEFER = FL_BFGS_DEFER, BF_FOLD_COLS = 128, BF_MAX_SLOTS = 8, BF_MAX_N = BF_MAX_SLOTS * 2 * T;
    static constexpr int L_STAGE = L_ALPHA + FL_MAX_MEMORY;
    static constexpr int L_COOP = L_STAGE + (METHOD == FL_SOLVER_BFGS ? 2 * BF_DEFER * BF_FOLD_COLS + 4 : 0);
    // cooperative form (reverse communication only): the partial sums of up to COOP_MAX_GROUPS workgroups, staged for the
    // left-to-right addition
    static constexpr int COOP_MAX_GROUPS = 256;
    static constexpr int L_LAM = L_COOP + (METHOD != FL_SOLVER_BFGS ? COOP_MAX_GROUPS * Reducer<16>::NVMAX : 0); // lambda[FL_MAX_CONSTRAINTS]
    static constexpr int L_CX = L_LAM + (AUG ? FL_MAX_CONSTRAINTS : 0);                                           // c(x)[FL_MAX_CONSTRAINTS]
    static constexpr int LDS_TOTAL = L_CX + (AUG ? FL_MAX_CONSTRAINTS : 0);
    // the cooperative form needs every element touched by its owner alone: not BFGS (the dense H is folded across rows), not an
    // objective that reads its neighbours' x (written by another workgroup, possibly behind another XCD's L2)
    static constexpr bool COOP_OK = METHOD != FL_SOLVER_BFGS && !(OBJ == FL_OBJ_ROSENBROCK || (OBJ == FL_OBJ_USER && BigUserObjective::NEIGHBOURS));
    static constexpr int RCI_SCALARS = 48;
    static constexpr int UNI_LEVEL = (METHOD == FL_SOLVER_BFGS) ? 2 : FL_UNI_LEVEL; // 128 VGPRs per wave: BFGS pins the search to SGPRs

    const SolveArgs &A;
    double *lds;
    int prob, n, nslot, tid;
    size_t npad;
    bool ual; // user rows (stride n) are 16-byte aligned
    Reducer<NW> R;
    double *x;
    const double *dd, *bb;
    double *p, *x0, *g0, *g, *hist;
    // uniform scalars (names as in Solver)
    double fnew, gg, pp, phid, phidold, a;
    int iters, nf, ng, status, phase, pending;
    int recent, cnt;
    double yy_recent, rho_recent;
    int main_it, h_valid, ndef, h_ident; // BFGS (names as in Solver)
    double miu, cc;                      // augmented Lagrangian (names as in Solver)
    int outer_it, inner_iters_total;
    double a_id;
    LineSearch ls;
    typename pick_type<OBJ == FL_OBJ_USER, BigUserObjective, BigNoObjective>::type uo; // the caller's streaming functor
    static constexpr bool USER_NEIGHBOURS = OBJ == FL_OBJ_USER && BigUserObjective::NEIGHBOURS;
    static constexpr bool X_BARRIERS = OBJ == FL_OBJ_ROSENBROCK || USER_NEIGHBOURS; // other threads' x is read
    enum { PH_INIT = 0, PH_LS = 1, PH_DONE = 2 };
    // COOPERATIVE form (reverse communication, few problems of very large n): G workgroups share one problem.  Workgroup
    // wg owns the slots c_lo <= c < c_hi of every thread (a contiguous range of ceil(nslot / G) slots), runs the same
    // scalar machine as its siblings and meets them in every reduction: each sums its own slots in the usual order
    // (thread, wave tree, waves left to right), publishes the partial, and all add the G partials left to right -- the
    // oracle's tree order with `groups` (oracle/fl_oracle.c: tree_reduce).  G = 1: everything as before.
    int G, wg, c_lo, c_hi;
    double *coop_part;       // [2][G][NVMAX] partial sums of this problem's workgroups (two generations)
    unsigned *coop_counter;  // arrivals at this problem's barriers (zeroed by the host before every launch); word 1: gave up
    unsigned *coop_host_flag; // the handle's host-visible "a barrier gave up" word (pinned host memory), may be NULL
    unsigned coop_gen;

    __host__ __device__ static int slots_for(int n) { return ((n + 1) / 2 + T - 1) / T; }

    __device__ __forceinline__ BigSolver(const SolveArgs &A_, double *lds_, double *rows_all, int problem = -1)
        : A(A_), lds(lds_), prob(problem >= 0 ? problem : (int)blockIdx.x), n(A_.n), tid(threadIdx.x), R{lds_ + L_RED, 0}
    {
        nslot = slots_for(n);
        npad = (size_t)nslot * T * 2;
        ual = (n & 1) == 0;
        x = A.x + (size_t)prob * n;
        dd = A.d ? A.d + (size_t)prob * n : nullptr;
        bb = A.b ? A.b + (size_t)prob * n : nullptr;
        double *rows = rows_all + (size_t)prob * ROWS * npad;
        p = rows;
        x0 = rows + npad;
        g0 = rows + 2 * npad;
        g = rows + 3 * npad;
        hist = nullptr;
        if constexpr (METHOD == FL_SOLVER_LBFGS) hist = A.hist + (size_t)prob * (size_t)(2 * A.mem) * npad;
        if constexpr (METHOD == FL_SOLVER_BFGS) hist = A.hist + (size_t)prob * bfgs_rows(n) * npad; // H, pending s_l q_l, y
        G = 1;
        wg = 0;
        c_lo = 0;
        c_hi = nslot;
        coop_part = nullptr;
        coop_counter = nullptr;
        coop_host_flag = nullptr;
        coop_gen = 0;
    }
    __host__ __device__ static int coop_slots_per_group(int n, int groups) { return (slots_for(n) + groups - 1) / groups; }
    // the number of workgroups that really share a problem when `groups` are asked for (no empty ones)
    __host__ __device__ static int coop_groups(int n, int groups)
    {
        if (groups <= 1) return 1;
        const int per = coop_slots_per_group(n, groups);
        return (slots_for(n) + per - 1) / per;
    }
    __device__ __forceinline__ void set_cooperative(int groups, int group, double *part, unsigned *counter, unsigned *host_flag = nullptr)
    {
        G = groups;
        wg = group;
        const int per = (nslot + groups - 1) / groups;
        c_lo = group * per;
        c_hi = c_lo + per < nslot ? c_lo + per : nslot;
        coop_part = part;
        coop_counter = counter;
        coop_host_flag = host_flag;
    }
    // Every workgroup of the problem has arrived.  What crosses workgroups -- the partial sums and this counter -- moves by
    // device-scope atomics (they bypass the XCDs' private L2s: /opt/skills/guides/MI355X_MICROARCH.md, "cross-workgroup
    // hand-off", the flag / counter recipe); vector elements never cross (each is touched by its owner alone, and a kernel
    // boundary lies between two steps).  So the ordering needs no device-wide fence -- a __threadfence() here writes the
    // whole XCD's dirty L2 back: measured ~60 us per barrier, and an agent-scope RELEASE on the counter emits the same
    // write-back -- only this: thread 0's partial-sum stores (write-through, `sc1`) must have LEFT THE WAVE'S STORE QUEUE
    // before its arrival is counted.  A workgroup-scope fence does not wait for them (round 3 relied on it: the ISA had the
    // stores, s_barrier and the atomic add with no wait in between, so a sibling could see the count reached and read a
    // partial sum from two generations ago) -- the explicit s_waitcnt vmcnt(0) does (gfx9: stores count in vmcnt), and
    // tests/test_kernel_resources.py checks that it is in the code object between the stores and the add.  On the other
    // side the partial sums are fetched by agent-scope loads issued behind the workgroup barrier that follows the poll.
    // Bounded wait: should the workgroups of a problem ever not be resident together (fl_rci_create sizes the grid from the
    // occupancy query, but a GPU shared with another process can still break it) the wait ends, the problem's flag word
    // and the handle's host-visible word are set, every later barrier of the problem returns at once (the counter can
    // never catch up), and the next fl_rci_step / fl_rci_results reports FL_ERR_LAUNCH.
    __device__ __forceinline__ void coop_barrier()
    {
        __syncthreads();
        ++coop_gen; // (every thread counts the barriers: the generation picks the partial sums' buffer in reduce())
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the partial sums have reached memory
            __hip_atomic_fetch_add(coop_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = coop_gen * (unsigned)G;
            if (__hip_atomic_load(coop_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) { // (nobody gave up yet)
                unsigned spins = 0;
                while (__hip_atomic_load(coop_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if ((spins & 1023u) == 0u && __hip_atomic_load(coop_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
                        break; // a sibling gave up
                    if (spins > (1u << 22)) {
                        __hip_atomic_store(coop_counter + 1, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (coop_host_flag) __hip_atomic_store(coop_host_flag, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        __syncthreads();
    }
    template <int NV> __device__ __forceinline__ void reduce(double (&v)[NV])
    {
        R.run(v);
        if (G == 1) return;
        double *mine = coop_part + ((size_t)(coop_gen & 1) * G + wg) * Reducer<NW>::NVMAX;
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) __hip_atomic_store(mine + i, v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double *all = coop_part + (size_t)(coop_gen & 1) * G * Reducer<NW>::NVMAX;
        coop_barrier(); // (advances coop_gen: `all` was taken before)
        // thread w fetches workgroup w's partial sums (G loads in flight at once: one after the other they cost a trip to
        // memory each), then everybody adds them left to right from LDS
        double *stage = lds + L_COOP;
        if (tid < G) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                stage[tid * NV + i] = __hip_atomic_load(all + (size_t)tid * Reducer<NW>::NVMAX + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = stage[i];
            for (int w = 1; w < G; ++w) t = t + stage[w * NV + i];
            v[i] = t;
        }
        __syncthreads(); // (the stage is free again)
    }
    __host__ __device__ static size_t bfgs_rows(int n) { return (size_t)n + 2 * BF_DEFER + 1; }

    // ---- element pairs
    __device__ __forceinline__ int e_of(int c) const // (thread index behind an empty asm: see Geo::tid())
    {
        int t = tid;
        asm volatile("" : "+v"(t));
        return (c * T + t) << 1;
    }
    __device__ __forceinline__ void ldu(const double *row, int e, double &u, double &v) const
    {
        if (ual && e + 1 < n) {
            const double2 t = *reinterpret_cast<const double2 *>(row + e);
            u = t.x;
            v = t.y;
        } else {
            u = (e < n) ? row[e] : 0.0;
            v = (e + 1 < n) ? row[e + 1] : 0.0;
        }
    }
    __device__ __forceinline__ void stu(double *row, int e, double u, double v) const
    {
        if (ual && e + 1 < n) {
            *reinterpret_cast<double2 *>(row + e) = make_double2(u, v);
        } else {
            if (e < n) row[e] = u;
            if (e + 1 < n) row[e + 1] = v;
        }
    }
    __device__ __forceinline__ static void ldw(const double *row, int e, double &u, double &v)
    {
        const double2 t = *reinterpret_cast<const double2 *>(row + e);
        u = t.x;
        v = t.y;
    }
    __device__ __forceinline__ static void stw(double *row, int e, double u, double v)
    {
        *reinterpret_cast<double2 *>(row + e) = make_double2(u, v);
    }
    // running sum in the register path's order: first term of the thread starts it
    __device__ __forceinline__ void acc2(double &s, int c, double ta, double tb) const
    {
        s = (c == c_lo) ? ta : s + ta;
        s = s + tb;
    }

    // ---------------------------------------------------------------- setup
    __device__ __forceinline__ void init()
    {
        iters = nf = ng = 0;
        status = FL_STATUS_CONVERGED;
        recent = -1;
        cnt = 0;
        yy_recent = rho_recent = 0.0;
        fnew = gg = pp = phid = phidold = a = 0.0;
        main_it = h_valid = ndef = h_ident = 0;
        a_id = 0.0;
        phase = PH_INIT;
        pending = 0;
        miu = cc = 0.0;
        outer_it = inner_iters_total = 0;
        if constexpr (OBJ == FL_OBJ_USER) uo.init(A, prob);
        if constexpr (AUG) { // (lambda lives in LDS during a launch and in the caller's array between launches)
            miu = A.miu0 > 1.0 ? A.miu0 : 1.0; // miu=max(1d0,miu0)
            if (tid < A.aug_m) lds[L_LAM + tid] = A.lambda[(size_t)prob * A.aug_m + tid];
            __syncthreads();
        }
    }
    // the caller's terms for one element pair, padding forced to zero
    __device__ __forceinline__ void user_pair(int e, double xa, double xb, double &ta, double &tb, double &ua, double &ub, double &ga, double &gb)
    {
        if constexpr (OBJ == FL_OBJ_USER) {
            uo.pair(e, n, x, xa, xb, ta, tb, ua, ub, ga, gb);
            if (e >= n) ta = ua = ga = 0.0;
            if (e + 1 >= n) tb = ub = gb = 0.0;
        }
    }
    __device__ __forceinline__ void clear_rows() // first launch: p = 0 like the register path, padding defined
    {
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            stw(p, e, 0.0, 0.0);
            stw(x0, e, 0.0, 0.0);
            stw(g0, e, 0.0, 0.0);
            stw(g, e, 0.0, 0.0);
        }
    }
    __device__ __forceinline__ int start()
    {
        ls.zn = 0; // (must_stop() reads it before the first search begins)
        phase = PH_INIT;
        pending = FL_REQ_F | FL_REQ_G | FL_REQ_NOMOVE;
        return pending;
    }

    // ---------------------------------------------------------------- evaluation
    __device__ __forceinline__ void move(double at)
    {
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double xa, xb, pa, pb;
            ldw(x0, e, xa, xb);
            ldw(p, e, pa, pb);
            stu(x, e, xa + at * pa, xb + at * pb);
        }
    }
    // built-in objective at the x in the user array: f, g (stored), g.p, g.g
    __device__ __forceinline__ void evaluate(double &f, double &gp, double &ggo)
    {
        if constexpr (X_BARRIERS) __syncthreads(); // neighbours' x are written
        double r[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double xa, xb, ga, gb, ta, tb, ua = 0.0, ub = 0.0;
            ldu(x, e, xa, xb);
            if constexpr (OBJ == FL_OBJ_QUARTIC) { // test/test.f90:630-663
                const double a3 = xa * xa * xa, b3 = xb * xb * xb;
                ta = a3 * xa;
                tb = b3 * xb;
                ga = 4.0 * a3;
                gb = 4.0 * b3;
            } else if constexpr (OBJ == FL_OBJ_DIAGQUAD) {
                double da, db, ba, bbv;
                ldu(dd, e, da, db);
                ldu(bb, e, ba, bbv);
                const double dxa = da * xa, dxb = db * xb;
                ta = dxa * xa;
                tb = dxb * xb;
                ua = ba * xa;
                ub = bbv * xb;
                ga = dxa - ba;
                gb = dxb - bbv;
            } else if constexpr (OBJ == FL_OBJ_ROSENBROCK) { // as Objective<FL_OBJ_ROSENBROCK>::eval
                const double xl = (e >= 1 && e - 1 < n) ? x[e - 1] : 0.0;
                const double xr = (e + 2 < n) ? x[e + 2] : 0.0;
                const double ul = xa - xl * xl, um = xb - xa * xa, ur = xr - xb * xb;
                const double va = 1.0 - xa, vb = 1.0 - xb;
                const double A_a = (e >= 1) ? 200.0 * ul : 0.0;
                const double A_b = 200.0 * um;
                ta = tb = ga = gb = 0.0;
                if (e <= n - 2) {
                    ta = 100.0 * (um * um) + va * va;
                    ga = A_a - 400.0 * xa * um - 2.0 * va;
                } else if (e == n - 1) {
                    ga = A_a;
                }
                if (e + 1 <= n - 2) {
                    tb = 100.0 * (ur * ur) + vb * vb;
                    gb = A_b - 400.0 * xb * ur - 2.0 * vb;
                } else if (e + 1 == n - 1) {
                    gb = A_b;
                }
            } else if constexpr (OBJ == FL_OBJ_USER) {
                user_pair(e, xa, xb, ta, tb, ua, ub, ga, gb);
            } else {
                ta = tb = ga = gb = 0.0;
            }
            stw(g, e, ga, gb);
            double pa, pb;
            ldw(p, e, pa, pb);
            acc2(r[0], c, ta, tb);
            if constexpr (OBJ == FL_OBJ_DIAGQUAD || OBJ == FL_OBJ_USER) acc2(r[1], c, ua, ub);
            acc2(r[2], c, ga * pa, gb * pb);
            acc2(r[3], c, ga * ga, gb * gb);
        }
        reduce(r);
        f = uni(combine_sums(r[0], r[1]));
        gp = uni(r[2]);
        ggo = uni(r[3]);
        if constexpr (X_BARRIERS) __syncthreads(); // all neighbour reads done before x moves again
    }
    __device__ __forceinline__ static double combine_sums(double s0, double s1)
    {
        if constexpr (OBJ == FL_OBJ_USER) return BigUserObjective::combine(s0, s1);
        else return Objective<OBJ, 1, 2>::combine(s0, s1);
    }
    // a trial in ONE pass: x = x0 + at p formed, stored and evaluated together (Rosenbrock's neighbours are formed
    // from x0, p the same way -- bitwise the neighbour thread's x -- so no barrier is needed)
    __device__ __forceinline__ void move_evaluate(double at, double &f, double &gp, double &ggo)
    {
        if constexpr (USER_NEIGHBOURS) { // (the caller's functor reads x itself: the trial point in a pass of its own)
            move(at);
            evaluate(f, gp, ggo);
            return;
        }
        if constexpr (OBJ == FL_OBJ_ROSENBROCK) __syncthreads(); // x0 / p rows of the neighbours are complete
        double r[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double oa, ob, pa, pb, ga, gb, ta, tb, ua = 0.0, ub = 0.0;
            ldw(x0, e, oa, ob);
            ldw(p, e, pa, pb);
            const double xa = oa + at * pa, xb = ob + at * pb;
            stu(x, e, xa, xb);
            if constexpr (OBJ == FL_OBJ_QUARTIC) {
                const double a3 = xa * xa * xa, b3 = xb * xb * xb;
                ta = a3 * xa;
                tb = b3 * xb;
                ga = 4.0 * a3;
                gb = 4.0 * b3;
            } else if constexpr (OBJ == FL_OBJ_DIAGQUAD) {
                double da, db, ba, bbv;
                ldu(dd, e, da, db);
                ldu(bb, e, ba, bbv);
                const double dxa = da * xa, dxb = db * xb;
                ta = dxa * xa;
                tb = dxb * xb;
                ua = ba * xa;
                ub = bbv * xb;
                ga = dxa - ba;
                gb = dxb - bbv;
            } else if constexpr (OBJ == FL_OBJ_ROSENBROCK) {
                const double xl = (e >= 1 && e - 1 < n) ? x0[e - 1] + at * p[e - 1] : 0.0;
                const double xr = (e + 2 < n) ? x0[e + 2] + at * p[e + 2] : 0.0;
                const double ul = xa - xl * xl, um = xb - xa * xa, ur = xr - xb * xb;
                const double va = 1.0 - xa, vb = 1.0 - xb;
                const double A_a = (e >= 1) ? 200.0 * ul : 0.0;
                const double A_b = 200.0 * um;
                ta = tb = ga = gb = 0.0;
                if (e <= n - 2) {
                    ta = 100.0 * (um * um) + va * va;
                    ga = A_a - 400.0 * xa * um - 2.0 * va;
                } else if (e == n - 1) {
                    ga = A_a;
                }
                if (e + 1 <= n - 2) {
                    tb = 100.0 * (ur * ur) + vb * vb;
                    gb = A_b - 400.0 * xb * ur - 2.0 * vb;
                } else if (e + 1 == n - 1) {
                    gb = A_b;
                }
            } else if constexpr (OBJ == FL_OBJ_USER) {
                user_pair(e, xa, xb, ta, tb, ua, ub, ga, gb);
            } else {
                ta = tb = ga = gb = 0.0;
            }
            stw(g, e, ga, gb);
            acc2(r[0], c, ta, tb);
            if constexpr (OBJ == FL_OBJ_DIAGQUAD || OBJ == FL_OBJ_USER) acc2(r[1], c, ua, ub);
            acc2(r[2], c, ga * pa, gb * pb);
            acc2(r[3], c, ga * ga, gb * gb);
        }
        reduce(r);
        f = uni(combine_sums(r[0], r[1]));
        gp = uni(r[2]);
        ggo = uni(r[3]);
    }
    // reverse communication: the caller's gradient [n] -> g, with g.p and g.g
    __device__ __forceinline__ void take_gradient(const double *g_user, double &gp, double &ggo)
    {
        double r[2] = {0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double ga, gb, pa, pb;
            ldu(g_user, e, ga, gb);
            stw(g, e, ga, gb);
            ldw(p, e, pa, pb);
            acc2(r[0], c, ga * pa, gb * pb);
            acc2(r[1], c, ga * ga, gb * gb);
        }
        reduce(r);
        gp = uni(r[0]);
        ggo = uni(r[1]);
    }

    // reverse communication, augmented Lagrangian: the caller's f, grad f [n], c [m], cd [m][n] -> L = f - lambda.c + miu/2 c.c and
    // grad L = grad f + cd^T (miu c - lambda), the sum over the constraints in order (L, Ld: NO.f90:2193-2228;
    // Solver::take_external_aug), with grad L . p and |grad L|^2
    __device__ __forceinline__ void take_external_aug(double fuser, bool have_f, bool have_g, const double *g_user, const double *c_user,
                                                      const double *cd_user, double &f, double &gp, double &ggo)
    {
        const int m = A.aug_m;
        double *cxs = lds + L_CX;
        __syncthreads(); // readers of the previous c(x) are done
        if (tid < m) cxs[tid] = c_user[tid];
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
            double r[2] = {0.0, 0.0};
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double ga, gb, pa, pb, ta = 0.0, tb = 0.0;
                ldu(g_user, e, ga, gb);
                for (int j = 0; j < m; ++j) {
                    const double v = miu * cxs[j] - lds[L_LAM + j];
                    double ra, rb;
                    ldu(cd_user + (size_t)j * n, e, ra, rb);
                    ta = ta + ra * v;
                    tb = tb + rb * v;
                }
                ga = ga + ta;
                gb = gb + tb;
                stw(g, e, ga, gb);
                ldw(p, e, pa, pb);
                acc2(r[0], c, ga * pa, gb * pb);
                acc2(r[1], c, ga * ga, gb * gb);
            }
            reduce(r);
            gp = uni(r[0]);
            ggo = uni(r[1]);
        }
    }

    // ---------------------------------------------------------------- machine (Solver::advance and friends)
    // (an objective that is not a number ends the problem: see Solver::not_finite)
    __device__ __forceinline__ bool not_finite(double fv) const { return (pending & FL_REQ_F) && fv != fv; }
    __device__ __forceinline__ bool must_stop(double fv) const { return not_finite(fv) || ls.stalled(); } // (Solver::must_stop)
    __device__ __forceinline__ void stop_not_finite()
    {
        if (ls.stalled()) {
            status = FL_STATUS_STALLED;
        } else {
            status = FL_STATUS_NOT_FINITE;
            fnew = __builtin_nan("");
        }
        phase = PH_DONE;
        pending = 0;
    }
    __device__ __forceinline__ int advance(double fv, double pv, double gg_new)
    {
        nf += (pending & FL_REQ_F) ? 1 : 0;
        ng += (pending & FL_REQ_G) ? 1 : 0;
        int rq;
        if (phase == PH_INIT) {
            rq = after_init(fv, gg_new);
        } else {
            gg = gg_new;
            rq = __builtin_amdgcn_readfirstlane(ls.step(fv, pv));
            ls.template uniformize<UNI_LEVEL>();
            if (rq == 0) rq = after_linesearch();
        }
        pending = rq;
        return rq;
    }
    __device__ __forceinline__ double request_point() const { return ls.a_eval; }
    __device__ __forceinline__ int max_linesearches() const
    {
        if constexpr (METHOD == FL_SOLVER_LBFGS) return A.mem + A.maxit;
        if constexpr (METHOD == FL_SOLVER_BFGS) return 0x7fffffff; // counted by main_it in begin_linesearch
        return A.maxit;
    }
    __device__ __forceinline__ int finished() // (the inner solve is over: Solver::inner_finished)
    {
        if constexpr (AUG) {
            const int m = A.aug_m;
            const double *cxs = lds + L_CX; // c(x) of the last evaluation = c at the current x
            double c2 = 0.0;
            for (int j = 0; j < m; ++j) c2 = c2 + cxs[j] * cxs[j];
            cc = uni(c2);
            ++outer_it;
            inner_iters_total += iters;
            iters = 0;
            if (c2 < A.precision * A.precision) { // if(dot_product(cx,cx)<tolsq) exit
                status = FL_STATUS_CONVERGED;
                phase = PH_DONE;
                return 0;
            }
            __syncthreads();
            if (tid < m) lds[L_LAM + tid] = lds[L_LAM + tid] - miu * cxs[tid]; // lambda=lambda-miu*cx
            __syncthreads();
            miu = uni(miu * A.incr); // miu=miu*incrmt
            if (outer_it >= A.maxit) {
                status = FL_STATUS_MAXIT;
                phase = PH_DONE;
                return 0;
            }
            recent = -1; // a fresh inner solve from the current x: it starts with an evaluation of L, L'
            cnt = 0;
            main_it = 0;
            h_valid = 0;
            ndef = 0;
            h_ident = 0;
            phase = PH_INIT;
            return FL_REQ_F | FL_REQ_G | FL_REQ_NOMOVE;
        }
        phase = PH_DONE;
        return 0;
    }
    __device__ __forceinline__ void neg_gradient_direction() // p=-fdnew
    {
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double ga, gb;
            ldw(g, e, ga, gb);
            stw(p, e, -ga, -gb);
        }
    }
    __device__ __forceinline__ int begin_linesearch()
    {
        int fused = A.fused;
        if constexpr (AUG) fused = 1; // AugmentedLagrangian always passes f_fd=L_Ld (NO.f90:2153, 2161)
        if constexpr (METHOD == FL_SOLVER_LBFGS) fused = fused && (iters >= A.mem); // NO.f90:448-460, 486-498
        if constexpr (METHOD == FL_SOLVER_BFGS) { // main loop do iIteration=1,maxit (NO.f90:717-929); the first step precedes it
            if (h_valid) {
                if (main_it >= A.maxit) {
                    status = FL_STATUS_MAXIT;
                    return finished();
                }
                ++main_it;
            }
            fused = fused && h_valid;
        }
        for (int c = c_lo; c < c_hi; ++c) { // xold=x; fdold=fdnew
            const int e = e_of(c);
            double u, v;
            ldu(x, e, u, v);
            stw(x0, e, u, v);
            if constexpr (NEEDS_G0) {
                ldw(g, e, u, v);
                stw(g0, e, u, v);
            }
        }
        phidold = phid;
        const int strong = (METHOD == FL_SOLVER_CG && A.cg_method == FL_CG_PR) ? 1 : A.strong;
        phase = PH_LS;
        const int rq = __builtin_amdgcn_readfirstlane(ls.begin(strong, fused, A.c1, A.c2, A.incr, a, fnew, phid));
        ls.template uniformize<UNI_LEVEL>();
        return rq;
    }
    __device__ __forceinline__ int after_init(double f, double gg0)
    {
        fnew = f;
        gg = gg0;
        neg_gradient_direction();
        phid = -gg;
        pp = gg;
        status = FL_STATUS_CONVERGED;
        if (gg < A.tol) return finished();
        a = uni((fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(gg));
        status = FL_STATUS_MAXIT;
        if (max_linesearches() <= 0) return finished();
        return begin_linesearch();
    }
    __device__ __forceinline__ int after_linesearch()
    {
        a = ls.a;
        fnew = ls.fx;
        ++iters;
        if (gg < A.tol) {
            status = FL_STATUS_CONVERGED;
            return finished();
        }
        if (pp * a * a < A.minstep) {
            status = FL_STATUS_STEP_CONVERGED;
            return finished();
        }
        if (iters >= max_linesearches()) {
            status = FL_STATUS_MAXIT;
            return finished();
        }
        if constexpr (METHOD == FL_SOLVER_SD) { // NO.f90:185-186
            neg_gradient_direction();
            phid = -gg;
            pp = gg;
            a = a * phidold / phid;
        } else if constexpr (METHOD == FL_SOLVER_CG) {
            direction_cg();
        } else if constexpr (METHOD == FL_SOLVER_BFGS) {
            direction_bfgs();
            h_valid = 1;
        } else {
            direction_lbfgs();
        }
        phid = uni(phid);
        pp = uni(pp);
        a = uni(a);
        return begin_linesearch();
    }

    // ---------------------------------------------------------------- directions
    __device__ __forceinline__ void direction_cg()
    {
        double beta;
        if (A.cg_method == FL_CG_DY) { // NO.f90:366
            double q[1] = {0.0};
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double ga, gb, oa, ob, pa, pb;
                ldw(g, e, ga, gb);
                ldw(g0, e, oa, ob);
                ldw(p, e, pa, pb);
                acc2(q[0], c, (ga - oa) * pa, (gb - ob) * pb);
            }
            reduce(q);
            beta = gg / q[0];
        } else { // NO.f90:387
            double q[2] = {0.0, 0.0};
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double ga, gb, oa, ob;
                ldw(g, e, ga, gb);
                ldw(g0, e, oa, ob);
                acc2(q[0], c, ga * (ga - oa), gb * (gb - ob));
                acc2(q[1], c, oa * oa, ob * ob);
            }
            reduce(q);
            beta = q[0] / q[1];
        }
        double q2[2] = {0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double ga, gb, pa, pb;
            ldw(g, e, ga, gb);
            ldw(p, e, pa, pb);
            pa = -ga + beta * pa;
            pb = -gb + beta * pb;
            stw(p, e, pa, pb);
            acc2(q2[0], c, ga * pa, gb * pb);
            acc2(q2[1], c, pa * pa, pb * pb);
        }
        reduce(q2);
        phid = q2[0];
        pp = q2[1];
        if (phid > 0.0) { // NO.f90:368-370
            neg_gradient_direction();
            phid = -gg;
            pp = gg;
        }
        a = a * phidold / phid;
    }

    __device__ __forceinline__ void direction_lbfgs()
    {
        double *rho_s = lds + L_RHO, *alpha_s = lds + L_ALPHA;
        const int mem = A.mem;
        recent = (recent + 1 == mem) ? 0 : recent + 1;
        if (cnt < mem) ++cnt;
        auto slot_of = [&](int j) {
            int s = recent - j;
            return s < 0 ? s + mem : s;
        };
        auto srow = [&](int j) { return hist + (size_t)(2 * slot_of(j)) * npad; };
        // After() (NO.f90:609-624) with the first dot product of Before() riding along: p starts as g
        double r[3] = {0.0, 0.0, 0.0};
        {
            double *s0 = srow(0), *y0 = s0 + npad;
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double xa, xb, oa, ob, ga, gb, ha, hb;
                ldu(x, e, xa, xb);
                ldw(x0, e, oa, ob);
                ldw(g, e, ga, gb);
                ldw(g0, e, ha, hb);
                const double sa = xa - oa, sb = xb - ob, ya = ga - ha, yb = gb - hb;
                stw(s0, e, sa, sb);
                stw(y0, e, ya, yb);
                acc2(r[0], c, ya * sa, yb * sb);
                acc2(r[1], c, ya * ya, yb * yb);
                acc2(r[2], c, sa * ga, sb * gb);
            }
        }
        reduce(r);
        if (tid == 0) rho_s[recent] = 1.0 / r[0];
        rho_recent = uni(1.0 / r[0]);
        yy_recent = uni(r[1]);
        // Before() (NO.f90:586-608): newest -> oldest
        double al = rho_recent * r[2];
        if (tid == 0) alpha_s[recent] = al;
        for (int j = 0; j < cnt; ++j) { // p = p - alpha_j y_j, next dot product in the same pass
            const double *yj = srow(j) + npad;
            const bool last = (j + 1 == cnt);
            const double *nxt = last ? yj : srow(j + 1); // last: the first dot of the way up is y_{cnt-1}.p
            const double *src = (j == 0) ? g : p;
            double q[1] = {0.0};
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double pa, pb, ya, yb, na, nb;
                ldw(src, e, pa, pb);
                ldw(yj, e, ya, yb);
                pa = pa - al * ya;
                pb = pb - al * yb;
                if (last) { // p=p/rho(recent)/(y.y)
                    pa = pa / rho_recent / yy_recent;
                    pb = pb / rho_recent / yy_recent;
                    na = ya;
                    nb = yb;
                } else {
                    ldw(nxt, e, na, nb);
                }
                stw(p, e, pa, pb);
                acc2(q[0], c, na * pa, nb * pb);
            }
            reduce(q);
            if (!last) {
                const int sl = slot_of(j + 1);
                al = rho_s[sl] * q[0];
                if (tid == 0) alpha_s[sl] = al;
            } else {
                al = q[0]; // carries y_{cnt-1}.p into the way up
            }
        }
        __syncthreads(); // alpha_s complete
        double ydotp = al;
        double rr[2] = {0.0, 0.0};
        for (int j = cnt - 1; j >= 0; --j) { // p = p + (alpha_j - rho_j y_j.p) s_j
            const int sl = slot_of(j);
            const double co = alpha_s[sl] - rho_s[sl] * ydotp;
            const double *sj = srow(j);
            const bool last = (j == 0);
            const double *nxt = last ? g : srow(j - 1) + npad;
            double q[1] = {0.0};
            rr[0] = rr[1] = 0.0;
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double pa, pb, sa, sb, na, nb;
                ldw(p, e, pa, pb);
                ldw(sj, e, sa, sb);
                ldw(nxt, e, na, nb);
                pa = pa + co * sa;
                pb = pb + co * sb;
                if (last) { // p=-p; phidnew=dot_product(fdnew,p)
                    pa = -pa;
                    pb = -pb;
                    acc2(rr[0], c, na * pa, nb * pb);
                    acc2(rr[1], c, pa * pa, pb * pb);
                } else {
                    acc2(q[0], c, na * pa, nb * pb);
                }
                stw(p, e, pa, pb);
            }
            if (last) {
                reduce(rr);
            } else {
                reduce(q);
                ydotp = q[0];
            }
        }
        phid = rr[0];
        pp = rr[1];
        a = 1.0;
    }

    // ---------------------------------------------------------------- BFGS, deferred rank-2 updates (NO.f90:996-1015, 711-715)
    __device__ __forceinline__ double *def_row(int r) const { return hist + ((size_t)n + r) * npad; }
    __device__ __forceinline__ void direction_bfgs()
    {
        double *H = hist, *yrow = def_row(2 * BF_DEFER);
        double *drho = lds + L_RHO, *dcs = drho + BF_DEFER; // rho_l, cs_l of the pending updates
        // s, y (y also as a row: the matvec reads y_j per column)
        double r1[1] = {0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double xa, xb, oa, ob, ga, gb, ha, hb;
            ldu(x, e, xa, xb);
            ldw(x0, e, oa, ob);
            ldw(g, e, ga, gb);
            ldw(g0, e, ha, hb);
            const double ya = ga - ha, yb = gb - hb;
            stw(yrow, e, ya, yb);
            acc2(r1[0], c, ya * (xa - oa), yb * (xb - ob));
        }
        reduce(r1);
        const double rho = uni(1.0 / r1[0]);
        if (!h_valid) {
            ndef = 0;
            h_ident = 1;
            a_id = a;
        }
        // q = H_cur y and w = H_cur g are kept as ROWS like everything else on this path (q in the row it will occupy
        // as q_l of the new pending update, w in p's row, which is free until the direction is rebuilt below): with
        // 1024 threads a wave has 128 VGPRs, and 8 slots of q and w in registers (64 of them) next to the column
        // loads spilled 240-310 VGPRs (1200 in the reverse-communication kernel).
        double *Sn = def_row(2 * ndef), *Qn = def_row(2 * ndef + 1), *wrow = p;
        if (h_ident) {
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double ya, yb, ga, gb;
                ldw(yrow, e, ya, yb);
                ldw(g, e, ga, gb);
                stw(Qn, e, a_id * ya, a_id * yb);
                stw(wrow, e, a_id * ga, a_id * gb);
            }
        } else {
            __syncthreads(); // y row complete
            // one read pass over H: q = H y, w = H g.  Two slots of rows at a time (every column is read in
            // nslot / 2 pieces of 2 KiB per wave), four columns in flight; thread i sums over j in order.
            constexpr int SG = 2, JU = 4;
            for (int c0 = 0; c0 < nslot; c0 += SG) {
                double q[2 * SG], w[2 * SG];
#pragma unroll
                for (int k = 0; k < 2 * SG; ++k) q[k] = w[k] = 0.0;
                for (int j = 0; j < n; j += JU) {
                    double h[JU][2 * SG];
#pragma unroll
                    for (int u = 0; u < JU; ++u) {
                        const double *col = H + (size_t)(j + u < n ? j + u : j) * npad;
#pragma unroll
                        for (int cc = 0; cc < SG; ++cc)
                            if (c0 + cc < nslot) ldw(col, e_of(c0 + cc), h[u][2 * cc], h[u][2 * cc + 1]);
                    }
#pragma unroll
                    for (int u = 0; u < JU; ++u) {
                        if (j + u < n) {
                            const double yj = yrow[j + u], gj = g[j + u];
#pragma unroll
                            for (int cc = 0; cc < SG; ++cc) {
                                if (c0 + cc < nslot) {
                                    q[2 * cc] = q[2 * cc] + h[u][2 * cc] * yj;
                                    q[2 * cc + 1] = q[2 * cc + 1] + h[u][2 * cc + 1] * yj;
                                    w[2 * cc] = w[2 * cc] + h[u][2 * cc] * gj;
                                    w[2 * cc + 1] = w[2 * cc + 1] + h[u][2 * cc + 1] * gj;
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < SG; ++cc) {
                    if (c0 + cc < nslot) {
                        stw(Qn, e_of(c0 + cc), q[2 * cc], q[2 * cc + 1]);
                        stw(wrow, e_of(c0 + cc), w[2 * cc], w[2 * cc + 1]);
                    }
                }
            }
        }
        for (int l = 0; l < ndef; ++l) { // corrections of the pending updates, oldest first
            const double *S = def_row(2 * l), *Q = def_row(2 * l + 1);
            double r[4] = {0.0, 0.0, 0.0, 0.0};
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double sa, sb, qa, qb, ya, yb, ga, gb;
                ldw(S, e, sa, sb);
                ldw(Q, e, qa, qb);
                ldw(yrow, e, ya, yb);
                ldw(g, e, ga, gb);
                acc2(r[0], c, sa * ya, sb * yb);
                acc2(r[1], c, qa * ya, qb * yb);
                acc2(r[2], c, sa * ga, sb * gb);
                acc2(r[3], c, qa * ga, qb * gb);
            }
            reduce(r);
            const double rl = drho[l], cl = dcs[l];
            for (int c = c_lo; c < c_hi; ++c) {
                const int e = e_of(c);
                double sa, sb, qa, qb, q0, q1, w0, w1;
                ldw(S, e, sa, sb);
                ldw(Q, e, qa, qb);
                ldw(Qn, e, q0, q1);
                ldw(wrow, e, w0, w1);
                const double rqa = rl * qa, rqb = rl * qb, rsa = rl * sa, rsb = rl * sb, cca = cl * sa, ccb = cl * sb;
                stw(Qn, e, q0 - rqa * r[0] - rsa * r[1] + cca * r[0], q1 - rqb * r[0] - rsb * r[1] + ccb * r[0]);
                stw(wrow, e, w0 - rqa * r[2] - rsa * r[3] + cca * r[2], w1 - rqb * r[2] - rsb * r[3] + ccb * r[2]);
            }
        }
        double r3[3] = {0.0, 0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double ya, yb, ga, gb, xa, xb, oa, ob, q0, q1;
            ldw(yrow, e, ya, yb);
            ldw(g, e, ga, gb);
            ldu(x, e, xa, xb);
            ldw(x0, e, oa, ob);
            ldw(Qn, e, q0, q1);
            acc2(r3[0], c, ya * q0, yb * q1);
            acc2(r3[1], c, (xa - oa) * ga, (xb - ob) * gb);
            acc2(r3[2], c, q0 * ga, q1 * gb);
        }
        reduce(r3);
        const double cs = uni(rho * rho * r3[0] + rho);
        double r4[2] = {0.0, 0.0};
        for (int c = c_lo; c < c_hi; ++c) {
            const int e = e_of(c);
            double ga, gb, xa, xb, oa, ob, q0, q1, w0, w1;
            ldw(g, e, ga, gb);
            ldu(x, e, xa, xb);
            ldw(x0, e, oa, ob);
            ldw(Qn, e, q0, q1);
            ldw(wrow, e, w0, w1);
            const double sa = xa - oa, sb = xb - ob;
            const double pa = -(w0 - (rho * q0) * r3[1] - (rho * sa) * r3[2] + (cs * sa) * r3[1]);
            const double pb = -(w1 - (rho * q1) * r3[1] - (rho * sb) * r3[2] + (cs * sb) * r3[1]);
            stw(p, e, pa, pb);
            stw(Sn, e, sa, sb);
            acc2(r4[0], c, ga * pa, gb * pb);
            acc2(r4[1], c, pa * pa, pb * pb);
        }
        if (tid == 0) {
            drho[ndef] = rho;
            dcs[ndef] = cs;
        }
        ++ndef;
        reduce(r4); // (its barrier also publishes rho_l, cs_l and the new rows)
        phid = uni(r4[0]); // finished -- two scalars -- before the fold: left pending, the reduction's 32 partial sums
        pp = uni(r4[1]);   // stay in vector registers across it
        a = 1.0;
        if (ndef == BF_DEFER) bfgs_fold();
    }
    // H <- H with the pending updates applied in order: one ELEMENT row per thread and sweep (the row factors of
    // the eight pending updates for one row are 24 doubles; for a 16-byte pair they were 96 VGPRs of the 128)
    __device__ __forceinline__ void bfgs_fold()
    {
        constexpr int J = BF_DEFER, CB = BF_FOLD_COLS;
        double *H = hist, *stage = lds + L_STAGE;
        const double *drho = lds + L_RHO, *dcs = drho + J;
        __syncthreads();
        // (one element per thread and sweep, CONTIGUOUS across the workgroup -- e = c2 * 1024 + tid, not the pair layout of the
        //  vector passes: a sweep over a column then touches whole 64-byte lines once instead of half of each in two sweeps.  Any
        //  thread may fold any element: the barriers at both ends of the fold separate it from the passes that own their pairs)
        for (int c2 = 0; c2 < 2 * nslot; ++c2) {
            int tl_ = tid;
            asm volatile("" : "+v"(tl_));
            const int e = c2 * T + tl_;
            double rq[J], rs[J], cf[J];
#pragma unroll
            for (int l = 0; l < J; ++l) {
                const double sv = def_row(2 * l)[e], qv = def_row(2 * l + 1)[e];
                const double rl = drho[l], cl = dcs[l];
                rq[l] = rl * qv;
                rs[l] = rl * sv;
                cf[l] = cl * sv;
            }
            for (int jb = 0; jb < n; jb += CB) {
                __syncthreads();
                for (int i = tid; i < 2 * J * CB; i += T) {
                    const int row = i / CB, col = i - row * CB;
                    stage[i] = (jb + col < n) ? def_row(row)[jb + col] : 0.0;
                }
                __syncthreads();
                const int jend = (n - jb < CB) ? n - jb : CB;
#pragma unroll 2
                for (int jj = 0; jj < jend; ++jj) {
                    const int j = jb + jj;
                    double *hp = H + (size_t)j * npad + e;
                    double ha = h_ident ? ((e == j) ? a_id : 0.0) : *hp;
#pragma unroll
                    for (int l = 0; l < J; ++l) {
                        const double sj = stage[(2 * l) * CB + jj], qj = stage[(2 * l + 1) * CB + jj];
                        ha = ha - rq[l] * sj - rs[l] * qj + cf[l] * sj;
                    }
                    *hp = ha;
                }
            }
        }
        __syncthreads();
        h_ident = 0;
        ndef = 0;
    }

    // ---------------------------------------------------------------- reverse communication: park / resume
    // (ownership rules of the parked state: fl_device.hpp, above Solver::save -- rule 3 is the barrier that ends load())
    __device__ __forceinline__ void save(double *sc, double *rho, double fv_c, double pv_c)
    {
        if (wg != 0) return; // cooperative form: every workgroup holds the same scalars; the first one parks them
        if constexpr (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_BFGS) { // rho ring / rho_l, cs_l of pending updates
            __syncthreads();
            if (tid < FL_MAX_MEMORY) rho[tid] = lds[L_RHO + tid];
        }
        if (tid == 0) {
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
            *iq++ = main_it; *iq++ = h_valid; *iq++ = ndef; *iq++ = h_ident;
            sc[29] = a_id;
            if constexpr (AUG) { // the outer loop (the multipliers themselves: A.lambda, below)
                sc[40] = miu;
                sc[41] = cc;
                int *aq = reinterpret_cast<int *>(sc + 42);
                aq[0] = outer_it;
                aq[1] = inner_iters_total;
            }
        }
        if constexpr (AUG) {
            __syncthreads();
            if (tid < A.aug_m) A.lambda[(size_t)prob * A.aug_m + tid] = lds[L_LAM + tid];
        }
    }
    __device__ __forceinline__ void load(const double *sc, const double *rho, double &fv_c, double &pv_c)
    {
        if constexpr (METHOD == FL_SOLVER_LBFGS || METHOD == FL_SOLVER_BFGS) {
            if (tid < FL_MAX_MEMORY) lds[L_RHO + tid] = rho[tid];
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
        main_it = *iq++; h_valid = *iq++; ndef = *iq++; h_ident = *iq++;
        a_id = sc[29];
        if constexpr (AUG) {
            miu = uni(sc[40]);
            cc = uni(sc[41]);
            const int *aq = reinterpret_cast<const int *>(sc + 42);
            outer_it = __builtin_amdgcn_readfirstlane(aq[0]);
            inner_iters_total = __builtin_amdgcn_readfirstlane(aq[1]);
        }
        pin_scalars();
        // every wave has read the parked scalars before thread 0 may overwrite them in save(): a step that only
        // takes an objective value has no other barrier (cooperative form: every workgroup of the problem has)
        // (cooperative form: the step writes the OTHER copy of the parked state -- rci_step_big_kernel -- so there is
        // nothing to protect across workgroups)
        __syncthreads();
    }

    // the parked scalars come back as one copy per lane: pin them to scalar registers (with 1024 threads a wave has
    // 128 VGPRs, and ~35 doubles of machine state would take 70 of them)
    __device__ __forceinline__ void pin_scalars()
    {
        fnew = uni(fnew); gg = uni(gg); pp = uni(pp); phid = uni(phid); phidold = uni(phidold); a = uni(a);
        yy_recent = uni(yy_recent); rho_recent = uni(rho_recent); a_id = uni(a_id);
        ls.template uniformize<2>();
        ls.fx0 = uni(ls.fx0);
        iters = __builtin_amdgcn_readfirstlane(iters); nf = __builtin_amdgcn_readfirstlane(nf);
        ng = __builtin_amdgcn_readfirstlane(ng); status = __builtin_amdgcn_readfirstlane(status);
        phase = __builtin_amdgcn_readfirstlane(phase); pending = __builtin_amdgcn_readfirstlane(pending);
        recent = __builtin_amdgcn_readfirstlane(recent); cnt = __builtin_amdgcn_readfirstlane(cnt);
        main_it = __builtin_amdgcn_readfirstlane(main_it); h_valid = __builtin_amdgcn_readfirstlane(h_valid);
        ndef = __builtin_amdgcn_readfirstlane(ndef); h_ident = __builtin_amdgcn_readfirstlane(h_ident);
    }

    __device__ __forceinline__ void finish() // x already holds the last evaluated point
    {
        if (tid == 0 && wg == 0) {
            // (fused cooperative form: a barrier that gave up waiting -- the problem's workgroups were not resident together --
            // leaves sums that mean nothing: the problem is reported as not solved)
            if (G > 1 && __hip_atomic_load(coop_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) status = FL_STATUS_NOT_SOLVED;
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

// n > 4096: vectors in HBM, one workgroup of 1024 threads per problem.
// groups > 1: the COOPERATIVE form of the fused solve (round 4) -- `groups` workgroups share one problem: blockIdx.x = problem *
// groups + group, each owns a contiguous range of every thread's slots, all run the same scalar machine and meet in every
// reduction (BigSolver::reduce: partial sums + a grid barrier of the problem's workgroups).  For FEW problems of very large n
// -- the reference's callers typically solve one problem of any dim -- which otherwise occupy one CU each: the whole solve stays
// one launch, a trial costs its HBM traffic at the chip's bandwidth plus one barrier.  The sums are those of the oracle's tree
// order with `groups` (fl_cooperative_groups_for reports it).  The host sizes the grid so that all workgroups are resident
// (occupancy query); should they not be, the barrier's bounded wait ends and the problem reports FL_STATUS_NOT_SOLVED.
template <int OBJ, int METHOD>
__global__ __launch_bounds__(1024) void fl_big_solve_kernel(SolveArgs A, double *rows, int groups, double *coop_part, unsigned *coop_counter)
{
    using S = BigSolver<OBJ, METHOD>;
    __shared__ __attribute__((aligned(16))) double lds[S::LDS_TOTAL];
    const int prob = groups > 1 ? (int)blockIdx.x / groups : (int)blockIdx.x;
    S s(A, lds, rows, prob);
    if constexpr (S::COOP_OK) {
        if (groups > 1)
            s.set_cooperative(groups, (int)blockIdx.x - prob * groups, coop_part + (size_t)prob * 2 * groups * Reducer<S::NW>::NVMAX,
                              coop_counter + 2 * prob);
    }
    s.init();
    s.clear_rows();
    int rq = s.start();
    double fv = 0.0, pv = 0.0, gg = 0.0;
    while (rq) {
        if (!(rq & FL_REQ_SAME)) {
            if (rq & FL_REQ_NOMOVE) s.evaluate(fv, pv, gg);
            else s.move_evaluate(s.request_point(), fv, pv, gg); // the trial point is formed and evaluated in one pass
        }
        if (s.must_stop(fv)) { // (see Solver::must_stop)
            s.stop_not_finite();
            break;
        }
        rq = s.advance(fv, pv, gg);
    }
    s.finish();
}

} // namespace fl
