// fl_solver_kernels.hip -- fused batched solvers for MI355X (gfx950): ONE workgroup
// owns ONE problem for its whole solve (SteepestDescent / ConjugateGradient / L-BFGS /
// BFGS, optionally inside the augmented-Lagrangian outer loop).  The machine itself is
// in fl_device.hpp; this file holds the kernel, the size/solver dispatch and the C ABI.
//
//   * a line-search trial (x = x0 + a p, f, grad f, g.p, g.g) touches no HBM at all;
//     only the L-BFGS (s,y) ring / the BFGS inverse Hessian stream from HBM.
//   * every scalar of the machine is workgroup-uniform: all branches are taken by all
//     threads, barriers are safe anywhere in the machine.
//   * problems finish after different numbers of trials; the grid is one workgroup per
//     problem and the hardware dispatcher back-fills CUs as workgroups retire (no
//     lock-step batch, no host round trips).
#include "fl_solver_launch.hpp"
#include "fl_big.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>



namespace fl {

static int device_compute_units();
// problems other host threads of this process run on the same device at the same time (fl_multi_solve's shards): 0 = none known
static thread_local int tls_concurrent_batch = 0;

// n > 4096: vectors in HBM, one workgroup of 1024 threads per problem (fl_big.hpp: fl_big_solve_kernel) -- or, for few
// problems, `groups` workgroups per problem (the cooperative form)
template <int OBJ, int M> static hipError_t launch_big_k(const SolveArgs &A, double *rows, int groups, double *part, unsigned *counter, hipStream_t st)
{
    hipLaunchKernelGGL((fl_big_solve_kernel<OBJ, M>), dim3(A.batch * (groups > 1 ? groups : 1)), dim3(1024), 0, st, A, rows, groups, part, counter);
    return hipGetLastError();
}
template <int OBJ, int M> static int big_per_cu()
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fl_big_solve_kernel<OBJ, M>, 1024, 0) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
    }
    return per_cu;
}
// op 0: launch; op 1: workgroups of this kernel that fit one CU (the occupancy query)
template <int OBJ> static int big_dispatch_m(int op, int method, const SolveArgs *A, double *rows, int groups, double *part, unsigned *counter, hipStream_t st)
{
    switch (method) {
    case FL_SOLVER_SD: return op ? big_per_cu<OBJ, FL_SOLVER_SD>() : (int)launch_big_k<OBJ, FL_SOLVER_SD>(*A, rows, groups, part, counter, st);
    case FL_SOLVER_CG: return op ? big_per_cu<OBJ, FL_SOLVER_CG>() : (int)launch_big_k<OBJ, FL_SOLVER_CG>(*A, rows, groups, part, counter, st);
    case FL_SOLVER_BFGS: return op ? big_per_cu<OBJ, FL_SOLVER_BFGS>() : (int)launch_big_k<OBJ, FL_SOLVER_BFGS>(*A, rows, groups, part, counter, st);
    default: return op ? big_per_cu<OBJ, FL_SOLVER_LBFGS>() : (int)launch_big_k<OBJ, FL_SOLVER_LBFGS>(*A, rows, groups, part, counter, st);
    }
}
static int big_dispatch(int op, int obj, int method, const SolveArgs *A, double *rows, int groups, double *part, unsigned *counter, hipStream_t st)
{
    switch (obj) {
    case FL_OBJ_QUARTIC: return big_dispatch_m<FL_OBJ_QUARTIC>(op, method, A, rows, groups, part, counter, st);
    case FL_OBJ_ROSENBROCK: return big_dispatch_m<FL_OBJ_ROSENBROCK>(op, method, A, rows, groups, part, counter, st);
    default: return big_dispatch_m<FL_OBJ_DIAGQUAD>(op, method, A, rows, groups, part, counter, st);
    }
}
// Workgroups per problem for a fused solve beyond n = 4096 (fl_cooperative_groups_for).  More than one where the chip would
// otherwise stand mostly idle: every workgroup must be RESIDENT (the barriers spin) -- CUs x what the occupancy query says fits
// one CU (capped at one: a sibling on the same CU would share its LDS pipe for nothing) / batch -- and own at least two slots of
// every thread; not below 8 slots per thread (n <= 14336: little to share out, and the sums' order -- with it the result's last
// bits -- would change for sizes where it buys next to nothing).  FL_COOP_GROUPS in the environment overrides (1: never).
int big_cooperative_groups(int method, int objective, int batch, int n, bool coop_ok, int per_cu)
{
    using BS = BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_SD>;
    if (!coop_ok || n <= 4096 || batch <= 0) return 1;
    int cus = device_compute_units();
    if (cus > BS::COOP_MAX_GROUPS) cus = BS::COOP_MAX_GROUPS;
    if (per_cu < 1) return 1;
    const int nslot = BS::slots_for(n);
    // (fl_multi_solve runs several shards per device at once: all their workgroups must be resident TOGETHER, or the barriers of
    // one shard would spin for CUs the others hold)
    if (tls_concurrent_batch > batch) batch = tls_concurrent_batch;
    int want = cus / batch;
    if (want > nslot / 2) want = nslot / 2;
    // (beyond 64 the barrier outweighs the bandwidth: thread 0's poll, G partial sums to fetch and add left to right per
    // reduction -- one problem of n = 2^20, L-BFGS: 1 / 8 / 32 / 64 / 128 / 256 workgroups 1135 / 152 / 51 / 40.5 / 48.6 / 73.6 ms)
    if (want > 64) want = 64;
    if (nslot < 8) want = 1;
    if (const char *e = std::getenv("FL_COOP_GROUPS")) {
        want = std::atoi(e);
        if (want > cus / batch) want = cus / batch;
        if (want > nslot) want = nslot;
    }
    return want > 1 ? BS::coop_groups(n, want) : 1;
}
// the launch itself with the cooperative form's scratch (partial sums, arrival counters) from the stream-ordered allocator
hipError_t launch_big_with_groups(int groups, int batch, hipStream_t st, const std::function<hipError_t(int, double *, unsigned *)> &launch)
{
    if (groups <= 1) return launch(1, nullptr, nullptr);
    const size_t part_bytes = (size_t)batch * 2 * groups * Reducer<16>::NVMAX * sizeof(double), cnt_bytes = (size_t)batch * 2 * sizeof(unsigned);
    char *scr = nullptr;
    if (hipMallocAsync((void **)&scr, part_bytes + cnt_bytes, st) != hipSuccess) {
        (void)hipGetLastError();
        return launch(1, nullptr, nullptr); // (no scratch: one workgroup per problem as ever)
    }
    hipError_t e = hipMemsetAsync(scr + part_bytes, 0, cnt_bytes, st);
    if (e == hipSuccess) e = launch(groups, reinterpret_cast<double *>(scr), reinterpret_cast<unsigned *>(scr + part_bytes));
    const hipError_t ef = hipFreeAsync(scr, st);
    return e == hipSuccess ? ef : e;
}

// ------------------------------------------------------------ host dispatch
struct GeoSel {
    int nw, ept;
};
static constexpr int FL_BIG_MAX_N = 1 << 27;
// n beyond the register path: 1024 threads, 2*ceil(ceil(n/2)/1024) element slots per thread (fl_big.hpp)
static bool select_big_geometry(int n, GeoSel &g)
{
    if (n <= 4096 || n > FL_BIG_MAX_N) return false;
    g = {16, 2 * BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_SD>::slots_for(n)};
    return true;
}
static bool select_geometry(int n, GeoSel &g)
{
    if (n <= 0) return false;
#ifdef FL_FORCE_GEO_N // tuning builds: the bench geometry for FL_FORCE_GEO_LO < n <= FL_FORCE_GEO_N
    if (n > FL_FORCE_GEO_LO && n <= FL_FORCE_GEO_N) {
        g = {FL_BENCH_NW, FL_BENCH_EPT};
        return true;
    }
#endif
    if (n <= 128) g = {1, 2};
    else if (n <= 256) g = {1, 4};
    else if (n <= 512) g = {1, 8}; // one wave per problem: no cross-wave step in any reduction, and the reduction / line-search
                                   // scalar work is done once instead of twice (2x4: 73.1, 1x8: 85.7 M it/s; profiles/r02/ab_geo.txt)
#ifdef FL_BENCH_NW
    else if (n <= 1024) g = {FL_BENCH_NW, FL_BENCH_EPT};
#endif
    else if (n <= 1024) g = {2, 8}; // measured: 2x8 27.7, 4x4 25.7, 8x2 12.4 M it/s (profiles/r01/geometry_ab.txt)
    else if (n <= 2048) g = {4, 8};
    else if (n <= 4096) g = {8, 8};
    else return false;
    return true;
}
// The FUSED kernels of SteepestDescent / ConjugateGradient without constraints run 512 < n <= 1024 with ONE wave x 16
// elements per thread: they keep no history, so their state fits one wave's registers (250 VGPRs, 2 waves per SIMD),
// every reduction stays inside the wave and the reduction / line-search scalar work is done once instead of twice
// (C3: 65.4 -> 54.3 ms).  Same padded length threads*ept as the layout geometry of n; L-BFGS (row buffers, ring pairs)
// and the dense solvers keep 2 x 8, and so do the reverse-communication kernels.
static bool select_fused_geometry(int n, int method, bool aug, GeoSel &g)
{
    if (!select_geometry(n, g)) return false;
    if (!aug && (method == FL_SOLVER_SD || method == FL_SOLVER_CG) && n > 512) g = {g.nw / 2, 16}; // 1 x 16, 2 x 16, 4 x 16
    // NewtonRaphson (also inside the augmented Lagrangian) keeps two waves x 4 for 256 < n <= 512: the Cholesky kernels of
    // one workgroup run 17 % faster on 128 threads than on 64 (profiles/r02/ab_geo.txt)
    if (method == FL_SOLVER_NEWTON && n > 256 && n <= 512) g = {2, 4};
    return true;
}
// The geometry is a function of n (and the solver) alone -- NOT of the batch size: round 4 measured "latency" geometries
// (2 x 2, 2 x 4, 4 x 2, 4 x 4, 8 x 2, 8 x 4: more waves x fewer elements per thread) for batches that under-fill the chip, all
// bit-exact against the oracle, and none pays: the unconstrained solvers are fastest in their throughput geometry at EVERY
// batch from 256 to 8192 (one exception, 8 %: L-BFGS n = 1024 at 256 problems on 4 x 4), and BASELINE config 5's shares are
// served better by helper waves (below), which leave the summation order -- hence every bit of the result -- alone
// (profiles/r04/geometry_by_batch.txt; the kernels are in the history at commit ca4f3a1).
static int device_compute_units()
{
    static std::atomic<int> cus{0};
    int c = cus.load(std::memory_order_relaxed);
    if (c > 0) return c;
    c = 256;
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
        c = pr.multiProcessorCount;
    else (void)hipGetLastError();
    cus.store(c, std::memory_order_relaxed);
    return c;
}

// Helper waves (fl_solve_rep_kernel): how many waves share a problem's objective-only shrink loop by trial -- the master that
// runs the machine plus R - 1 helpers.  Only where that loop runs as a tight loop at all (Solver::spec_shrinking: augmented
// Lagrangian around L-BFGS / CG, diagonal-quadratic or quartic objective, constraint blocks that are aligned lane groups) and
// the geometry is one wave (128 < n <= 512).  Invisible in the results, so there is no option for it: by batch alone.
// Measured on BASELINE config 5's family (ms; profiles/r04/geometry_by_batch.txt), R forced for the whole launch:
//   batch      256   512  1024  1536  2048  3072  4096  8192
//   R = 1     61.2  62.0  70.3  70.5  73.0  80.8 104.6 139.0
//   R = 2     41.0  50.9  54.1  56.0  63.2  84.8 101.8 178.2
//   R = 4     29.6  36.2  53.5  67.6  72.2 105.7 138.6 255.0
//   shipped   29.7  36.2  41.5  49.7  54.3  58.1  68.9 113.5   <- R by batch for the START of the launch (below), then STAGED
//                                                                 (solve(): the unfinished problems move to more waves)
// FL_FORCE_REPLICAS in the environment overrides (tuning: tools/geometry_by_batch.py; it also switches the staging off).
#ifndef FL_REP4_MAX_BATCH
#define FL_REP4_MAX_BATCH 512 // (problems per 256 CUs)
#define FL_REP2_MAX_BATCH 2560
#endif
#ifndef FL_STAGE_T2
#define FL_STAGE_T2 1536 // staged launches: problems left when the one-helper stage takes over ... (per 256 CUs)
#define FL_STAGE_T4 512  // ... and the three-helper stage
#endif
// (fl_multi_solve runs several shards on one device at once: each of its threads says how many problems the device holds)
static int select_replicas(const GeoSel &g, int objective, int method, int n, int m, int batch)
{
    if (tls_concurrent_batch > batch) batch = tls_concurrent_batch;
    if (g.nw != 1 || (g.ept != 8 && g.ept != 4)) return 1;
    if (objective != FL_OBJ_DIAGQUAD && objective != FL_OBJ_QUARTIC) return 1;
    if (method != FL_SOLVER_LBFGS && method != FL_SOLVER_CG) return 1;
    const int w = n / m;
    if (w != 32 && w != 64 && w != 128) return 1; // (Solver::init: cshift)
    if (const char *f = getenv("FL_FORCE_REPLICAS")) {
        const int r = atoi(f);
        return (r == 2 || r == 4 || r == 8) ? r : 1;
    }
    const long long scaled = (long long)batch * 256 / device_compute_units();
    if (scaled <= FL_REP4_MAX_BATCH) return 4;
    if (scaled <= FL_REP2_MAX_BATCH) return 2;
    return 1;
}

// the problems a staged launch left paused -> the next launch's list (order irrelevant: the problems are independent)
__global__ void collect_paused_kernel(const int *status, int batch, int *list, int *sched)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < batch && status[k] == FL_STATUS_PAUSED) list[atomicAdd(sched + 1, 1)] = k;
}

#ifndef FL_ONLY_BENCH // the geometries are compiled in fl_solver_g*.hip
extern template hipError_t launch_rep<1, 8>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_rep<1, 4>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_vec<1, 16>(int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_vec<2, 16>(int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_vec<4, 16>(int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_newton<2, 4>(int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<1, 2>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<1, 4>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<1, 8>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<2, 8>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<4, 8>(int, int, int, const SolveArgs &, hipStream_t);
extern template hipError_t launch_o<8, 8>(int, int, int, const SolveArgs &, hipStream_t);
#endif
static hipError_t launch(const GeoSel &g, int obj, int method, int aug, const SolveArgs &A, hipStream_t st)
{
#ifdef FL_ONLY_BENCH // tuning builds (tools/variants.sh): only the bench.py instantiation
#ifndef FL_BENCH_NW
#define FL_BENCH_NW 4
#define FL_BENCH_EPT 4
#endif
    return launch_k<FL_BENCH_NW, FL_BENCH_EPT, FL_OBJ_DIAGQUAD, FL_SOLVER_LBFGS, 0>(A, st);
#else
    if (g.nw == 1 && g.ept == 16) return launch_vec<1, 16>(obj, method, A, st); // (select_fused_geometry: SD / CG, no constraints)
    if (g.nw == 2 && g.ept == 16) return launch_vec<2, 16>(obj, method, A, st);
    if (g.nw == 4 && g.ept == 16) return launch_vec<4, 16>(obj, method, A, st);
    if (g.nw == 2 && g.ept == 4) return launch_newton<2, 4>(obj, aug, A, st);      // (select_fused_geometry: NewtonRaphson)
    if (g.nw == 1 && g.ept == 2) return launch_o<1, 2>(obj, method, aug, A, st);
    if (g.nw == 1 && g.ept == 4) return launch_o<1, 4>(obj, method, aug, A, st);
    if (g.nw == 1 && g.ept == 8) return launch_o<1, 8>(obj, method, aug, A, st);
    if (g.nw == 2 && g.ept == 8) return launch_o<2, 8>(obj, method, aug, A, st);
    if (g.nw == 4 && g.ept == 8) return launch_o<4, 8>(obj, method, aug, A, st);
    return launch_o<8, 8>(obj, method, aug, A, st);
#endif
}

// The launches an augmented-Lagrangian solve of `batch` problems is made of: stage k runs `rep[k]` waves per problem and hands
// the problems still unfinished when at most `pause[k]` are left (0: runs to the end) to stage k + 1.  One stage = one launch.
static int aug_launch_plan(const GeoSel &g, int objective, int method, int n, int m, int batch, int (&rep)[3], int (&pause)[3])
{
    const int r0 = select_replicas(g, objective, method, n, m, batch);
    rep[0] = r0;
    pause[0] = 0;
    const bool eligible = select_replicas(g, objective, method, n, m, 1) > 1 && !getenv("FL_FORCE_REPLICAS");
    const char *stg = getenv("FL_AUG_STAGED");
    const long long cus = device_compute_units();
    int t2 = (int)((long long)FL_STAGE_T2 * cus / 256), t4 = (int)((long long)FL_STAGE_T4 * cus / 256);
    if (const char *e2 = getenv("FL_STAGE_T2")) t2 = atoi(e2) > 0 ? atoi(e2) : t2; // (tuning: tools/stage_sweep.py)
    if (const char *e4 = getenv("FL_STAGE_T4")) t4 = atoi(e4) > 0 && atoi(e4) <= t2 ? atoi(e4) : t4;
    if (!eligible || r0 >= 4 || batch <= t4 || (stg && stg[0] == '0')) return 1;
    int ns = 0;
    if (r0 == 1 && batch > t2) { rep[ns] = 1; pause[ns++] = t2; }
    rep[ns] = ns ? 2 : r0; pause[ns++] = t4;
    rep[ns] = 4; pause[ns++] = 0;
    return ns;
}

struct AugArgs {
    int m;
    double miu0;
    double *lambda;
    int *outer;
    double *cnorm2;
};

static int solve(int method, int objective, int batch, int n, double *x, const double *d, const double *b,
                 const fl_options *opt, void *ws, size_t ws_bytes, double *f, double *gg, int32_t *iters,
                 int32_t *status, int32_t *nf, int32_t *ng, const AugArgs *aug, void *stream)
{
    if (!x || !opt || batch <= 0 || n <= 0) return FL_ERR_INVALID_ARGUMENT;
    if (objective < FL_OBJ_QUARTIC || objective > FL_OBJ_DIAGQUAD) return FL_ERR_INVALID_ARGUMENT;
    if (objective == FL_OBJ_DIAGQUAD && (!d || !b)) return FL_ERR_INVALID_ARGUMENT;
    if (opt->cg_method != FL_CG_DY && opt->cg_method != FL_CG_PR) return FL_ERR_INVALID_ARGUMENT;
    GeoSel g;
    bool big = false;
    if (!select_fused_geometry(n, method, aug != nullptr, g)) {
        // beyond the register path: SD / CG / L-BFGS continue with vectors in HBM; the dense solvers and the
        // augmented Lagrangian do not
        if (aug || method == FL_SOLVER_NEWTON) return FL_ERR_UNSUPPORTED_SIZE;
        // BFGS: quasi-Newton updates only (the exact-Hessian refresh is a dense Cholesky), H [n][npad] up to n = 16384
        if (method == FL_SOLVER_BFGS && (opt->exact_step > 0 || n > BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>::BF_MAX_N))
            return FL_ERR_UNSUPPORTED_SIZE;
        if (!select_big_geometry(n, g)) return FL_ERR_UNSUPPORTED_SIZE;
        big = true;
    }
    const int mem = opt->memory > 1 ? opt->memory : 1; // mem=max(1,Memory)
    if (method == FL_SOLVER_LBFGS && mem > FL_MAX_MEMORY) return FL_ERR_UNSUPPORTED_SIZE;
    if (method == FL_SOLVER_LBFGS || method == FL_SOLVER_BFGS || method == FL_SOLVER_NEWTON) {
        if (!ws || ws_bytes < fl_workspace_bytes_for(method, batch, n, opt)) return FL_ERR_WORKSPACE;
    }
    SolveArgs A;
    fill_solve_args(A, method, batch, n, x, d, b, opt, ws, f, gg, iters, status, nf, ng);
    if (aug) {
        if (aug->m < 1 || aug->m > FL_MAX_CONSTRAINTS || n % aug->m != 0 || !aug->lambda)
            return FL_ERR_INVALID_ARGUMENT;
        // the reference's whole menu (NO.f90:2074-2185): NewtonRaphson and BFGS take the analytic Hessian of L (the fdd /
        // cdd branch, Ldd) -- the built-in objectives and constraints have one; BFGS with exact_step <= 0 never asks
        if (method != FL_SOLVER_LBFGS && method != FL_SOLVER_CG && method != FL_SOLVER_BFGS && method != FL_SOLVER_NEWTON)
            return FL_ERR_INVALID_ARGUMENT;
        A.aug_m = aug->m;
        A.miu0 = aug->miu0;
        A.lambda = aug->lambda;
        A.outer = aug->outer;
        A.cnorm2 = aug->cnorm2;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (big) {
        // workspace: [hist: batch*2*mem rows][p, x0, g0, g: batch*4 rows], rows of npad doubles; SD / CG have no
        // workspace argument in the C ABI: their four rows come from the stream-ordered allocator
        const size_t npad = (size_t)g.nw * 64 * g.ept, rows_bytes = (size_t)batch * 4 * npad * sizeof(double);
        double *rows = nullptr;
        const bool own_rows = (method == FL_SOLVER_SD || method == FL_SOLVER_CG);
        if (method == FL_SOLVER_LBFGS) {
            rows = static_cast<double *>(ws) + (size_t)batch * 2 * (size_t)A.mem * npad;
        } else if (method == FL_SOLVER_BFGS) {
            rows = static_cast<double *>(ws) + (size_t)batch * BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>::bfgs_rows(n) * npad;
        } else if (hipMallocAsync((void **)&rows, rows_bytes, st) != hipSuccess) {
            (void)hipGetLastError(); // the failed allocation must not taint the caller's next call
            return FL_ERR_WORKSPACE;
        }
        const bool coop_ok = method != FL_SOLVER_BFGS && objective != FL_OBJ_ROSENBROCK;
        const int groups = big_cooperative_groups(method, objective, batch, n, coop_ok, coop_ok ? big_dispatch(1, objective, method, nullptr, nullptr, 1, nullptr, nullptr, st) : 0);
        hipError_t e = launch_big_with_groups(groups, batch, st, [&](int G, double *part, unsigned *counter) {
            return (hipError_t)big_dispatch(0, objective, method, &A, rows, G, part, counter, st);
        });
        if (own_rows) { // stream-ordered: the rows are released once the kernel above has finished
            const hipError_t ef = hipFreeAsync(rows, st);
            if (e == hipSuccess) e = ef;
        }
        return launch_status(e);
    }
#ifndef FL_ONLY_BENCH
    if (aug) {
        const int rep = select_replicas(g, objective, method, n, aug->m, batch);
        auto launch_with = [&](int r, const SolveArgs &B) {
            if (r > 1) return g.ept == 8 ? launch_rep<1, 8>(r, objective, method, B, st) : launch_rep<1, 4>(r, objective, method, B, st);
            return launch(g, objective, method, 1, B, st);
        };
        // Where helper waves exist for this kernel, a batch too large for them still ENDS as a small one: the evaluation counts
        // of the problems differ 16-fold (BASELINE config 5: 8.5 k ... 150 k), so the last third of the launch is a tail of a
        // few hundred long problems on a mostly idle chip.  STAGED: the launch picked by batch size (plain, or one helper wave)
        // runs until at most FL_STAGE_T2 problems are unfinished; those pause at their next outer iteration's boundary -- where
        // the reference starts a fresh inner solve from (x, lambda, miu) anyway, so resuming there changes no bit -- and a
        // second launch continues them with one helper wave each, a third the last FL_STAGE_T4 with three.  Stream-ordered
        // throughout: the lists are built on the device, a listed launch is sized for the most problems it can get and its
        // surplus workgroups leave at once.  BASELINE config 5: 140 -> 112 ms; 4096 problems: 102 -> 70 ms (the thresholds sit
        // on a flat optimum: profiles/r04/c5_staged.txt).  FL_AUG_STAGED=0 in the environment: one launch as before.
        int stage_rep[3], stage_pause[3];
        const int ns = aug_launch_plan(g, objective, method, n, aug->m, batch, stage_rep, stage_pause);
        if (ns > 1 && status) {
            const int list_cap = stage_pause[0];
            // scratch: [2] scheduler words | list [list_cap] | paused state [batch][FL_PSTATE] doubles
            const size_t list_off = 16, ps_off = (list_off + (size_t)list_cap * sizeof(int) + 15) & ~(size_t)15;
            const size_t bytes = ps_off + (size_t)batch * FL_PSTATE * sizeof(double);
            char *scr = nullptr;
            if (hipMallocAsync((void **)&scr, bytes, st) != hipSuccess) {
                (void)hipGetLastError();
                return launch_status(launch_with(rep, A)); // (no scratch: the single launch)
            }
            int *sched = reinterpret_cast<int *>(scr), *list = reinterpret_cast<int *>(scr + list_off);
            SolveArgs B = A;
            B.sched = sched;
            B.pstate = reinterpret_cast<double *>(scr + ps_off);
            hipError_t e = hipMemsetAsync(sched, 0, 16, st);
            B.pause_below = stage_pause[0];
            if (e == hipSuccess) e = launch_with(stage_rep[0], B);
            for (int sgi = 1; sgi < ns && e == hipSuccess; ++sgi) {
                e = hipMemsetAsync(sched + 1, 0, sizeof(int), st);
                if (e != hipSuccess) break;
                hipLaunchKernelGGL(collect_paused_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, status, batch, list, sched);
                e = hipGetLastError();
                if (e != hipSuccess) break;
                B.list = list;
                B.resume = 1;
                B.pause_below = stage_pause[sgi];
                B.pause_grid = stage_pause[sgi - 1]; // (at most that many paused)
                e = launch_with(stage_rep[sgi], B);
            }
            const hipError_t ef = hipFreeAsync(scr, st);
            if (e == hipSuccess) e = ef;
            return launch_status(e);
        }
        if (rep > 1) return launch_status(launch_with(rep, A));
    }
#endif
    hipError_t e = launch(g, objective, method, aug != nullptr, A, st);
    return launch_status(e);
}

// how many of the newest (s, y) pairs the fused L-BFGS kernel for this objective and dimension keeps on the chip
// (registers + LDS ring): the rest of the ring streams from HBM twice per iteration
template <int NW, int EPT> static int onchip_pairs_o(int obj)
{
    switch (obj) {
    case FL_OBJ_QUARTIC: { using S = Solver<NW, EPT, FL_OBJ_QUARTIC, FL_SOLVER_LBFGS, 0>; return S::LDS_PAIRS + S::REG_PAIRS; }
    case FL_OBJ_ROSENBROCK: { using S = Solver<NW, EPT, FL_OBJ_ROSENBROCK, FL_SOLVER_LBFGS, 0>; return S::LDS_PAIRS + S::REG_PAIRS; }
    default: { using S = Solver<NW, EPT, FL_OBJ_DIAGQUAD, FL_SOLVER_LBFGS, 0>; return S::LDS_PAIRS + S::REG_PAIRS; }
    }
}

} // namespace fl

extern "C" {

int fl_version(void) { return 104; }

int fl_augmented_lagrangian_launch_plan(int solver, int objective, int batch, int n, int m, int *waves, int *pause_below, int max_stages)
{
    fl::GeoSel g;
    if (batch <= 0 || m <= 0 || !fl::select_fused_geometry(n, solver, true, g)) return FL_ERR_INVALID_ARGUMENT;
    int rep[3], pause[3];
    const int ns = fl::aug_launch_plan(g, objective, solver, n, m, batch, rep, pause);
    for (int k = 0; k < ns && k < max_stages; ++k) {
        if (waves) waves[k] = rep[k] * g.nw;
        if (pause_below) pause_below[k] = pause[k];
    }
    return ns;
}

void fl_internal_set_concurrent_batch(int problems) { fl::tls_concurrent_batch = problems > 0 ? problems : 0; }

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
    o->exact_step = 20; // NO.f90:652-653
}

size_t fl_workspace_bytes_for(int solver, int batch, int n, const fl_options *opt)
{
    if (!opt) return 0;
    if (solver == FL_SOLVER_BFGS || solver == FL_SOLVER_NEWTON) {
        fl::GeoSel g;
        if (batch <= 0) return 0;
        if (!fl::select_geometry(n, g)) return fl_workspace_bytes(solver, batch, n, 0); // beyond the register path

        const size_t npad = (size_t)(g.nw * 64 * g.ept);
        if (solver == FL_SOLVER_NEWTON) return (size_t)batch * (size_t)n * npad * sizeof(double);
        // H (+ Hessian / factor and inverse factor with ExactStep > 0) + the rows of the deferred updates (n > 1024)
        const size_t defer = npad >= FL_BFGS_DEFER_NPAD ? 2 * FL_BFGS_DEFER : 0;
        return (size_t)batch * ((size_t)(opt->exact_step > 0 ? 3 : 1) * (size_t)n + defer) * npad * sizeof(double);
    }
    return fl_workspace_bytes(solver, batch, n, opt->memory);
}

int fl_newton_raphson_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                              const double *b_dev, const fl_options *opt, void *workspace_dev,
                              size_t workspace_bytes, double *f_dev, double *gg_dev, int32_t *iters_dev,
                              int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream)
{
    return fl::solve(FL_SOLVER_NEWTON, objective, batch, n, x_dev, d_dev, b_dev, opt, workspace_dev, workspace_bytes,
                     f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, nullptr, stream);
}

int fl_lbfgs_onchip_pairs(int objective, int n)
{
    fl::GeoSel g;
    if (!fl::select_geometry(n, g)) return 0; // vectors-in-HBM path: nothing of the ring stays on the chip
    if (g.nw == 1 && g.ept == 2) return fl::onchip_pairs_o<1, 2>(objective);
    if (g.nw == 1 && g.ept == 4) return fl::onchip_pairs_o<1, 4>(objective);
    if (g.nw == 1 && g.ept == 8) return fl::onchip_pairs_o<1, 8>(objective);
    if (g.nw == 2 && g.ept == 8) return fl::onchip_pairs_o<2, 8>(objective);
    if (g.nw == 4 && g.ept == 8) return fl::onchip_pairs_o<4, 8>(objective);
    return fl::onchip_pairs_o<8, 8>(objective);
}

int fl_reduction_geometry(int n, int *threads, int *ept)
{
    fl::GeoSel g;
    if (!fl::select_geometry(n, g) && !fl::select_big_geometry(n, g)) return FL_ERR_UNSUPPORTED_SIZE;
    if (threads) *threads = g.nw * 64;
    if (ept) *ept = g.ept;
    return FL_OK;
}

// rank-2 updates the fused BFGS kernels of dimension n keep pending before they fold them into H (0: every update is applied at once):
// the update form a bit-exact replay must use (oracle update_form 100 + this)
int fl_bfgs_deferred_updates(int n)
{
    fl::GeoSel g;
    if (n <= 0) return 0;
    if (!fl::select_geometry(n, g)) return n <= fl::BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>::BF_MAX_N ? FL_BFGS_DEFER : 0; // vectors in HBM
    return g.nw * 64 * g.ept >= FL_BFGS_DEFER_NPAD ? FL_BFGS_DEFER : 0;
}

// workgroups that share one problem in the fused solve of this batch on the current device (1: none; > 1: the cooperative
// form of the vectors-in-HBM path, whose sums are those of the oracle's tree order with `groups`)
int fl_cooperative_groups_for(int solver, int objective, int batch, int n)
{
    if (n <= 4096 || batch <= 0 || objective < FL_OBJ_QUARTIC || objective > FL_OBJ_DIAGQUAD) return 1;
    if (solver != FL_SOLVER_SD && solver != FL_SOLVER_CG && solver != FL_SOLVER_LBFGS) return 1;
    const bool coop_ok = objective != FL_OBJ_ROSENBROCK;
    return fl::big_cooperative_groups(solver, objective, batch, n, coop_ok,
                                      coop_ok ? fl::big_dispatch(1, objective, solver, nullptr, nullptr, 1, nullptr, nullptr, nullptr) : 0);
}

int fl_reduction_geometry_for(int solver, int n, int *threads, int *ept)
{
    fl::GeoSel g;
    if (!fl::select_fused_geometry(n, solver, false, g)) return fl_reduction_geometry(n, threads, ept);
    if (threads) *threads = g.nw * 64;
    if (ept) *ept = g.ept;
    return FL_OK;
}

size_t fl_workspace_bytes(int solver, int batch, int n, int memory)
{
    fl::GeoSel g;
    if (batch <= 0) return 0;
    if (!fl::select_geometry(n, g)) { // vectors-in-HBM path: the ring / the inverse Hessian plus the four vector rows
        if (!fl::select_big_geometry(n, g)) return 0;
        const size_t npad = (size_t)g.nw * 64 * g.ept;
        using BB = fl::BigSolver<FL_OBJ_QUARTIC, FL_SOLVER_BFGS>;
        if (solver == FL_SOLVER_BFGS && n <= BB::BF_MAX_N) return (size_t)batch * (BB::bfgs_rows(n) + 4) * npad * sizeof(double);
        if (solver != FL_SOLVER_LBFGS) return 0;
        const size_t mem = memory > 1 ? (size_t)memory : 1;
        return (size_t)batch * (2 * mem + 4) * npad * sizeof(double);
    }
    const size_t npad = (size_t)(g.nw * 64 * g.ept);
    if (solver == FL_SOLVER_LBFGS) {
        const size_t mem = memory > 1 ? (size_t)memory : 1;
        return (size_t)batch * 2 * mem * npad * sizeof(double);
    }
    if (solver == FL_SOLVER_BFGS) // ExactStep <= 0; see fl_workspace_bytes_for
        return (size_t)batch * ((size_t)n + (npad >= FL_BFGS_DEFER_NPAD ? 2 * FL_BFGS_DEFER : 0)) * npad * sizeof(double);
    return 0;
}

int fl_lbfgs_batched(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                     const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                     double *gg_dev, int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                     void *stream)
{
    return fl::solve(FL_SOLVER_LBFGS, objective, batch, n, x_dev, d_dev, b_dev, opt, workspace_dev, workspace_bytes,
                     f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, nullptr, stream);
}

int fl_bfgs_batched(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                    const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev, double *gg_dev,
                    int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream)
{
    return fl::solve(FL_SOLVER_BFGS, objective, batch, n, x_dev, d_dev, b_dev, opt, workspace_dev, workspace_bytes,
                     f_dev, gg_dev, iters_dev, status_dev, nf_dev, ng_dev, nullptr, stream);
}

int fl_conjugate_gradient_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                  const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                  int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                  void *stream)
{
    return fl::solve(FL_SOLVER_CG, objective, batch, n, x_dev, d_dev, b_dev, opt, nullptr, 0, f_dev, gg_dev,
                     iters_dev, status_dev, nf_dev, ng_dev, nullptr, stream);
}

int fl_steepest_descent_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                void *stream)
{
    return fl::solve(FL_SOLVER_SD, objective, batch, n, x_dev, d_dev, b_dev, opt, nullptr, 0, f_dev, gg_dev,
                     iters_dev, status_dev, nf_dev, ng_dev, nullptr, stream);
}

int fl_augmented_lagrangian_batched(int solver, int objective, int batch, int n, int m, double *x_dev,
                                    const double *d_dev, const double *b_dev, double *lambda_dev, double miu0,
                                    const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                                    double *cnorm2_dev, int32_t *iters_dev, int32_t *outer_dev, int32_t *status_dev,
                                    int32_t *nf_dev, int32_t *ng_dev, void *stream)
{
    fl::AugArgs aug = {m, miu0, lambda_dev, outer_dev, cnorm2_dev};
    return fl::solve(solver, objective, batch, n, x_dev, d_dev, b_dev, opt, workspace_dev, workspace_bytes, f_dev,
                     nullptr, iters_dev, status_dev, nf_dev, ng_dev, &aug, stream);
}

} // extern "C"
