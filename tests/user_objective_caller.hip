// A caller-compiled objective in the fused solver kernels (include/fl_user_objective.hpp): the diagonal quadratic
// f = 1/2 sum d x^2 - sum b x restated as a USER functor must reproduce the built-in FL_OBJ_DIAGQUAD bit for bit --
// minimiser, objective, g.g, iteration and evaluation counts -- for L-BFGS (n = 1024: two waves x 8, n = 256: one
// wave x 4), ConjugateGradient, SteepestDescent and quasi-Newton BFGS (n = 256).  Built and run by
// tests/test_gpu_user_objective.py on the GPU box; prints one line per case and "ALL OK".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

template <int NW, int EPT> struct MyQuadratic;
template <int NW, int EPT> struct MySpheres; // the caller's CONSTRAINTS (AugmentedLagrangian's c, cd: NO.f90:1928-1934) -- see aug_case()
#define FL_USER_OBJECTIVE MyQuadratic
#define FL_USER_CONSTRAINTS MySpheres
#define FL_USER_TUNE_LIKE FL_OBJ_DIAGQUAD
#include "../include/fl_user_objective.hpp"

struct Params { // the objective's own parameter block (device memory): here just a scale that is 1
    double half;
};

template <int NW, int EPT> struct MyQuadratic {
    static constexpr int LDS_DOUBLES = 0;
    double d[EPT], b[EPT], half;
    __device__ void init(const fl::SolveArgs &A, int prob, double *)
    {
        fl::load_user<NW, EPT>(A.d + (size_t)prob * A.n, A.n, d);
        fl::load_user<NW, EPT>(A.b + (size_t)prob * A.n, A.n, b);
        half = static_cast<const Params *>(A.user)->half;
    }
    __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int, double *)
    {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const double dx = d[k] * x[k];
            const double t0 = dx * x[k], t1 = b[k] * x[k];
            g[k] = dx - b[k];
            s0 = (k == 0) ? t0 : s0 + t0;
            s1 = (k == 0) ? t1 : s1 + t1;
        }
        if (half != 0.5) s0 = __builtin_nan(""); // (the parameter block did arrive)
    }
    __device__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
};

// m block spheres c_j = sum_{i in block j} x_i^2 - 1: the library's own constraint family restated as a user functor
template <int NW, int EPT> struct MySpheres {
    using G = fl::Geo<NW, EPT>;
    int w;
    __device__ void init(const fl::SolveArgs &A, int) { w = A.n / A.aug_m; }
    __device__ void partial(const double (&x)[EPT], double (&cp)[8], int n)
    {
        for (int j = 0; j < 8; ++j) {
            double acc = 0.0;
            for (int k = 0; k < EPT; ++k) {
                const int e = G::e0(k >> 1) + (k & 1);
                const double t = (e < n && e / w == j) ? x[k] * x[k] : 0.0;
                acc = (k == 0) ? t : acc + t;
            }
            cp[j] = acc;
        }
    }
    __device__ double offset(int) const { return -1.0; }
    __device__ void add_gradient(const double (&x)[EPT], const double (&v)[8], double (&g)[EPT], int n)
    {
        for (int k = 0; k < EPT; ++k) {
            const int e = G::e0(k >> 1) + (k & 1);
            if (e < n) {
                double vv = 0.0;
                for (int j = 0; j < 8; ++j) vv = (e / w == j) ? v[j] : vv;
                g[k] = g[k] + (2.0 * x[k]) * vv;
            }
        }
    }
};

#define CK(x)                                                                   \
    do {                                                                        \
        if ((x) != hipSuccess) {                                                \
            std::printf("HIP error at %s:%d\n", __FILE__, __LINE__);            \
            return 2;                                                           \
        }                                                                       \
    } while (0)

template <int NW, int EPT> static int one_case(int solver, const char *name, int batch, int n, double precision)
{
    const size_t N = (size_t)batch * n;
    double *d, *b, *xa, *xb, *fa, *fb, *ga, *gb;
    int32_t *ia, *ib, *sa, *sb, *nfa, *nfb, *nga, *ngb;
    Params hp = {0.5}, *pd;
    CK(hipMalloc(&d, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&xa, N * 8)); CK(hipMalloc(&xb, N * 8));
    CK(hipMalloc(&fa, batch * 8)); CK(hipMalloc(&fb, batch * 8)); CK(hipMalloc(&ga, batch * 8)); CK(hipMalloc(&gb, batch * 8));
    CK(hipMalloc(&ia, batch * 4)); CK(hipMalloc(&ib, batch * 4)); CK(hipMalloc(&sa, batch * 4)); CK(hipMalloc(&sb, batch * 4));
    CK(hipMalloc(&nfa, batch * 4)); CK(hipMalloc(&nfb, batch * 4)); CK(hipMalloc(&nga, batch * 4)); CK(hipMalloc(&ngb, batch * 4));
    CK(hipMalloc(&pd, sizeof hp));
    CK(hipMemcpy(pd, &hp, sizeof hp, hipMemcpyHostToDevice));
    if (fl_synth_diag_spectrum(7, batch, n, 10.0, 300.0, d, nullptr) != FL_OK) return 3;
    if (fl_synth_uniform(7, batch, n, -1.0, 1.0, b, nullptr) != FL_OK) return 3;
    CK(hipMemset(xa, 0, N * 8)); CK(hipMemset(xb, 0, N * 8));
    fl_options o;
    fl_default_options(&o, solver);
    o.precision = precision;
    o.exact_step = 0;
    if (solver == FL_SOLVER_SD || solver == FL_SOLVER_BFGS) o.max_iteration = 60;
    const size_t wsb = fl_workspace_bytes_for(solver, batch, n, &o);
    void *wa = nullptr, *wb = nullptr;
    if (wsb) { CK(hipMalloc(&wa, wsb)); CK(hipMalloc(&wb, wsb)); }
    int rc;
    switch (solver) {
    case FL_SOLVER_SD: rc = fl_steepest_descent_batched(FL_OBJ_DIAGQUAD, batch, n, xa, d, b, &o, fa, ga, ia, sa, nfa, nga, nullptr); break;
    case FL_SOLVER_CG: rc = fl_conjugate_gradient_batched(FL_OBJ_DIAGQUAD, batch, n, xa, d, b, &o, fa, ga, ia, sa, nfa, nga, nullptr); break;
    case FL_SOLVER_BFGS: rc = fl_bfgs_batched(FL_OBJ_DIAGQUAD, batch, n, xa, d, b, &o, wa, wsb, fa, ga, ia, sa, nfa, nga, nullptr); break;
    default: rc = fl_lbfgs_batched(FL_OBJ_DIAGQUAD, batch, n, xa, d, b, &o, wa, wsb, fa, ga, ia, sa, nfa, nga, nullptr); break;
    }
    if (rc != FL_OK) { std::printf("%s: built-in solver failed %d\n", name, rc); return 4; }
    rc = fl::user::solve<NW, EPT>(solver, batch, n, xb, d, b, pd, &o, wb, wsb, fb, gb, ib, sb, nfb, ngb, nullptr);
    if (rc != FL_OK) { std::printf("%s: fl::user::solve failed %d\n", name, rc); return 5; }
    CK(hipDeviceSynchronize());
    std::vector<double> ha(N), hb(N), hfa(batch), hfb(batch), hga(batch), hgb(batch);
    std::vector<int32_t> hia(batch), hib(batch), hna(batch), hnb(batch), hsa(batch), hsb(batch);
    CK(hipMemcpy(ha.data(), xa, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), xb, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hfa.data(), fa, batch * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hfb.data(), fb, batch * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hga.data(), ga, batch * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hgb.data(), gb, batch * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hia.data(), ia, batch * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hib.data(), ib, batch * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hna.data(), nfa, batch * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hnb.data(), nfb, batch * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hsa.data(), sa, batch * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hsb.data(), sb, batch * 4, hipMemcpyDeviceToHost));
    long it = 0;
    for (int k = 0; k < batch; ++k) it += hia[k];
    const bool same = std::memcmp(ha.data(), hb.data(), N * 8) == 0 && std::memcmp(hfa.data(), hfb.data(), batch * 8) == 0 &&
                      std::memcmp(hga.data(), hgb.data(), batch * 8) == 0 && hia == hib && hna == hnb && hsa == hsb;
    std::printf("%s n=%d batch=%d: %ld iterations, user functor %s the built-in objective\n", name, n, batch, it,
                same ? "reproduces bit for bit" : "DIFFERS from");
    // a geometry that is not the one of this n is refused, not run
    if (fl::user::solve<NW, EPT>(solver, batch, n > 128 ? 100 : 1000, xb, d, b, pd, &o, wb, wsb, fb, gb, ib, sb, nfb, ngb, nullptr) !=
        FL_ERR_INVALID_ARGUMENT)
        return 6;
    for (void *p : {(void *)d, (void *)b, (void *)xa, (void *)xb, (void *)fa, (void *)fb, (void *)ga, (void *)gb, (void *)ia, (void *)ib,
                    (void *)sa, (void *)sb, (void *)nfa, (void *)nfb, (void *)nga, (void *)ngb, (void *)pd, wa, wb})
        if (p) (void)hipFree(p);
    return same && it > 0 ? 0 : 1;
}

// AugmentedLagrangian around the user's objective AND the user's constraints (fl::user::solve_auglag) against
// fl_augmented_lagrangian_batched on BASELINE config 5's shape: n = 512, 8 block spheres, L-BFGS inside
static int aug_case()
{
    const int batch = 24, n = 512, m = 8;
    const size_t N = (size_t)batch * n;
    double *d, *b, *xa, *xb, *la, *lb, *fa, *fb, *ca, *cb;
    int32_t *ia, *ib, *oa, *ob, *sa, *sb, *nfa, *nfb, *nga, *ngb;
    Params hp = {0.5}, *pd;
    CK(hipMalloc(&d, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&xa, N * 8)); CK(hipMalloc(&xb, N * 8));
    CK(hipMalloc(&la, batch * m * 8)); CK(hipMalloc(&lb, batch * m * 8)); CK(hipMalloc(&fa, batch * 8)); CK(hipMalloc(&fb, batch * 8));
    CK(hipMalloc(&ca, batch * 8)); CK(hipMalloc(&cb, batch * 8));
    for (int32_t **p : {&ia, &ib, &oa, &ob, &sa, &sb, &nfa, &nfb, &nga, &ngb}) CK(hipMalloc(p, batch * 4));
    CK(hipMalloc(&pd, sizeof hp));
    CK(hipMemcpy(pd, &hp, sizeof hp, hipMemcpyHostToDevice));
    if (fl_synth_diag_spectrum(9, batch, n, 2.0, 10.0, d, nullptr) != FL_OK || fl_synth_uniform(9, batch, n, -1.0, 1.0, b, nullptr) != FL_OK ||
        fl_synth_uniform(11, batch, n, 0.05, 0.15, xa, nullptr) != FL_OK)
        return 3;
    CK(hipMemcpy(xb, xa, N * 8, hipMemcpyDeviceToDevice));
    CK(hipMemset(la, 0, batch * m * 8)); CK(hipMemset(lb, 0, batch * m * 8));
    fl_options o;
    fl_default_options(&o, FL_SOLVER_LBFGS);
    o.precision = 1e-9;
    const size_t wsb = fl_workspace_bytes_for(FL_SOLVER_LBFGS, batch, n, &o);
    void *wa, *wb;
    CK(hipMalloc(&wa, wsb)); CK(hipMalloc(&wb, wsb));
    int rc = fl_augmented_lagrangian_batched(FL_SOLVER_LBFGS, FL_OBJ_DIAGQUAD, batch, n, m, xa, d, b, la, 1.0, &o, wa, wsb, fa, ca, ia, oa, sa, nfa, nga, nullptr);
    if (rc != FL_OK) return 4;
    rc = fl::user::solve_auglag<1, 8>(FL_SOLVER_LBFGS, batch, n, m, xb, d, b, pd, lb, 1.0, &o, wb, wsb, fb, cb, ib, ob, sb, nfb, ngb, nullptr);
    if (rc != FL_OK) { std::printf("fl::user::solve_auglag failed %d\n", rc); return 5; }
    CK(hipDeviceSynchronize());
    std::vector<double> ha(N), hb(N), hla(batch * m), hlb(batch * m), hca(batch), hcb(batch);
    std::vector<int32_t> hna(batch), hnb(batch), hoa(batch), hob(batch), hsa(batch);
    CK(hipMemcpy(ha.data(), xa, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), xb, N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hla.data(), la, batch * m * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hlb.data(), lb, batch * m * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hca.data(), ca, batch * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hcb.data(), cb, batch * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hna.data(), nfa, batch * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hnb.data(), nfb, batch * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hoa.data(), oa, batch * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hob.data(), ob, batch * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hsa.data(), sa, batch * 4, hipMemcpyDeviceToHost));
    long nf = 0;
    bool conv = true;
    for (int k = 0; k < batch; ++k) { nf += hna[k]; conv = conv && hsa[k] == 0; }
    const bool same = std::memcmp(ha.data(), hb.data(), N * 8) == 0 && std::memcmp(hla.data(), hlb.data(), batch * m * 8) == 0 &&
                      std::memcmp(hca.data(), hcb.data(), batch * 8) == 0 && hna == hnb && hoa == hob;
    std::printf("AugmentedLagrangian n=%d m=%d batch=%d: %ld objective evaluations, user objective + user constraints %s the built-in ones\n", n, m,
                batch, nf, same ? "reproduce bit for bit" : "DIFFER from");
    return same && conv && nf > 0 ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad |= aug_case();
    bad |= one_case<2, 8>(FL_SOLVER_LBFGS, "LBFGS", 512, 1024, 1e-6);
    bad |= one_case<1, 4>(FL_SOLVER_LBFGS, "LBFGS", 256, 256, 1e-7);
    bad |= one_case<1, 4>(FL_SOLVER_CG, "ConjugateGradient", 256, 256, 1e-7);
    bad |= one_case<1, 4>(FL_SOLVER_SD, "SteepestDescent", 64, 200, 1e-4);
    bad |= one_case<1, 4>(FL_SOLVER_BFGS, "BFGS", 32, 256, 1e-7);
    std::printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
