// A caller-compiled objective for n > 4096 (include/fl_user_stream_objective.hpp): the diagonal quadratic restated as a
// STREAMING functor -- asked one element pair at a time by the vectors-in-HBM kernel -- must reproduce the built-in
// FL_OBJ_DIAGQUAD bit for bit (L-BFGS n = 6001, ConjugateGradient n = 5000, SteepestDescent n = 4500, quasi-Newton BFGS
// n = 4200).  Built and run by tests/test_gpu_user_objective.py on the GPU box; prints one line per case and "ALL OK".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../include/fl_user_stream_objective.hpp" // (first inclusion: fl::SolveArgs)

struct MyBigQuadratic {
    static constexpr bool NEIGHBOURS = false;
    const double *d, *b;
    __device__ void init(const fl::SolveArgs &A, int prob)
    {
        d = A.d + (size_t)prob * A.n;
        b = A.b + (size_t)prob * A.n;
    }
    __device__ void pair(int e, int n, const double *, double xa, double xb, double &ta, double &tb, double &ua, double &ub, double &ga,
                         double &gb)
    {
        const double da = e < n ? d[e] : 0.0, db = e + 1 < n ? d[e + 1] : 0.0;
        const double ba = e < n ? b[e] : 0.0, bb = e + 1 < n ? b[e + 1] : 0.0;
        const double dxa = da * xa, dxb = db * xb;
        ta = dxa * xa;
        tb = dxb * xb;
        ua = ba * xa;
        ub = bb * xb;
        ga = dxa - ba;
        gb = dxb - bb;
    }
    __device__ static double combine(double s0, double s1) { return 0.5 * s0 - s1; }
};
#define FL_USER_STREAM_OBJECTIVE MyBigQuadratic
#include "../include/fl_user_stream_objective.hpp" // (second inclusion: fl::user::solve_stream around the class)

#define CK(x)                                                                   \
    do {                                                                        \
        if ((x) != hipSuccess) {                                                \
            std::printf("HIP error at %s:%d\n", __FILE__, __LINE__);            \
            return 2;                                                           \
        }                                                                       \
    } while (0)

static int one_case(int solver, const char *name, int batch, int n, double precision)
{
    const size_t N = (size_t)batch * n;
    double *d, *b, *xa, *xb, *fa, *fb, *ga, *gb;
    int32_t *ia, *ib, *sa, *sb, *nfa, *nfb, *nga, *ngb;
    CK(hipMalloc(&d, N * 8)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&xa, N * 8)); CK(hipMalloc(&xb, N * 8));
    CK(hipMalloc(&fa, batch * 8)); CK(hipMalloc(&fb, batch * 8)); CK(hipMalloc(&ga, batch * 8)); CK(hipMalloc(&gb, batch * 8));
    for (int32_t **p : {&ia, &ib, &sa, &sb, &nfa, &nfb, &nga, &ngb}) CK(hipMalloc(p, batch * 4));
    if (fl_synth_diag_spectrum(7, batch, n, 10.0, 300.0, d, nullptr) != FL_OK) return 3;
    if (fl_synth_uniform(7, batch, n, -1.0, 1.0, b, nullptr) != FL_OK) return 3;
    CK(hipMemset(xa, 0, N * 8)); CK(hipMemset(xb, 0, N * 8));
    fl_options o;
    fl_default_options(&o, solver);
    o.precision = precision;
    o.exact_step = 0;
    if (solver == FL_SOLVER_SD || solver == FL_SOLVER_BFGS) o.max_iteration = 40;
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
    rc = fl::user::solve_stream(solver, batch, n, xb, d, b, nullptr, &o, wb, wsb, fb, gb, ib, sb, nfb, ngb, nullptr);
    if (rc != FL_OK) { std::printf("%s: fl::user::solve_stream failed %d\n", name, rc); return 5; }
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
    std::printf("%s n=%d batch=%d: %ld iterations, streaming functor %s the built-in objective\n", name, n, batch, it,
                same ? "reproduces bit for bit" : "DIFFERS from");
    // the register path's sizes are refused here
    if (fl::user::solve_stream(solver, batch, 1024, xb, d, b, nullptr, &o, wb, wsb, fb, gb, ib, sb, nfb, ngb, nullptr) != FL_ERR_UNSUPPORTED_SIZE)
        return 6;
    for (void *p : {(void *)d, (void *)b, (void *)xa, (void *)xb, (void *)fa, (void *)fb, (void *)ga, (void *)gb, (void *)ia, (void *)ib,
                    (void *)sa, (void *)sb, (void *)nfa, (void *)nfb, (void *)nga, (void *)ngb, wa, wb})
        if (p) (void)hipFree(p);
    return same && it > 0 ? 0 : 1;
}

int main()
{
    int bad = 0;
    bad |= one_case(FL_SOLVER_LBFGS, "LBFGS", 24, 6001, 1e-7);
    bad |= one_case(FL_SOLVER_CG, "ConjugateGradient", 16, 5000, 1e-7);
    bad |= one_case(FL_SOLVER_SD, "SteepestDescent", 8, 4500, 1e-4);
    bad |= one_case(FL_SOLVER_BFGS, "BFGS", 4, 4200, 1e-6);
    std::printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
