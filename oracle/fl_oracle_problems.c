/*
 * fl_oracle_problems.c -- synthetic objectives / constraints and the OpenMP
 * batched driver of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see fl_oracle.h).
 *
 * Objectives (BASELINE.json configs):
 *   FLO_QUARTIC    f = sum x_i^4            (reference test/test.f90:630-663)
 *   FLO_ROSENBROCK f = sum_{i<n} 100 (x_{i+1}-x_i^2)^2 + (1-x_i)^2   (chained)
 *   FLO_DIAGQUAD   f = 1/2 sum d_i x_i^2 - sum b_i x_i
 * Constraint family: block spheres c_j = sum_{i in block j} x_i^2 - 1, M blocks
 * of N/M (M = 1 is the reference test's unit sphere, test/test.f90:699-721).
 * In FLO_SUM_SEQ mode sums run left to right exactly like the survey's probe
 * drivers; in FLO_SUM_TREE mode they use the HIP kernels' reduction order.
 */
#include "fl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

extern double flo_tree_sum(int n, const double *term);
extern int flo_get_sum_mode(void);
#define p_mode (flo_get_sum_mode())

static double sum_terms(int n, const double *t)
{
    if (p_mode == FLO_SUM_TREE) return flo_tree_sum(n, t);
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + t[i];
    return s;
}

void flo_prob_f(double *fx, const double *x, int n, void *ctx)
{
    const flo_problem *P = (const flo_problem *)ctx;
    double *t = (double *)malloc(sizeof(double) * 2 * (size_t)n), *t2 = t + n;
    if (P->kind == FLO_QUARTIC) {
        for (int i = 0; i < n; ++i) t[i] = x[i] * x[i] * x[i] * x[i]; /* x**4 = ((x*x)*x)*x, as flang -O2 expands it */
        *fx = sum_terms(n, t);
    } else if (P->kind == FLO_ROSENBROCK) {
        if (p_mode == FLO_SUM_SEQ) { /* fx=fx+100*(x(i+1)-x(i)**2)**2+(1-x(i))**2 */
            double s = 0.0;
            for (int i = 0; i < n - 1; ++i) {
                double u = x[i + 1] - x[i] * x[i], v = 1.0 - x[i];
                s = s + 100.0 * (u * u) + v * v;
            }
            *fx = s;
        } else {
            for (int i = 0; i < n; ++i) t[i] = 0.0;
            for (int i = 0; i < n - 1; ++i) {
                double u = x[i + 1] - x[i] * x[i], v = 1.0 - x[i];
                t[i] = 100.0 * (u * u) + v * v;
            }
            *fx = sum_terms(n, t);
        }
    } else { /* 0.5*sum(d*x*x)-sum(b*x) */
        for (int i = 0; i < n; ++i) {
            t[i] = P->d[i] * x[i] * x[i];
            t2[i] = P->b[i] * x[i];
        }
        *fx = 0.5 * sum_terms(n, t) - sum_terms(n, t2);
    }
    free(t);
}

void flo_prob_fd(double *g, const double *x, int n, void *ctx)
{
    const flo_problem *P = (const flo_problem *)ctx;
    if (P->kind == FLO_QUARTIC) {
        for (int i = 0; i < n; ++i) g[i] = 4.0 * (x[i] * x[i] * x[i]); /* 4*x**3 */
    } else if (P->kind == FLO_ROSENBROCK) {
        for (int i = 0; i < n; ++i) g[i] = 0.0;
        for (int i = 0; i < n - 1; ++i) {
            double u = x[i + 1] - x[i] * x[i];
            g[i] = g[i] - 400.0 * x[i] * u - 2.0 * (1.0 - x[i]);
            g[i + 1] = g[i + 1] + 200.0 * u;
        }
    } else {
        for (int i = 0; i < n; ++i) g[i] = P->d[i] * x[i] - P->b[i];
    }
}

int flo_prob_ffd(double *fx, double *g, const double *x, int n, void *ctx)
{
    flo_prob_f(fx, x, n, ctx);
    flo_prob_fd(g, x, n, ctx);
    return 0;
}

int flo_prob_fdd(double *H, const double *x, int n, void *ctx)
{
    const flo_problem *P = (const flo_problem *)ctx;
    memset(H, 0, sizeof(double) * (size_t)n * n);
#define H_(i, j) H[(size_t)(j) * n + (i)]
    if (P->kind == FLO_QUARTIC) {
        for (int i = 0; i < n; ++i) H_(i, i) = 12.0 * x[i] * x[i];
    } else if (P->kind == FLO_ROSENBROCK) {
        for (int i = 0; i < n - 1; ++i) {
            H_(i, i) += 1200.0 * x[i] * x[i] - 400.0 * x[i + 1] + 2.0;
            H_(i + 1, i + 1) += 200.0;
            H_(i, i + 1) += -400.0 * x[i];
            H_(i + 1, i) += -400.0 * x[i];
        }
    } else {
        for (int i = 0; i < n; ++i) H_(i, i) = P->d[i];
    }
#undef H_
    return 0;
}

/* block spheres: c_j = dot(x_blk, x_blk) - 1 */
void flo_prob_c(double *cx, const double *x, int m, int n, void *ctx)
{
    (void)ctx;
    const int w = n / m;
    for (int j = 0; j < m; ++j) {
        if (p_mode == FLO_SUM_TREE) {
            /* GPU order: every block sum is a masked full-width tree reduction */
            double *t = (double *)calloc((size_t)n, sizeof(double));
            for (int i = j * w; i < (j + 1) * w; ++i) t[i] = x[i] * x[i];
            cx[j] = flo_tree_sum(n, t) - 1.0;
            free(t);
        } else {
            double s = 0.0;
            for (int i = j * w; i < (j + 1) * w; ++i) s = s + x[i] * x[i];
            cx[j] = s - 1.0;
        }
    }
}

void flo_prob_cd(double *cdx, const double *x, int m, int n, void *ctx)
{
    (void)ctx;
    const int w = n / m;
    memset(cdx, 0, sizeof(double) * (size_t)n * m);
    for (int j = 0; j < m; ++j)
        for (int i = j * w; i < (j + 1) * w; ++i) cdx[(size_t)j * n + i] = 2.0 * x[i];
}

/* c''_j = 2 I on block j: cddx(N,N,M) column-major */
int flo_prob_cdd(double *cddx, const double *x, int m, int n, void *ctx)
{
    (void)ctx;
    (void)x;
    const int w = n / m;
    memset(cddx, 0, sizeof(double) * (size_t)n * n * m);
    for (int j = 0; j < m; ++j)
        for (int i = j * w; i < (j + 1) * w; ++i) cddx[(size_t)j * n * n + (size_t)i * n + i] = 2.0;
    return 0;
}

static void set_modes(int sum_mode, int threads, int ept) { flo_set_sum_mode(sum_mode, threads, ept); }

int flo_solve_batch(int solver, int kind, int B, int n, double *x, const double *d, const double *b,
                    const flo_opts *o, int use_ffd, int bfgs_form, int sum_mode, int threads, int ept,
                    int nthreads, double *fout, int *iters, int *status, int *nf, int *ng, double *gg)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < B; ++k) {
        flo_problem P;
        flo_stats st;
        set_modes(sum_mode, threads, ept);
        P.kind = kind;
        P.d = d ? d + (size_t)k * n : NULL;
        P.b = b ? b + (size_t)k * n : NULL;
        double *xk = x + (size_t)k * n;
        flo_ffd_t ffd = use_ffd ? flo_prob_ffd : NULL;
        if (solver == FLO_SD)
            flo_steepest_descent(flo_prob_f, flo_prob_fd, ffd, xk, n, o, &P, &st);
        else if (solver == FLO_CG)
            flo_conjugate_gradient(flo_prob_f, flo_prob_fd, ffd, xk, n, o, &P, &st);
        else if (solver == FLO_LBFGS)
            flo_lbfgs(flo_prob_f, flo_prob_fd, ffd, xk, n, o, &P, &st);
        /* bfgs_form + 4096: the caller passes no fdd -- the reference then differentiates f' by MKL's djacobi
         * (flo_central_hessian) where it wants a Hessian */
        else if (solver == 4 /* NewtonRaphson */)
            flo_newton(flo_prob_f, flo_prob_fd, ffd, (bfgs_form & 4096) ? NULL : flo_prob_fdd, xk, n, o, &P, &st);
        else
            flo_bfgs(flo_prob_f, flo_prob_fd, ffd, (o->exact_step > 0 && !(bfgs_form & 4096)) ? flo_prob_fdd : NULL, xk, n, o,
                     bfgs_form & 4095, &P, &st);
        if (fout) fout[k] = st.f;
        if (iters) iters[k] = st.iters;
        if (status) status[k] = st.status;
        if (nf) nf[k] = st.nf;
        if (ng) ng[k] = st.ng;
        if (gg) gg[k] = st.gg;
    }
    return used;
}

int flo_auglag_batch(int solver, int kind, int B, int n, int m, double *x, const double *d, const double *b,
                     double *lambda, double miu0, const flo_opts *o, int use_ffd, int sum_mode, int threads,
                     int ept, int nthreads, double *fout, int *iters, int *outer, int *nf, int *ng,
                     double *cnorm2)
{
    int used = 1;
    /* a checker must not crash on the arguments a test hands it (round 3, gpurun_out/r03_t13.log: a NULL fdd reached
     * flo_newton from here and every OpenMP thread called it): refuse what cannot be run, with a code */
    if (!x || !lambda || !o || B < 0 || n <= 0 || m <= 0 || m > n) return -1;
    if (kind == FLO_DIAGQUAD && (!d || !b)) return -1;
    if (solver != FLO_LBFGS && solver != FLO_CG && solver != FLO_BFGS && solver != 4) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int k = 0; k < B; ++k) {
        flo_problem P;
        flo_stats st;
        int out = 0;
        double cc = 0.0;
        set_modes(sum_mode, threads, ept);
        P.kind = kind;
        P.d = d ? d + (size_t)k * n : NULL;
        P.b = b ? b + (size_t)k * n : NULL;
        int as = solver == FLO_LBFGS ? 1 : (solver == FLO_CG ? 2 : (solver == 4 ? 3 : 0));
        /* NewtonRaphson, and BFGS with ExactStep > 0, take the analytic Hessians (the fdd / cdd branch) */
        /* (use_ffd & 2: the caller passes no fdd / cdd -- the reference then differentiates grad L by MKL's djacobi) */
        const int hess = ((as == 3) || (as == 0 && o->exact_step > 0)) && !(use_ffd & 2);
        flo_augmented_lagrangian_h(flo_prob_f, flo_prob_fd, (use_ffd & 1) ? flo_prob_ffd : NULL, hess ? flo_prob_fdd : NULL,
                                   flo_prob_c, flo_prob_cd, hess ? flo_prob_cdd : NULL, x + (size_t)k * n, n, m, as,
                                   lambda + (size_t)k * m, miu0, o, &P, &st, &out, &cc);
        /* objective (not Lagrangian) at the solution */
        double fx;
        flo_prob_f(&fx, x + (size_t)k * n, n, &P);
        if (fout) fout[k] = fx;
        if (iters) iters[k] = st.iters;
        if (outer) outer[k] = out;
        if (nf) nf[k] = st.nf;
        if (ng) ng[k] = st.ng;
        if (cnorm2) cnorm2[k] = cc;
    }
    return used;
}

/* f, grad f, c, cd of the built-in problems for a whole batch at the given points: what a caller of the
 * reverse-communication API computes between two steps (tests: the "user" side of fl_rci_step_auglag).
 * g [B][n], c [B][m], cd [B][m][n] (row j = grad c_j); any output may be NULL. */
int flo_prob_eval_batch(int kind, int B, int n, int m, const double *x, const double *d, const double *b, int sum_mode,
                        int threads, int ept, int nthreads, double *f, double *g, double *c, double *cd)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int k = 0; k < B; ++k) {
        flo_problem P;
        set_modes(sum_mode, threads, ept);
        P.kind = kind;
        P.d = d ? d + (size_t)k * n : NULL;
        P.b = b ? b + (size_t)k * n : NULL;
        const double *xk = x + (size_t)k * n;
        if (f) flo_prob_f(f + k, xk, n, &P);
        if (g) flo_prob_fd(g + (size_t)k * n, xk, n, &P);
        if (c) flo_prob_c(c + (size_t)k * m, xk, m, n, &P);
        if (cd) flo_prob_cd(cd + (size_t)k * m * n, xk, m, n, &P);
    }
    return used;
}
