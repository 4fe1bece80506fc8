/*
 * fl_oracle.c -- CPU restatement of the reference's line-search optimisers.
 * TEST INFRASTRUCTURE ONLY (see fl_oracle.h).  Written from the semantics of
 * /root/reference/source/NonlinearOptimization.f90 ("NO.f90"); every routine
 * cites the lines it follows.  Arithmetic is ordered exactly as the Fortran
 * expressions evaluate (left to right, no FMA contraction: build with
 * -ffp-contract=off), including the reference's quirks:
 *   - StrongWolfe's "search for larger a" branch calls zoom and then keeps
 *     looping with fx=fx0 (NO.f90:1507-1514); the _fdwithf twin returns
 *     (NO.f90:1628-1632);
 *   - Wolfe_fdwithf never calls f_fd (NO.f90:1373-1459 is a copy of Wolfe);
 *   - L-BFGS' first line search and pre-iterations never use f_fd
 *     (NO.f90:448-460, 486-498);
 *   - no curvature safeguard on rho = 1/(y.s) (NO.f90:471, 509, 623).
 */
#include "fl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ sums */
static __thread int g_mode = FLO_SUM_SEQ;
static __thread int g_threads = 64;
static __thread int g_ept = 2;

void flo_set_sum_mode(int mode, int threads, int ept)
{
    g_mode = mode;
    g_threads = threads;
    g_ept = ept;
}

int flo_get_sum_mode(void) { return g_mode; }

/* element held by thread t in register slot k: 16-byte (2-double) chunks are
 * dealt round-robin over the threads -- the HIP kernels' coalesced layout */
static inline int elem_of(int t, int k, int T) { return (((k >> 1) * T + t) << 1) + (k & 1); }

/* groups > 1 (the cooperative form of the vectors-in-HBM kernels, csrc/fl_big.hpp): `groups` workgroups share one
 * problem; workgroup w owns the register slots [w*per, (w+1)*per) of every thread, per = ceil(slots / groups), sums them in
 * the usual order (thread, wave tree, waves left to right), and the workgroups' partial sums are added left to right.
 * A process-wide setting (flo_set_sum_groups), read by every summation in FLO_SUM_TREE mode. */
static int g_groups = 1;
void flo_set_sum_groups(int groups) { g_groups = groups > 1 ? groups : 1; }

static double tree_reduce_range(int n, const double *a, const double *b, int k_lo, int k_hi)
{
    const int T = g_threads, NW = T / 64;
    double wave[64]; /* up to 16 waves */
    double lane[64], tmp[64];
    for (int w = 0; w < NW; ++w) {
        for (int l = 0; l < 64; ++l) {
            const int t = w * 64 + l;
            double acc = 0.0;
            for (int k = k_lo; k < k_hi; ++k) {
                const int e = elem_of(t, k, T);
                double term = 0.0;
                if (e < n) term = b ? a[e] * b[e] : a[e];
                acc = (k == k_lo) ? term : acc + term;
            }
            lane[l] = acc;
        }
        /* the kernels' wave tree (csrc/fl_reduce.hpp): halves, quarter rows, then mirror steps inside a row */
        for (int l = 0; l < 32; ++l) tmp[l] = lane[l] + lane[l + 32];
        for (int l = 0; l < 16; ++l) lane[l] = tmp[l] + tmp[l + 16];
        for (int i = 0; i < 8; ++i) tmp[i] = lane[i] + lane[15 - i];
        for (int i = 0; i < 4; ++i) lane[i] = tmp[i] + tmp[7 - i];
        tmp[0] = lane[0] + lane[2];
        tmp[1] = lane[1] + lane[3];
        wave[w] = tmp[0] + tmp[1];
    }
    double tot = wave[0];
    for (int w = 1; w < NW; ++w) tot = tot + wave[w];
    return tot;
}
static double tree_reduce(int n, const double *a, const double *b)
{
    const int E = g_ept;
    if (g_groups <= 1) return tree_reduce_range(n, a, b, 0, E);
    const int slots = E / 2, per = (slots + g_groups - 1) / g_groups;
    double tot = 0.0;
    for (int w = 0; w * per < slots; ++w) {
        const int c_hi = (w + 1) * per < slots ? (w + 1) * per : slots;
        const double part = tree_reduce_range(n, a, b, 2 * w * per, 2 * c_hi);
        tot = (w == 0) ? part : tot + part;
    }
    return tot;
}

double flo_tree_sum(int n, const double *term) { return tree_reduce(n, term, NULL); }

/* dot_product intrinsic (NO.f90:442 and everywhere): sequential, or GPU order */
double flo_dot(int n, const double *a, const double *b)
{
    if (g_mode == FLO_SUM_TREE) return tree_reduce(n, a, b);
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

static double seq_dot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

void flo_defaults(flo_opts *o)
{
    o->strong = 1;
    o->maxit = 1000;
    o->precision = 1e-15;
    o->minstep = 1e-15;
    o->c1 = 1e-4;
    o->c2 = 0.9;
    o->increment = 1.05;
    o->memory = 10;
    o->exact_step = 20;
    o->method = 0;
    o->clamp = 1;
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* ----------------------------------------------------------- line search */
typedef struct {
    double c1, c2abs, fx0, phid0;
    flo_f_t f;
    flo_fd_t fd;
    flo_ffd_t f_fd;
    void *ctx;
    flo_stats *st;
    int n;
    double *x, *x0, *fdx, *a, *fx;
    const double *p;
} ls_t;

static void ls_setx(ls_t *L, double a) /* x=x0+a*p */
{
    for (int i = 0; i < L->n; ++i) L->x[i] = L->x0[i] + a * L->p[i];
}
static void ls_f(ls_t *L)
{
    L->f(L->fx, L->x, L->n, L->ctx);
    L->st->nf++;
}
static void ls_fd(ls_t *L)
{
    L->fd(L->fdx, L->x, L->n, L->ctx);
    L->st->ng++;
}
static void ls_both(ls_t *L, int fused)
{
    if (fused) {
        L->f_fd(L->fx, L->fdx, L->x, L->n, L->ctx);
        L->st->nf++;
        L->st->ng++;
    } else {
        ls_f(L);
        ls_fd(L);
    }
}
static int armijo_ok(const ls_t *L) { return *L->fx <= L->fx0 + L->c1 * (*L->a) * L->phid0; }

/* zoom of Wolfe, NO.f90:1347-1370: quadratic interpolation, f-only trials */
static void w_zoom(ls_t *L, double *low, double *up, double *flow, double *fup, double *phidlow)
{
    double *a = L->a, phidnew, phidlow_m_a = *phidlow * *a;
    for (;;) {
        *a = phidlow_m_a * *a / 2.0 / (*flow + phidlow_m_a - *fup);
        if (!(*a > *low && *a < *up)) *a = (*low + *up) / 2.0;
        ls_setx(L, *a);
        ls_f(L);
        if (*L->fx > L->fx0 + L->c1 * *a * L->phid0) {
            *up = *a;
            if (*up - *low < 1e-15 || (*up - *low) / dmax(fabs(*low), fabs(*up)) < 1e-15) {
                ls_fd(L);
                return;
            }
            *fup = *L->fx;
        } else {
            ls_fd(L);
            phidnew = flo_dot(L->n, L->fdx, L->p);
            if (phidnew > L->c2abs) return;
            *low = *a;
            if (*up - *low < 1e-15 || (*up - *low) / dmax(fabs(*low), fabs(*up)) < 1e-15) return;
            *flow = *L->fx;
            *phidlow = phidnew;
            phidlow_m_a = *phidlow * *a;
        }
    }
}

/* Wolfe, NO.f90:1286-1371 (Wolfe_fdwithf 1373-1459 is the same code) */
void flo_wolfe(double c1, double c2, flo_f_t f, flo_fd_t fd, double *x, double *a, const double *p, double *fx,
               double phid0, double *fdx, int n, double increment, void *ctx, flo_stats *st)
{
    double incrmt = dmax(1.0 + 1e-15, increment);
    double ftemp, atemp, aold, fold, phidx;
    double *x0 = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(x0, x, sizeof(double) * (size_t)n);
    ls_t L = {c1, c2 * fabs(phid0), *fx, phid0, f, fd, NULL, ctx, st, n, x, x0, fdx, a, fx, p};
    ls_setx(&L, *a);
    ls_f(&L);
    if (armijo_ok(&L)) { /* search for larger a */
        for (;;) {
            aold = *a;
            fold = *fx;
            *a = aold * incrmt;
            ls_setx(&L, *a);
            ls_f(&L);
            if (*fx > L.fx0 + c1 * *a * phid0) {
                ls_setx(&L, aold);
                ls_fd(&L);
                phidx = flo_dot(n, fdx, p);
                if (phidx > L.c2abs) {
                    *a = aold;
                    *fx = fold;
                } else {
                    atemp = *a;
                    ftemp = *fx;
                    w_zoom(&L, &aold, &atemp, &fold, &ftemp, &phidx);
                }
                break;
            }
        }
    } else { /* search for smaller a */
        for (;;) {
            aold = *a;
            fold = *fx;
            *a = aold / incrmt;
            ls_setx(&L, *a);
            ls_f(&L);
            if (armijo_ok(&L)) {
                ls_fd(&L);
                phidx = flo_dot(n, fdx, p);
                if (phidx < L.c2abs) {
                    atemp = *a;
                    ftemp = *fx;
                    w_zoom(&L, &atemp, &aold, &ftemp, &fold, &phidx);
                }
                break;
            }
            if (*a < 1e-15) {
                ls_fd(&L);
                break;
            }
        }
    }
    free(x0);
}

/* zoom of StrongWolfe, NO.f90:1557-1579 (= 1675-1697): cubic interpolation */
static void sw_zoom(ls_t *L, int fused, double *low, double *up, double *flow, double *fup, double *phidlow,
                    double *phidup)
{
    double *a = L->a, phidnew, d1, d2;
    for (;;) {
        d1 = *phidlow + *phidup - 3.0 * (*flow - *fup) / (*low - *up);
        d2 = *up - *low;
        if (d2 > 0.0)
            d2 = sqrt(d1 * d1 - *phidlow * *phidup);
        else
            d2 = -sqrt(d1 * d1 - *phidlow * *phidup);
        *a = *up - (*up - *low) * (*phidup + d2 - d1) / (*phidup - *phidlow + 2.0 * d2);
        if (!(*a > dmin(*low, *up) && *a < dmax(*low, *up))) *a = (*low + *up) / 2.0;
        ls_setx(L, *a);
        ls_both(L, fused);
        phidnew = flo_dot(L->n, L->fdx, L->p);
        if (*L->fx > L->fx0 + L->c1 * *a * L->phid0 || *L->fx >= *flow) {
            *up = *a;
            *fup = *L->fx;
            *phidup = phidnew;
        } else {
            if (fabs(phidnew) <= L->c2abs) return;
            if (phidnew * (*up - *low) >= 0.0) {
                *up = *low;
                *fup = *flow;
                *phidup = *phidlow;
            }
            *low = *a;
            *flow = *L->fx;
            *phidlow = phidnew;
        }
        if (fabs(*up - *low) < 1e-15 || fabs(*up - *low) / dmax(fabs(*low), fabs(*up)) < 1e-15) return;
    }
}

/* StrongWolfe NO.f90:1462-1580 (f_fd == NULL) and StrongWolfe_fdwithf
 * NO.f90:1582-1698 (f_fd != NULL) */
void flo_strong_wolfe(double c1, double c2, flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, double *a,
                      const double *p, double *fx, double phid0, double *fdx, int n, double increment, void *ctx,
                      flo_stats *st)
{
    const int w = (f_fd != NULL);
    double incrmt = dmax(1.0 + 1e-15, increment);
    double ftemp, atemp, aold, fold, phidnew, phidold;
    double *x0 = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(x0, x, sizeof(double) * (size_t)n);
    ls_t L = {c1, c2 * fabs(phid0), *fx, phid0, f, fd, f_fd, ctx, st, n, x, x0, fdx, a, fx, p};
    ls_setx(&L, *a);
    if (w)
        ls_both(&L, 1);
    else
        ls_f(&L);
    if (armijo_ok(&L)) { /* satisfied, try to search for larger a */
        if (!w) ls_fd(&L);
        phidnew = flo_dot(n, fdx, p);
        if (phidnew > 0.0) { /* curve is heading up */
            if (fabs(phidnew) <= L.c2abs) goto done;
            for (;;) {
                aold = *a;
                fold = *fx;
                phidold = phidnew;
                *a = aold / incrmt;
                ls_setx(&L, *a);
                ls_both(&L, w);
                phidnew = flo_dot(n, fdx, p);
                if (*fx >= fold || phidnew <= 0.0) {
                    atemp = *a;
                    ftemp = *fx;
                    sw_zoom(&L, w, &aold, &atemp, &fold, &ftemp, &phidold, &phidnew);
                    goto done;
                }
                if (*a < 1e-15) goto done;
            }
        } else { /* search for larger a */
            for (;;) {
                aold = *a;
                fold = *fx;
                phidold = phidnew;
                *a = aold * incrmt;
                ls_setx(&L, *a);
                ls_both(&L, w);
                phidnew = flo_dot(n, fdx, p);
                if (*fx > L.fx0 + c1 * *a * phid0 || *fx >= fold) {
                    atemp = *a;
                    ftemp = *fx;
                    sw_zoom(&L, w, &aold, &atemp, &fold, &ftemp, &phidold, &phidnew);
                    goto done;
                }
                if (phidnew > 0.0) {
                    if (fabs(phidnew) <= L.c2abs) goto done;
                    atemp = *a;
                    ftemp = *fx;
                    sw_zoom(&L, w, &atemp, &aold, &ftemp, &fold, &phidnew, &phidold);
                    if (w) goto done; /* NO.f90:1632 returns */
                    *fx = L.fx0;      /* NO.f90:1512: no return, loop goes on */
                }
            }
        }
    } else { /* violated, first search for smaller a satisfying sufficient decrease */
        for (;;) {
            aold = *a;
            fold = *fx;
            *a = aold / incrmt;
            ls_setx(&L, *a);
            ls_f(&L);
            if (armijo_ok(&L)) {
                ls_fd(&L);
                phidnew = flo_dot(n, fdx, p);
                if (fabs(phidnew) <= L.c2abs) goto done;
                if (phidnew < 0.0) { /* within [a, aold] */
                    ls_setx(&L, aold);
                    ls_fd(&L);
                    phidold = flo_dot(n, fdx, p);
                    atemp = *a;
                    ftemp = *fx;
                    sw_zoom(&L, w, &atemp, &aold, &ftemp, &fold, &phidnew, &phidold);
                    goto done;
                } else {
                    for (;;) {
                        aold = *a;
                        fold = *fx;
                        phidold = phidnew;
                        *a = aold / incrmt;
                        ls_setx(&L, *a);
                        ls_both(&L, w);
                        phidnew = flo_dot(n, fdx, p);
                        if (*fx >= fold || phidnew <= 0.0) {
                            atemp = *a;
                            ftemp = *fx;
                            sw_zoom(&L, w, &aold, &atemp, &fold, &ftemp, &phidold, &phidnew);
                            goto done;
                        }
                        if (*a < 1e-15) goto done;
                    }
                }
            }
            if (*a < 1e-15) {
                ls_fd(&L);
                goto done;
            }
        }
    }
done:
    free(x0);
}

/* one line search, chosen like the reference's main loops choose it */
static void line_search(const flo_opts *o, double c1, double c2, int fused, flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd,
                        double *x, double *a, const double *p, double *fx, double phid0, double *fdx, int n,
                        void *ctx, flo_stats *st)
{
    if (o->strong)
        flo_strong_wolfe(c1, c2, f, fd, fused ? f_fd : NULL, x, a, p, fx, phid0, fdx, n, o->increment, ctx, st);
    else
        flo_wolfe(c1, c2, f, fd, x, a, p, fx, phid0, fdx, n, o->increment, ctx, st);
    st->iters++;
}

static void clamp_c(const flo_opts *o, double *c1, double *c2)
{
    *c1 = o->c1;
    *c2 = o->c2;
    if (o->clamp) { /* NO.f90:83-86 */
        *c1 = dmax(1e-15, o->c1);
        *c2 = dmin(1.0 - 1e-15, dmax(*c1 + 1e-15, o->c2));
    }
}

static void initial_eval(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *fnew, double *g, const double *x, int n,
                         void *ctx, flo_stats *st)
{
    if (f_fd)
        f_fd(fnew, g, x, n, ctx);
    else {
        f(fnew, x, n, ctx);
        fd(g, x, n, ctx);
    }
    st->nf++;
    st->ng++;
}

static void st_zero(flo_stats *st)
{
    st->status = FLO_CONVERGED;
    st->iters = st->nf = st->ng = 0;
    st->f = 0.0;
    st->gg = 0.0;
}

/* SteepestDescent, NO.f90:55-188 */
void flo_steepest_descent(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o, void *ctx,
                          flo_stats *st)
{
    double tol = o->precision * o->precision, minstep = o->minstep * o->minstep, c1, c2;
    double a, fnew, phidnew, phidold;
    double *p = (double *)malloc(sizeof(double) * 2 * (size_t)n), *g = p + n;
    clamp_c(o, &c1, &c2);
    st_zero(st);
    initial_eval(f, fd, f_fd, &fnew, g, x, n, ctx, st);
    for (int i = 0; i < n; ++i) p[i] = -g[i];
    phidnew = -flo_dot(n, g, g);
    st->f = fnew;
    st->gg = -phidnew;
    if (-phidnew < tol) goto out;
    a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(-phidnew);
    st->status = FLO_MAXIT;
    for (int it = 1; it <= o->maxit; ++it) {
        phidold = phidnew;
        line_search(o, c1, c2, f_fd != NULL, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        /* After(), NO.f90:171-187 */
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) {
            st->status = FLO_CONVERGED;
            break;
        }
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            break;
        }
        for (int i = 0; i < n; ++i) p[i] = -g[i];
        phidnew = -flo_dot(n, g, g);
        a = a * phidold / phidnew;
    }
out:
    free(p);
}

/* ConjugateGradient, NO.f90:193-394; DY() 352-372, PR() 373-393.
 * (ConjugateGradient_basic NO.f90:2249-2346 = the same with clamp=0, f_fd=NULL) */
void flo_conjugate_gradient(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o, void *ctx,
                            flo_stats *st)
{
    double tol = o->precision * o->precision, minstep = o->minstep * o->minstep, c1, c2;
    double a, fnew, phidnew, phidold;
    double *p = (double *)malloc(sizeof(double) * 4 * (size_t)n), *g = p + n, *gold = g + n, *tmp = gold + n;
    flo_opts oo = *o;
    if (o->method == 1) oo.strong = 1; /* PR always uses the strong Wolfe searcher (NO.f90:311-344) */
    clamp_c(o, &c1, &c2);
    st_zero(st);
    initial_eval(f, fd, f_fd, &fnew, g, x, n, ctx, st);
    for (int i = 0; i < n; ++i) p[i] = -g[i];
    phidnew = -flo_dot(n, g, g);
    st->f = fnew;
    st->gg = -phidnew;
    if (-phidnew < tol) goto out;
    a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(-phidnew);
    st->status = FLO_MAXIT;
    for (int it = 1; it <= o->maxit; ++it) {
        memcpy(gold, g, sizeof(double) * (size_t)n);
        phidold = phidnew;
        line_search(&oo, c1, c2, f_fd != NULL, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) {
            st->status = FLO_CONVERGED;
            break;
        }
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            break;
        }
        double beta;
        if (o->method == 0) { /* DY: p=-g+(g.g)/((g-gold).p)*p */
            for (int i = 0; i < n; ++i) tmp[i] = g[i] - gold[i];
            beta = flo_dot(n, g, g) / flo_dot(n, tmp, p);
        } else { /* PR: p=-g+(g.(g-gold))/(gold.gold)*p */
            for (int i = 0; i < n; ++i) tmp[i] = g[i] - gold[i];
            beta = flo_dot(n, g, tmp) / flo_dot(n, gold, gold);
        }
        for (int i = 0; i < n; ++i) p[i] = -g[i] + beta * p[i];
        phidnew = flo_dot(n, g, p);
        if (phidnew > 0.0) { /* ascent direction: reset to steepest descent */
            for (int i = 0; i < n; ++i) p[i] = -g[i];
            phidnew = -flo_dot(n, g, g);
        }
        a = a * phidold / phidnew;
    }
out:
    free(p);
}

/* LBFGS, NO.f90:398-625; Before() 586-608, After() 609-624 */
void flo_lbfgs(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o, void *ctx,
               flo_stats *st)
{
    const int mem = o->memory > 1 ? o->memory : 1;
    double tol = o->precision * o->precision, minstep = o->minstep * o->minstep, c1, c2;
    double a, fnew, phidnew;
    size_t N = (size_t)n;
    double *p = (double *)malloc(sizeof(double) * (4 * N + 2 * N * (size_t)(mem + 1) + 2 * (size_t)(mem + 1)));
    double *g = p + N, *xold = g + N, *gold = xold + N, *s = gold + N, *y = s + N * (size_t)(mem + 1);
    double *rho = y + N * (size_t)(mem + 1), *alpha = rho + (mem + 1);
    int recent, i;
    clamp_c(o, &c1, &c2);
    st_zero(st);
    initial_eval(f, fd, f_fd, &fnew, g, x, n, ctx, st);
    for (i = 0; i < n; ++i) p[i] = -g[i];
    phidnew = -flo_dot(n, g, g);
    st->f = fnew;
    st->gg = -phidnew;
    if (-phidnew < tol) goto out;
    a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(-phidnew);
    memcpy(xold, x, sizeof(double) * N);
    memcpy(gold, g, sizeof(double) * N);
    /* first line search never uses f_fd (NO.f90:448-460) */
    line_search(o, c1, c2, 0, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
    phidnew = flo_dot(n, g, g);
    st->f = fnew;
    st->gg = phidnew;
    if (phidnew < tol) goto out;
    if (flo_dot(n, p, p) * a * a < minstep) {
        st->status = FLO_STEP_CONVERGED;
        goto out;
    }
    recent = 0;
    for (i = 0; i < n; ++i) {
        s[i] = x[i] - xold[i];
        y[i] = g[i] - gold[i];
    }
    rho[0] = 1.0 / flo_dot(n, y, s);
    for (int it = 1; it <= mem - 1; ++it) { /* pre-iterate to get enough history, NO.f90:472-510 */
        memcpy(xold, x, sizeof(double) * N);
        memcpy(gold, g, sizeof(double) * N);
        memcpy(p, g, sizeof(double) * N);
        for (i = recent; i >= 0; --i) {
            alpha[i] = rho[i] * flo_dot(n, s + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] - alpha[i] * y[N * i + k];
        }
        {
            double yy = flo_dot(n, y + N * recent, y + N * recent);
            for (int k = 0; k < n; ++k) p[k] = p[k] / rho[recent] / yy;
        }
        for (i = 0; i <= recent; ++i) {
            phidnew = rho[i] * flo_dot(n, y + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] + (alpha[i] - phidnew) * s[N * i + k];
        }
        for (int k = 0; k < n; ++k) p[k] = -p[k];
        phidnew = flo_dot(n, g, p);
        a = 1.0;
        line_search(o, c1, c2, 0, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) goto out;
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            goto out;
        }
        recent = recent + 1;
        for (int k = 0; k < n; ++k) {
            s[N * recent + k] = x[k] - xold[k];
            y[N * recent + k] = g[k] - gold[k];
        }
        rho[recent] = 1.0 / flo_dot(n, y + N * recent, s + N * recent);
    }
    st->status = FLO_MAXIT;
    for (int it = 1; it <= o->maxit; ++it) {
        /* Before() */
        memcpy(xold, x, sizeof(double) * N);
        memcpy(gold, g, sizeof(double) * N);
        memcpy(p, g, sizeof(double) * N);
        for (i = recent; i >= 0; --i) {
            alpha[i] = rho[i] * flo_dot(n, s + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] - alpha[i] * y[N * i + k];
        }
        for (i = mem - 1; i >= recent + 1; --i) {
            alpha[i] = rho[i] * flo_dot(n, s + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] - alpha[i] * y[N * i + k];
        }
        {
            double yy = flo_dot(n, y + N * recent, y + N * recent);
            for (int k = 0; k < n; ++k) p[k] = p[k] / rho[recent] / yy;
        }
        for (i = recent + 1; i <= mem - 1; ++i) {
            phidnew = rho[i] * flo_dot(n, y + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] + (alpha[i] - phidnew) * s[N * i + k];
        }
        for (i = 0; i <= recent; ++i) {
            phidnew = rho[i] * flo_dot(n, y + N * i, p);
            for (int k = 0; k < n; ++k) p[k] = p[k] + (alpha[i] - phidnew) * s[N * i + k];
        }
        for (int k = 0; k < n; ++k) p[k] = -p[k];
        phidnew = flo_dot(n, g, p);
        a = 1.0;
        line_search(o, c1, c2, f_fd != NULL, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        /* After() */
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) {
            st->status = FLO_CONVERGED;
            break;
        }
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            break;
        }
        recent = (recent + 1) % mem;
        for (int k = 0; k < n; ++k) {
            s[N * recent + k] = x[k] - xold[k];
            y[N * recent + k] = g[k] - gold[k];
        }
        rho[recent] = 1.0 / flo_dot(n, y + N * recent, s + N * recent);
    }
out:
    free(p);
}

/* ---------------------------------------------------------------- dense */
/* dsyL2U, LA.f90:260-265: mirror the strictly lower triangle to the upper */
void flo_syL2U(double *A, int n)
{
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A[(size_t)j * n + i] = A[(size_t)i * n + j];
}

/* My_dpotri, LA.f90:798-812: dpotrf('L') then dpotri('L') (LAPACK semantics,
 * unblocked restatement; rounding may differ from MKL's blocked kernels) */
int flo_dpotri_lower(double *A, int n)
{
#define A_(i, j) A[(size_t)(j) * n + (i)]
    for (int j = 0; j < n; ++j) { /* Cholesky, lower */
        double ajj = A_(j, j);
        for (int k = 0; k < j; ++k) ajj = ajj - A_(j, k) * A_(j, k);
        if (!(ajj > 0.0)) return j + 1;
        ajj = sqrt(ajj);
        A_(j, j) = ajj;
        for (int i = j + 1; i < n; ++i) {
            double v = A_(i, j);
            for (int k = 0; k < j; ++k) v = v - A_(i, k) * A_(j, k);
            A_(i, j) = v / ajj;
        }
    }
    /* W = inv(L) by rows: W(j,:) = (e_j - sum_{k<j} L(j,k) W(k,:)) / L(j,j); then A^{-1} = W^T W.
     * Every sum runs over k in ascending order (what the HIP routines of fl_dense.hpp compute, so the
     * two agree bit for bit); both triangles are filled, which makes the reference's syL2U a no-op. */
    {
        double *W = (double *)calloc((size_t)n * n, sizeof(double)); /* W[j*n + c] = W(j,c) */
        for (int j = 0; j < n; ++j) {
            for (int c = 0; c < n; ++c) {
                double v = (c == j) ? 1.0 : 0.0;
                for (int k = 0; k < j; ++k) v = v - A_(j, k) * W[(size_t)k * n + c];
                W[(size_t)j * n + c] = v / A_(j, j);
            }
        }
        for (int b = 0; b < n; ++b)
            for (int a = 0; a < n; ++a) {
                double v = 0.0;
                for (int k = 0; k < n; ++k) v = v + W[(size_t)k * n + a] * W[(size_t)k * n + b];
                A_(a, b) = v;
            }
        free(W);
    }
#undef A_
    return 0;
}

/* My_dposv, LA.f90:719-730: dposv('L'): Cholesky factor in A (lower), b <- A^{-1} b; b untouched if it
 * fails.  Forward substitution column-oriented (dtrsv 'L','N'); backward x_j = (z_j - sum_{i>j} L(i,j) x_i)
 * / L(j,j) with the sum taken top-down in FLO_SUM_SEQ (reference BLAS order is bottom-up: same value to
 * rounding) and in the kernels' reduction order in FLO_SUM_TREE. */
/* My_dsysv (LA.f90:695-703: dsysv 'L', symmetric indefinite): the kernels' restatement -- the matrix the lower
 * triangle defines, Gaussian elimination with partial pivoting (largest |a|, ties to the smaller row index), rows
 * never moved, right-hand side carried along, back substitution in axpy form (csrc/fl_dense_kernels.hip
 * dsysv_kernel).  LAPACK's Bunch-Kaufman factorisation inside MKL gives the same solution to rounding.
 * Returns info (0, or k+1 when no non-zero pivot is left at step k; b untouched then). */
int flo_dsysv(double *A, double *b, int n)
{
    size_t N = (size_t)n;
    int *pstep = (int *)malloc(sizeof(int) * 2 * N), *piv = pstep + N;
    double *w = (double *)malloc(sizeof(double) * N);
    int info = 0;
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A[j * N + i] = A[i * N + j];
    memcpy(w, b, sizeof(double) * N);
    for (int i = 0; i < n; ++i) pstep[i] = n;
    for (int k = 0; k < n && info == 0; ++k) {
        double best = -1.0;
        int p = -1;
        for (int i = 0; i < n; ++i)
            if (pstep[i] == n && fabs(A[k * N + i]) > best) {
                best = fabs(A[k * N + i]);
                p = i;
            }
        if (!(best > 0.0)) {
            info = k + 1;
            break;
        }
        piv[k] = p;
        pstep[p] = k;
        const double apk = A[k * N + p], bp = w[p];
        for (int i = 0; i < n; ++i) {
            if (pstep[i] != n) continue;
            const double l = A[k * N + i] / apk;
            w[i] = w[i] - l * bp;
            if (l != 0.0)
                for (int j = k + 1; j < n; ++j) A[j * N + i] = A[j * N + i] - l * A[j * N + p];
        }
    }
    if (info == 0) {
        for (int k = n - 1; k >= 0; --k) {
            const int p = piv[k];
            const double xk = w[p] / A[k * N + p];
            for (int i = 0; i < n; ++i)
                if (pstep[i] < k) w[i] = w[i] - A[k * N + i] * xk;
            b[k] = xk;
        }
    }
    free(w);
    free(pstep);
    return info;
}

/* LagrangianMultiplier, NO.f90:1950-1993: Newton iteration on the KKT system of L = f - lambda.c.
 * -L' = [C lambda - f'; c],  L'' = [f'' - sum_k lambda_k c''_k, .; -C^T, 0] (lower triangle), solved by My_dsysv.
 * The assembly and the convergence test are plain sequential sums (they run next to the callbacks on the host in
 * the MI355X build), the solve is flo_dsysv.  Returns the number of Newton steps taken. */
int flo_lagrangian_multiplier(flo_fd_t fd, flo_fdd_t fdd, flo_c_t c, flo_cd_t cd, flo_cdd_t cdd, double *x,
                              double *lambda, int n, int m, int maxit, double precision, void *ctx)
{
    if (!fd || !fdd || !c || !cd || !cdd || !x) return -1; /* all five callbacks are mandatory in the reference too (NO.f90:1950-1952) */
    const int dim = n + m;
    size_t N = (size_t)n, D = (size_t)dim;
    const double tol = precision * precision;
    double *mLd = (double *)malloc(sizeof(double) * (D + m + N * m + N * N * m + D * D));
    double *cx = mLd + D, *cdx = cx + m, *cddx = cdx + N * m, *Ldd = cddx + N * N * m;
    int it = 0;
    for (int iter = 1; iter <= maxit; ++iter) {
        fd(mLd, x, n, ctx);
        c(cx, x, m, n, ctx);
        cd(cdx, x, m, n, ctx);
        for (int i = 0; i < n; ++i) { /* minusLd(1:N)=matmul(cdx,lambda)-minusLd(1:N) */
            double t = 0.0;
            for (int k = 0; k < m; ++k) t = t + cdx[k * N + i] * lambda[k];
            mLd[i] = t - mLd[i];
        }
        for (int k = 0; k < m; ++k) mLd[n + k] = cx[k];
        double nrm = 0.0;
        for (int i = 0; i < dim; ++i) nrm = nrm + mLd[i] * mLd[i];
        if (nrm < tol) break;
        for (size_t q = 0; q < D * D; ++q) Ldd[q] = 0.0;
        {
            double *H = (double *)malloc(sizeof(double) * N * N);
            fdd(H, x, n, ctx);
            cdd(cddx, x, m, n, ctx);
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) { /* Ldd(i,j)=fdd(i,j)-sum_k cddx(i,j,k) lambda_k */
                    double t = 0.0;
                    for (int k = 0; k < m; ++k) t = t + cddx[k * N * N + j * N + i] * lambda[k];
                    Ldd[j * D + i] = H[j * N + i] - t;
                }
            free(H);
        }
        for (int k = 0; k < m; ++k)
            for (int j = 0; j < n; ++j) Ldd[j * D + n + k] = -cdx[k * N + j]; /* Ldd(N+1:dim,1:N)=-transpose(cdx) */
        if (flo_dsysv(Ldd, mLd, dim) != 0) break;
        for (int i = 0; i < n; ++i) x[i] = x[i] + mLd[i];
        for (int k = 0; k < m; ++k) lambda[k] = lambda[k] + mLd[n + k];
        it = iter;
    }
    free(mLd);
    return it;
}

int flo_dposv_lower(double *A, double *b, int n)
{
#define A_(i, j) A[(size_t)(j) * n + (i)]
    for (int j = 0; j < n; ++j) {
        double ajj = A_(j, j);
        for (int k = 0; k < j; ++k) ajj = ajj - A_(j, k) * A_(j, k);
        if (!(ajj > 0.0)) return j + 1;
        ajj = sqrt(ajj);
        A_(j, j) = ajj;
        for (int i = j + 1; i < n; ++i) {
            double v = A_(i, j);
            for (int k = 0; k < j; ++k) v = v - A_(i, k) * A_(j, k);
            A_(i, j) = v / ajj;
        }
    }
    for (int j = 0; j < n; ++j) {
        const double q = b[j] / A_(j, j);
        b[j] = q;
        for (int i = j + 1; i < n; ++i) b[i] = b[i] - q * A_(i, j);
    }
    double *t = (double *)calloc((size_t)n, sizeof(double));
    for (int j = n - 1; j >= 0; --j) {
        for (int i = 0; i < n; ++i) t[i] = (i > j) ? A_(i, j) * b[i] : 0.0;
        double s;
        if (g_mode == FLO_SUM_TREE)
            s = flo_tree_sum(n, t);
        else {
            s = 0.0;
            for (int i = 0; i < n; ++i) s = s + t[i];
        }
        b[j] = (b[j] - s) / A_(j, j);
    }
    free(t);
#undef A_
    return 0;
}

/* central-difference Jacobian of the gradient: MKL's djacobi(fd_j,dim,dim,H,x,1d-8) (NO.f90:676, 981, 1067, 1258).
 * MKL is closed; its step rule was read off the points at which the real routine of the build image calls fcn
 * (tools/make_mkl_golden.py) and this restatement returns the real routine's bits on tests/golden/mkl_djacobi.npz
 * (tests/test_mkl_pins.py):  |x_j| > eps: f' at x_j (1 +- eps), h = eps x_j;  else f' at x_j +- eps, h = eps;
 * H(:,j) = (f'_plus - f'_minus) * (0.5 / h). */
void flo_central_hessian(flo_fd_t fd, double *H, double *x, int n, void *ctx, flo_stats *st)
{
    const double eps = 1e-8;
    double *gp = (double *)malloc(sizeof(double) * 2 * (size_t)n), *gm = gp + n;
    for (int j = 0; j < n; ++j) {
        double xj = x[j], h;
        if (fabs(xj) > eps) {
            h = eps * xj;
            x[j] = xj * (1.0 + eps);
            fd(gp, x, n, ctx);
            x[j] = xj * (1.0 - eps);
            fd(gm, x, n, ctx);
        } else {
            h = eps;
            x[j] = xj + eps;
            fd(gp, x, n, ctx);
            x[j] = xj - eps;
            fd(gm, x, n, ctx);
        }
        x[j] = xj;
        if (st) st->ng += 2;
        const double w = 0.5 / h;
        for (int i = 0; i < n; ++i) H[(size_t)j * n + i] = (gp[i] - gm[i]) * w;
    }
    free(gp);
}
static void central_hessian(flo_fd_t fd, double *H, double *x, int n, void *ctx, flo_stats *st)
{
    flo_central_hessian(fd, H, x, n, ctx, st);
}

/* p=-matmul(H,g), sequential in k (flang runtime matmul order) */
static void neg_matvec(int n, const double *H, const double *g, double *p)
{
    for (int i = 0; i < n; ++i) p[i] = 0.0;
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i) p[i] = p[i] + H[(size_t)k * n + i] * g[k];
    for (int i = 0; i < n; ++i) p[i] = -p[i];
}

/* rank-2 inverse-Hessian update.  form 0: as written, U=I-rho y s^T,
 * H=U^T (H U) + rho s s^T with two dense matmuls (NO.f90:958-962; first-step
 * variant with a*U instead of H U at NO.f90:711-715 when first!=0).
 * form 1: the algebraically equal O(n^2) expression the HIP kernel evaluates:
 *   q = H y, t = y.q, H_ij <- H_ij - rho q_i s_j - rho s_i q_j + (rho^2 t + rho) s_i s_j
 * (H symmetric; first step: H = a I so q = a y). */
static void bfgs_update(int n, double *H, double *U, const double *s, const double *y, double rho, int first,
                        double a, int form)
{
    size_t N = (size_t)n;
    if (form == 0) {
        double *T = (double *)malloc(sizeof(double) * N * N);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) U[j * N + i] = -(rho * (y[i] * s[j]));
        for (int i = 0; i < n; ++i) U[i * N + i] = U[i * N + i] + 1.0;
        if (first) {
            for (size_t k = 0; k < N * N; ++k) T[k] = a * U[k];
        } else {
            for (size_t k = 0; k < N * N; ++k) T[k] = 0.0;
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < n; ++k) {
                    double ukj = U[j * N + k];
                    for (int i = 0; i < n; ++i) T[j * N + i] = T[j * N + i] + H[k * N + i] * ukj;
                }
        }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                double v = 0.0;
                for (int k = 0; k < n; ++k) v = v + U[i * N + k] * T[j * N + k];
                H[j * N + i] = v + rho * (s[i] * s[j]);
            }
        free(T);
    } else {
        double *q = (double *)malloc(sizeof(double) * N);
        if (first) {
            for (int i = 0; i < n; ++i) q[i] = a * y[i];
        } else {
            for (int i = 0; i < n; ++i) { /* q_i = sum_k H(i,k) y_k, sequential in k (H column-major) */
                double v = 0.0;
                for (int k = 0; k < n; ++k) v = v + H[k * N + i] * y[k];
                q[i] = v;
            }
        }
        double t = flo_dot(n, y, q);
        double cs = rho * rho * t + rho;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                double h = first ? (i == j ? a : 0.0) : H[j * N + i];
                H[j * N + i] = h - rho * q[i] * s[j] - rho * s[i] * q[j] + cs * s[i] * s[j];
            }
        free(q);
    }
}

/* form 100+J: the rank-2 update DEFERRED (what the HIP kernels do for n > 1024, csrc/fl_device.hpp
 * direction_bfgs_deferred): H stays as it is for J iterations, the J updates
 *   H_{l+1} = H_l - rho_l q_l s_l^T - rho_l s_l q_l^T + cs_l s_l s_l^T,  q_l = H_l y_l
 * are kept as (s_l, q_l, rho_l, cs_l); H_l y and H_l g are formed as H y, H g plus the corrections, and after
 * the J-th the updates are folded into H element by element in the order they occurred -- bitwise the H that
 * applying them one at a time would have produced from the same q_l.  One read pass over H per iteration. */
typedef struct {
    int J, nd, ident;
    double aid;
    double *S, *Q; /* [J][n] */
    double rho[64], cs[64];
} bfgs_lazy;

static void lazy_fold(int n, double *H, bfgs_lazy *L)
{
    size_t N = (size_t)n;
    for (int b = 0; b < n; ++b)
        for (int a = 0; a < n; ++a) {
            double h = L->ident ? (a == b ? L->aid : 0.0) : H[b * N + a];
            for (int l = 0; l < L->nd; ++l) {
                const double *S = L->S + l * N, *Q = L->Q + l * N;
                h = h - (L->rho[l] * Q[a]) * S[b] - (L->rho[l] * S[a]) * Q[b] + (L->cs[l] * S[a]) * S[b];
            }
            H[b * N + a] = h;
        }
    L->ident = 0;
    L->nd = 0;
}

/* p = -H_new g, where H_new = H_cur updated with (s, y); first != 0: H_cur = a I (NO.f90:711-715) */
static void lazy_direction(int n, double *H, bfgs_lazy *L, const double *s, const double *y, double rho, int first,
                           double a, const double *g, double *p)
{
    size_t N = (size_t)n;
    double *q = (double *)malloc(sizeof(double) * 2 * N), *w = q + N;
    if (first) {
        L->ident = 1;
        L->aid = a;
        L->nd = 0;
    }
    for (int i = 0; i < n; ++i) {
        if (L->ident) {
            q[i] = L->aid * y[i];
            w[i] = L->aid * g[i];
        } else {
            double v = 0.0, u = 0.0;
            for (int k = 0; k < n; ++k) {
                v = v + H[k * N + i] * y[k];
                u = u + H[k * N + i] * g[k];
            }
            q[i] = v;
            w[i] = u;
        }
    }
    for (int l = 0; l < L->nd; ++l) {
        const double *S = L->S + l * N, *Q = L->Q + l * N;
        const double sy = flo_dot(n, S, y), qy = flo_dot(n, Q, y), sg = flo_dot(n, S, g), qg = flo_dot(n, Q, g);
        for (int i = 0; i < n; ++i) {
            const double rq = L->rho[l] * Q[i], rs = L->rho[l] * S[i], cc = L->cs[l] * S[i];
            q[i] = q[i] - rq * sy - rs * qy + cc * sy;
            w[i] = w[i] - rq * sg - rs * qg + cc * sg;
        }
    }
    const double t = flo_dot(n, y, q), cs = rho * rho * t + rho;
    const double sg = flo_dot(n, s, g), qg = flo_dot(n, q, g);
    for (int i = 0; i < n; ++i) p[i] = -(w[i] - (rho * q[i]) * sg - (rho * s[i]) * qg + (cs * s[i]) * sg);
    memcpy(L->S + L->nd * N, s, sizeof(double) * N);
    memcpy(L->Q + L->nd * N, q, sizeof(double) * N);
    L->rho[L->nd] = rho;
    L->cs[L->nd] = cs;
    L->nd++;
    if (L->nd == L->J) lazy_fold(n, H, L);
    free(q);
}

/* one update of a column-major n x n inverse Hessian (rho computed here), for the tests of the
 * stand-alone GPU update kernels: form 0 = the reference's two matmuls, form 1 = rank-2 */
void flo_bfgs_update(int n, double *H, const double *s, const double *y, int form)
{
    double *U = (double *)malloc(sizeof(double) * (size_t)n * n);
    const double rho = 1.0 / flo_dot(n, y, s);
    bfgs_update(n, H, U, s, y, rho, 0, 0.0, form);
    free(U);
}

/* BFGS, NO.f90:632-1022 */
void flo_bfgs(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, double *x, int n, const flo_opts *o,
              int update_form, void *ctx, flo_stats *st)
{
    const int freq = o->exact_step;
    double tol = o->precision * o->precision, minstep = o->minstep * o->minstep, c1, c2;
    double a, fnew, phidnew, rho;
    size_t N = (size_t)n;
    double *p = (double *)malloc(sizeof(double) * (4 * N + 2 * N * N));
    double *g = p + N, *s = g + N, *y = s + N, *U = y + N, *H = U + N * N;
    int info = 1;
    bfgs_lazy L;
    L.J = update_form >= 100 ? update_form - 100 : 0;
    L.nd = 0;
    L.ident = 0;
    L.aid = 0.0;
    L.S = L.J > 0 ? (double *)malloc(sizeof(double) * 2 * (size_t)L.J * N) : NULL;
    L.Q = L.J > 0 ? L.S + (size_t)L.J * N : NULL;
    clamp_c(o, &c1, &c2);
    st_zero(st);
    initial_eval(f, fd, f_fd, &fnew, g, x, n, ctx, st);
    st->f = fnew;
    st->gg = flo_dot(n, g, g);
    if (freq > 0) { /* NO.f90:674-682 */
        if (fdd)
            fdd(H, x, n, ctx);
        else
            central_hessian(fd, H, x, n, ctx, st);
        info = flo_dpotri_lower(H, n);
        if (info == 0) {
            flo_syL2U(H, n);
            neg_matvec(n, H, g, p);
            phidnew = flo_dot(n, g, p);
            a = 1.0;
        }
    }
    if (freq <= 0 || info != 0) { /* NO.f90:683-716 */
        for (int i = 0; i < n; ++i) p[i] = -g[i];
        phidnew = -flo_dot(n, g, g);
        if (-phidnew < tol) goto out;
        a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(-phidnew);
        memcpy(s, x, sizeof(double) * N);
        memcpy(y, g, sizeof(double) * N);
        line_search(o, c1, c2, 0, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) goto out;
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            goto out;
        }
        for (int i = 0; i < n; ++i) {
            s[i] = x[i] - s[i];
            y[i] = g[i] - y[i];
        }
        rho = 1.0 / flo_dot(n, y, s);
        if (L.J > 0) {
            lazy_direction(n, H, &L, s, y, rho, 1, a, g, p);
        } else {
            bfgs_update(n, H, U, s, y, rho, 1, a, update_form);
            neg_matvec(n, H, g, p);
        }
        phidnew = flo_dot(n, g, p);
        a = 1.0;
    }
    st->status = FLO_MAXIT;
    for (int it = 1; it <= o->maxit; ++it) {
        memcpy(s, x, sizeof(double) * N);
        memcpy(y, g, sizeof(double) * N);
        line_search(o, c1, c2, f_fd != NULL, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        /* After / After_NumericalHessian / After_NoHessian, NO.f90:935-1015 */
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) {
            st->status = FLO_CONVERGED;
            break;
        }
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            break;
        }
        int i = 1;
        if (freq > 0) {
            i = it % freq;
            if (i == 0) { /* every freq steps compute exact Hessian */
                if (fdd)
                    fdd(U, x, n, ctx);
                else
                    central_hessian(fd, U, x, n, ctx, st);
                i = flo_dpotri_lower(U, n);
                if (i == 0) { /* sycp(H,U); syL2U(H) */
                    for (int c = 0; c < n; ++c)
                        for (int r = c; r < n; ++r) H[c * N + r] = U[c * N + r];
                    flo_syL2U(H, n);
                    L.nd = 0; /* pending updates belong to the matrix that has just been replaced */
                    L.ident = 0;
                    neg_matvec(n, H, g, p);
                    phidnew = flo_dot(n, g, p);
                    a = 1.0;
                }
            }
        }
        if (i != 0) {
            for (int k = 0; k < n; ++k) {
                s[k] = x[k] - s[k];
                y[k] = g[k] - y[k];
            }
            rho = 1.0 / flo_dot(n, y, s);
            if (L.J > 0) {
                lazy_direction(n, H, &L, s, y, rho, 0, 0.0, g, p);
            } else {
                bfgs_update(n, H, U, s, y, rho, 0, 0.0, update_form);
                neg_matvec(n, H, g, p);
            }
            phidnew = flo_dot(n, g, p);
            a = 1.0;
        }
    }
out:
    free(L.S);
    free(p);
}

/* NewtonRaphson with analytic Hessian (fdd present), NO.f90:1026-1271; After() 1216-1238 */
void flo_newton(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, double *x, int n, const flo_opts *o, void *ctx,
                flo_stats *st)
{
    double tol = o->precision * o->precision, minstep = o->minstep * o->minstep, c1, c2;
    double a = 0.0, fnew, phidnew, phidold;
    size_t N = (size_t)n;
    double *p = (double *)malloc(sizeof(double) * (2 * N + N * N)), *g = p + N, *H = g + N;
    int info;
    clamp_c(o, &c1, &c2);
    st_zero(st);
    initial_eval(f, fd, f_fd, &fnew, g, x, n, ctx, st);
    st->f = fnew;
    st->gg = flo_dot(n, g, g);
    /* info=fdd(Hessian,x,dim), or -- fdd absent -- djacobi(fd_j,dim,dim,Hessian,x,1d-8) (NO.f90:1065-1067) */
    if (fdd) fdd(H, x, n, ctx);
    else central_hessian(fd, H, x, n, ctx, st);
    for (int i = 0; i < n; ++i) p[i] = -g[i];
    info = flo_dposv_lower(H, p, n);
    if (info == 0) {
        phidnew = flo_dot(n, g, p);
        a = 1.0;
    } else { /* Hessian is not positive definite, use steepest descent direction */
        for (int i = 0; i < n; ++i) p[i] = -g[i];
        phidnew = -flo_dot(n, g, g);
        if (-phidnew < tol) goto out;
        a = (fnew == 0.0) ? 1.0 : fabs(fnew) / sqrt(-phidnew);
    }
    st->status = FLO_MAXIT;
    for (int it = 1; it <= o->maxit; ++it) {
        phidold = phidnew;
        line_search(o, c1, c2, f_fd != NULL, f, fd, f_fd, x, &a, p, &fnew, phidnew, g, n, ctx, st);
        phidnew = flo_dot(n, g, g);
        st->f = fnew;
        st->gg = phidnew;
        if (phidnew < tol) {
            st->status = FLO_CONVERGED;
            break;
        }
        if (flo_dot(n, p, p) * a * a < minstep) {
            st->status = FLO_STEP_CONVERGED;
            break;
        }
        for (int i = 0; i < n; ++i) p[i] = -g[i];
        if (fdd) fdd(H, x, n, ctx);
        else central_hessian(fd, H, x, n, ctx, st);
        info = flo_dposv_lower(H, p, n);
        if (info == 0) {
            phidnew = flo_dot(n, g, p);
            a = 1.0;
        } else {
            for (int i = 0; i < n; ++i) p[i] = -g[i];
            phidnew = -phidnew;
            a = a * phidold / phidnew;
        }
    }
out:
    free(p);
}

/* ------------------------------------------------- augmented Lagrangian */
typedef struct {
    flo_f_t f;
    flo_fd_t fd;
    flo_ffd_t f_fd;
    flo_c_t c;
    flo_cd_t cd;
    flo_fdd_t fdd; /* with cdd: the Hessian of L is available (Ldd, NO.f90:2229) */
    int (*cdd)(double *, const double *, int, int, void *);
    void *ctx;
    int m;
    double *lambda, miu, *cx, *cdx, *v, *cddx, *tmp;
    int nf, ng, nc;
} al_t;

static void al_lx(al_t *A, double *Lx) /* Lx - lambda.cx + miu/2*cx.cx, NO.f90:2198 */
{
    *Lx = *Lx - seq_dot(A->m, A->lambda, A->cx) + A->miu / 2.0 * seq_dot(A->m, A->cx, A->cx);
}
static void al_ldx(al_t *A, double *Ldx, int n) /* Ldx+matmul(cdx,miu*cx-lambda), NO.f90:2205 */
{
    for (int j = 0; j < A->m; ++j) A->v[j] = A->miu * A->cx[j] - A->lambda[j];
    for (int i = 0; i < n; ++i) {
        double t = 0.0;
        for (int j = 0; j < A->m; ++j) t = t + A->cdx[(size_t)j * n + i] * A->v[j];
        Ldx[i] = Ldx[i] + t;
    }
}
static void al_L(double *Lx, const double *x, int n, void *vp) /* L, NO.f90:2193-2199 */
{
    al_t *A = (al_t *)vp;
    A->f(Lx, x, n, A->ctx);
    A->nf++;
    A->c(A->cx, x, A->m, n, A->ctx);
    A->nc++;
    al_lx(A, Lx);
}
static void al_Ld(double *Ldx, const double *x, int n, void *vp) /* Ld, NO.f90:2200-2206 */
{
    al_t *A = (al_t *)vp;
    A->fd(Ldx, x, n, A->ctx);
    A->ng++;
    A->c(A->cx, x, A->m, n, A->ctx);
    A->nc++;
    A->cd(A->cdx, x, A->m, n, A->ctx);
    al_ldx(A, Ldx, n);
}
static int al_L_Ld(double *Lx, double *Ldx, const double *x, int n, void *vp) /* NO.f90:2207-2228 */
{
    al_t *A = (al_t *)vp;
    if (A->f_fd) { /* L_Ld_fdwithf */
        A->f_fd(Lx, Ldx, x, n, A->ctx);
        A->nf++;
        A->ng++;
        A->c(A->cx, x, A->m, n, A->ctx);
        A->nc++;
        al_lx(A, Lx);
        A->cd(A->cdx, x, A->m, n, A->ctx);
        al_ldx(A, Ldx, n);
    } else { /* L_Ld */
        A->f(Lx, x, n, A->ctx);
        A->nf++;
        A->c(A->cx, x, A->m, n, A->ctx);
        A->nc++;
        al_lx(A, Lx);
        A->fd(Ldx, x, n, A->ctx);
        A->ng++;
        A->cd(A->cdx, x, A->m, n, A->ctx);
        al_ldx(A, Ldx, n);
    }
    return 0;
}

/* Ldd, NO.f90:2229-2241:  i=fdd(Lddx,x,N); i=cdd(cddx,x,M,N); call c(cx,...); call cd(cdx,...); cx=miu*cx-lambda;
 * Lddxtemp(:,i)=matmul(cddx(i,:,:),cx); Lddx=Lddx+Lddxtemp+matmul(cdx,transpose(cdx))   (no miu on the last term: as
 * written).  H column-major n x n; cddx(N,N,M): element (i,k,j) at i + k*n + j*n*n; cdx(N,M): (k,j) at k + j*n. */
static int al_Ldd(double *H, const double *x, int n, void *vp)
{
    al_t *A = (al_t *)vp;
    const int m = A->m;
    A->fdd(H, x, n, A->ctx);
    A->cdd(A->cddx, x, m, n, A->ctx);
    A->c(A->cx, x, m, n, A->ctx);
    A->nc++;
    A->cd(A->cdx, x, m, n, A->ctx);
    for (int j = 0; j < m; ++j) A->v[j] = A->miu * A->cx[j] - A->lambda[j];
    for (int i = 0; i < n; ++i)     /* column i */
        for (int k = 0; k < n; ++k) { /* row k */
            double t = 0.0, p = 0.0;
            for (int j = 0; j < m; ++j) t = t + A->cddx[(size_t)i + (size_t)k * n + (size_t)j * n * n] * A->v[j];
            for (int j = 0; j < m; ++j) p = p + A->cdx[(size_t)k + (size_t)j * n] * A->cdx[(size_t)i + (size_t)j * n];
            H[(size_t)k + (size_t)i * n] = (H[(size_t)k + (size_t)i * n] + t) + p;
        }
    return 0;
}

/* AugmentedLagrangian, NO.f90:2005-2241 (LBFGS case 2150-2167, CG 2168-2185,
 * BFGS without fdd/cdd 2131-2148).  st->nf/ng count USER f/fd calls. */
/* update form of the inner BFGS (0 = as written; 1 / 100+J = what the kernels evaluate): set once before a run */
static int al_bfgs_form = 0;
void flo_set_auglag_bfgs_form(int form) { al_bfgs_form = form; }

void flo_augmented_lagrangian(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_c_t c, flo_cd_t cd, double *x, int n,
                              int m, int solver, double *lambda, double miu0, const flo_opts *o, void *ctx,
                              flo_stats *st, int *outer_iters, double *cnorm2)
{
    flo_augmented_lagrangian_h(f, fd, f_fd, NULL, c, cd, NULL, x, n, m, solver, lambda, miu0, o, ctx, st, outer_iters, cnorm2);
}

/* the same with fdd and cdd (both or neither): solver 3 = NewtonRaphson (NO.f90:2074-2130), solver 0 = BFGS takes the
 * exact inverse Hessian of L every o->exact_step iterations when they are present (2131-2148) */
void flo_augmented_lagrangian_h(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, flo_c_t c, flo_cd_t cd,
                                int (*cdd)(double *, const double *, int, int, void *), double *x, int n, int m, int solver,
                                double *lambda, double miu0, const flo_opts *o, void *ctx, flo_stats *st,
                                int *outer_iters, double *cnorm2)
{
    al_t A;
    const int hess = fdd && cdd;
    A.fdd = fdd;
    A.cdd = cdd;
    A.cddx = hess ? (double *)malloc(sizeof(double) * (size_t)n * n * m) : NULL;
    A.tmp = NULL;
    flo_stats in;
    double tolsq = o->precision * o->precision, cc = 0.0;
    A.f = f;
    A.fd = fd;
    A.f_fd = f_fd;
    A.c = c;
    A.cd = cd;
    A.ctx = ctx;
    A.m = m;
    A.lambda = lambda;
    A.miu = dmax(1.0, miu0);
    A.cx = (double *)malloc(sizeof(double) * (2 * (size_t)m + (size_t)n * m));
    A.v = A.cx + m;
    A.cdx = A.v + m;
    A.nf = A.ng = A.nc = 0;
    st_zero(st);
    st->status = FLO_MAXIT;
    int it;
    for (it = 1; it <= o->maxit; ++it) {
        if (solver == 1)
            flo_lbfgs(al_L, al_Ld, al_L_Ld, x, n, o, &A, &in);
        else if (solver == 2)
            flo_conjugate_gradient(al_L, al_Ld, al_L_Ld, x, n, o, &A, &in);
        else if (solver == 3)
            flo_newton(al_L, al_Ld, al_L_Ld, hess ? al_Ldd : NULL, x, n, o, &A, &in);
        else
            flo_bfgs(al_L, al_Ld, al_L_Ld, hess ? al_Ldd : NULL, x, n, o, al_bfgs_form, &A, &in);
        st->iters += in.iters;
        st->f = in.f;
        st->gg = in.gg;
        c(A.cx, x, m, n, ctx);
        A.nc++;
        cc = seq_dot(m, A.cx, A.cx);
        if (cc < tolsq) {
            st->status = FLO_CONVERGED;
            break;
        }
        for (int j = 0; j < m; ++j) lambda[j] = lambda[j] - A.miu * A.cx[j];
        A.miu = A.miu * o->increment;
    }
    st->nf = A.nf;
    st->ng = A.ng;
    if (outer_iters) *outer_iters = it > o->maxit ? o->maxit : it;
    if (cnorm2) *cnorm2 = cc;
    free(A.cx);
    free(A.cddx);
}
