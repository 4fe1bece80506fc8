/*
 * fl_oracle.h -- CPU restatement (plain C) of the line-search optimisers of the
 * reference's source/NonlinearOptimization.f90 ("NO.f90").
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported CPU baseline.  The product (libFL.so,
 * fortran-library_amd/csrc) never links, loads or calls it.
 *
 * Parity status: the reference ships no golden vectors for this path
 * (test/test.f90 seeds from the wall clock) and cannot be built in this image
 * without a hand-written stand-in for Intel's closed mkl_rci.f90 (NO.f90:15),
 * so no reference build is made.  The restatement is pinned against the
 * reference outputs recorded in BASELINE.md section 2 (final objective values
 * and f / grad evaluation counts measured on the reference during the survey);
 * see tests/test_oracle_pins.py.  The numerical-Hessian branches call MKL's djacobi: its restatement here
 * (flo_central_hessian) is pinned to the real routine's outputs (tests/golden/mkl_djacobi.npz).
 *
 * Two summation modes:
 *   FLO_SUM_SEQ  : every dot_product / objective sum runs left to right like
 *                  the reference compiled without fast-math (NO.f90 uses the
 *                  dot_product intrinsic everywhere, e.g. NO.f90:442, 591).
 *   FLO_SUM_TREE : the same algorithm with every sum taken in the fixed
 *                  reduction order of the HIP kernels (thread-strided partials,
 *                  the 64-lane tree of csrc/fl_reduce.hpp: halves, quarter rows,
 *                  mirror steps inside a row; waves left to right) so that the GPU
 *                  result can be compared BIT FOR BIT.
 */
#ifndef FL_ORACLE_H
#define FL_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* callbacks: the reference's (NO.f90:33-38, 1928-1937) plus a context pointer */
typedef void (*flo_f_t)(double *fx, const double *x, int n, void *ctx);
typedef void (*flo_fd_t)(double *g, const double *x, int n, void *ctx);
typedef int (*flo_ffd_t)(double *fx, double *g, const double *x, int n, void *ctx);
typedef int (*flo_fdd_t)(double *H, const double *x, int n, void *ctx);
typedef void (*flo_c_t)(double *cx, const double *x, int m, int n, void *ctx);
typedef void (*flo_cd_t)(double *cdx, const double *x, int m, int n, void *ctx);

enum { FLO_SUM_SEQ = 0, FLO_SUM_TREE = 1 };
/* thread-local: summation mode and GPU reduction geometry (threads per problem,
 * elements per thread) used by FLO_SUM_TREE */
void flo_set_sum_mode(int mode, int threads, int ept);
void flo_set_sum_groups(int groups); /* FLO_SUM_TREE: workgroups sharing one problem (cooperative vectors-in-HBM kernels); process-wide */
double flo_dot(int n, const double *a, const double *b);
double flo_tree_sum(int n, const double *term);

enum { FLO_CONVERGED = 0, FLO_STEP_CONVERGED = 1, FLO_MAXIT = 2 };

typedef struct {
    /* optional arguments of the reference, all explicit (defaults: flo_defaults) */
    int strong;       /* Strong         (default 1)     */
    int maxit;        /* MaxIteration   (default 1000)  */
    double precision; /* Precision      (default 1e-15) */
    double minstep;   /* MinStepLength  (default 1e-15) */
    double c1, c2;    /* WolfeConst1/2  (1e-4, 0.9; CG 0.45) */
    double increment; /* Increment      (default 1.05)  */
    int memory;       /* L-BFGS Memory  (default 10)    */
    int exact_step;   /* BFGS ExactStep (default 20)    */
    int method;       /* CG: 0 = DY, 1 = PR */
    int clamp;        /* apply the fail-safe clamps of NO.f90:83-86 (1) or not (_basic, 0) */
} flo_opts;

typedef struct {
    int status;  /* FLO_* */
    int iters;   /* line searches performed */
    int nf, ng;  /* f / grad evaluations (f_fd counts as one of each) */
    double f;    /* objective at exit */
    double gg;   /* g.g at exit */
} flo_stats;

void flo_defaults(flo_opts *o);

/* line searchers: NO.f90:1286 (Wolfe), 1373 (Wolfe_fdwithf), 1462 (StrongWolfe),
 * 1582 (StrongWolfe_fdwithf).  p is the direction; on exit x = x0 + a p. */
void flo_wolfe(double c1, double c2, flo_f_t f, flo_fd_t fd, double *x, double *a, const double *p,
               double *fx, double phid0, double *fdx, int n, double increment, void *ctx, flo_stats *st);
void flo_strong_wolfe(double c1, double c2, flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, double *a,
                      const double *p, double *fx, double phid0, double *fdx, int n, double increment,
                      void *ctx, flo_stats *st);

/* solvers: f_fd may be NULL ("absent") */
void flo_steepest_descent(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o,
                          void *ctx, flo_stats *st); /* NO.f90:55  */
void flo_conjugate_gradient(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o,
                            void *ctx, flo_stats *st); /* NO.f90:193 */
void flo_lbfgs(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, double *x, int n, const flo_opts *o, void *ctx,
               flo_stats *st); /* NO.f90:398 */
/* BFGS (NO.f90:632).  update_form 0 = as written (two dense matmuls, NO.f90:961),
 * 1 = algebraically equal rank-2 form (what the HIP kernel computes). fdd may be NULL;
 * exact_step > 0 without fdd: central differences with djacobi's step rule (flo_central_hessian). */
void flo_bfgs(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, double *x, int n, const flo_opts *o,
              int update_form, void *ctx, flo_stats *st);
/* NewtonRaphson with analytic Hessian (NO.f90:1026): Cholesky solve, steepest-descent fallback */
void flo_newton(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, double *x, int n, const flo_opts *o, void *ctx,
                flo_stats *st);
void flo_central_hessian(flo_fd_t fd, double *H, double *x, int n, void *ctx, flo_stats *st); /* MKL djacobi(fd_j,n,n,H,x,1d-8); st may be NULL */
int flo_dposv_lower(double *A, double *b, int n); /* My_dposv LA.f90:719 : returns info, b untouched if it fails */
int flo_dsysv(double *A, double *b, int n);       /* My_dsysv LA.f90:695 : symmetric indefinite (lower), elimination with partial pivoting */
typedef int (*flo_cdd_t)(double *cddx, const double *x, int m, int n, void *ctx); /* c''(x): N x N x M */
int flo_lagrangian_multiplier(flo_fd_t fd, flo_fdd_t fdd, flo_c_t c, flo_cd_t cd, flo_cdd_t cdd, double *x,
                              double *lambda, int n, int m, int maxit, double precision, void *ctx); /* NO.f90:1950 */
/* AugmentedLagrangian (NO.f90:2005): solver 0 = BFGS, 1 = LBFGS, 2 = ConjugateGradient.
 * lambda[m] in/out (reference: lambda0 copy), miu0 as given. outer_iters returns the
 * number of outer iterations. */
void flo_augmented_lagrangian_h(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_fdd_t fdd, flo_c_t c, flo_cd_t cd,
                                int (*cdd)(double *, const double *, int, int, void *), double *x, int n, int m, int solver,
                                double *lambda, double miu0, const flo_opts *o, void *ctx, flo_stats *st,
                                int *outer_iters, double *cnorm2); /* with fdd, cdd: Ldd (solver 3 = NewtonRaphson) */
void flo_set_auglag_bfgs_form(int form); /* BFGS update form inside flo_augmented_lagrangian (default 0 = as written) */
void flo_augmented_lagrangian(flo_f_t f, flo_fd_t fd, flo_ffd_t f_fd, flo_c_t c, flo_cd_t cd, double *x, int n,
                              int m, int solver, double *lambda, double miu0, const flo_opts *o, void *ctx,
                              flo_stats *st, int *outer_iters, double *cnorm2);

/* one BFGS update of a column-major inverse Hessian: form 0 = U^T (H U) + rho s s^T as two
 * matmuls (NO.f90:958-962), form 1 = the rank-2 expression */
void flo_bfgs_update(int n, double *H, const double *s, const double *y, int form);

/* dense helpers restated from LinearAlgebra.f90 (column-major) */
int flo_dpotri_lower(double *A, int n); /* My_dpotri LA.f90:798 : dpotrf+dpotri 'L'; returns info */
void flo_syL2U(double *A, int n);       /* dsyL2U LA.f90:260 */

/* ---- built-in synthetic problems (fl_oracle_problems.c) ---- */
enum { FLO_QUARTIC = 0, FLO_ROSENBROCK = 1, FLO_DIAGQUAD = 2 };
typedef struct {
    int kind;
    const double *d, *b; /* DIAGQUAD data */
    /* block-sphere constraints c_j = sum_{i in block j} x_i^2 - 1 (m blocks of n/m) */
} flo_problem;
void flo_prob_f(double *fx, const double *x, int n, void *ctx);
void flo_prob_fd(double *g, const double *x, int n, void *ctx);
int flo_prob_ffd(double *fx, double *g, const double *x, int n, void *ctx);
int flo_prob_fdd(double *H, const double *x, int n, void *ctx);
void flo_prob_c(double *cx, const double *x, int m, int n, void *ctx);
void flo_prob_cd(double *cdx, const double *x, int m, int n, void *ctx);
int flo_prob_cdd(double *cddx, const double *x, int m, int n, void *ctx);

enum { FLO_SD = 0, FLO_CG = 1, FLO_LBFGS = 2, FLO_BFGS = 3 };
/* batched driver, OpenMP over problems (one problem per thread, like the reference
 * which is single-threaded per problem).  x[B][n] in/out, d/b [B][n] or NULL,
 * outputs [B].  sum_mode/threads/ept select the summation order.  use_ffd: pass
 * f_fd to the solver (the reference's "f_fd present").  Returns threads used. */
int flo_solve_batch(int solver, int kind, int B, int n, double *x, const double *d, const double *b,
                    const flo_opts *o, int use_ffd, int bfgs_form, int sum_mode, int threads, int ept,
                    int nthreads, double *fout, int *iters, int *status, int *nf, int *ng, double *gg);
int flo_prob_eval_batch(int kind, int B, int n, int m, const double *x, const double *d, const double *b, int sum_mode,
                        int threads, int ept, int nthreads, double *f, double *g, double *c, double *cd);
int flo_auglag_batch(int solver, int kind, int B, int n, int m, double *x, const double *d, const double *b,
                     double *lambda, double miu0, const flo_opts *o, int use_ffd, int sum_mode, int threads,
                     int ept, int nthreads, double *fout, int *iters, int *outer, int *nf, int *ng,
                     double *cnorm2);

#ifdef __cplusplus
}
#endif
#endif
