/*
 * fl_nlopt.h -- C ABI of libFL.so (MI355X / gfx950): batched line-search optimisers.
 *
 * This is the drop-in boundary for the NonlinearOptimization hot path of
 * YifanShenSZ/Fortran-Library.  Every entry point is extern "C", takes plain
 * pointers and sizes (no torch / C++ types) and is what a Fortran bind(C)
 * interface, a C++ header, or Python ctypes binds (INTEGRATION.md shows each).
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   fl_lbfgs_batched               <- subroutine LBFGS              source/NonlinearOptimization.f90:398-625
 *   fl_conjugate_gradient_batched  <- subroutine ConjugateGradient   NonlinearOptimization.f90:193-394
 *                                     (+ ConjugateGradient_basic     NonlinearOptimization.f90:2249-2346,
 *                                      cpp/NonlinearOptimization.hpp:294-324)
 *   fl_steepest_descent_batched    <- subroutine SteepestDescent     NonlinearOptimization.f90:55-188
 *                                     (cpp/NonlinearOptimization.hpp:279-292)
 *   line search inside all of them <- Wolfe / StrongWolfe (+_fdwithf) NonlinearOptimization.f90:1286-1698
 *
 * Differences from the reference, by design: a BATCH of independent problems
 * per call (problem-major [batch][n] fp64 arrays in device memory), objectives
 * selected from built-in device functors instead of host callbacks, and
 * per-problem outputs (objective, iteration / evaluation counts, status) that the
 * reference never returned.  Option names, defaults, clamps and termination
 * tests are the reference's (NO.f90:41-51, 73-86, 419-434).
 *
 * All pointers named "dev" are device (HBM) pointers.  All calls are
 * asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 * stream); outputs are valid after the stream is synchronised.  Return value:
 * FL_OK or a negative FL_ERR_* code; nothing is printed and nothing runs on the
 * CPU as a fallback -- without a usable HIP device every call fails with
 * FL_ERR_NO_DEVICE; a launch the runtime refuses is FL_ERR_LAUNCH.
 */
#ifndef FL_NLOPT_H
#define FL_NLOPT_H
#ifndef __HIPCC_RTC__ /* (hiprtc brings its own) */
#include <stddef.h>
#include <stdint.h>
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define FL_OK 0
#define FL_ERR_INVALID_ARGUMENT (-1) /* NULL pointer, batch/n <= 0, unknown enum */
#define FL_ERR_UNSUPPORTED_SIZE (-2) /* Memory > FL_MAX_MEMORY; n > 4096 for NewtonRaphson, BFGS with exact_step > 0 and */
                                     /* the augmented Lagrangian; n > 16384 for BFGS; n > 2^27 for SD / CG / L-BFGS      */
#define FL_ERR_WORKSPACE (-3)        /* workspace missing or too small */
#define FL_ERR_NO_DEVICE (-4)        /* no usable HIP device (hipGetDeviceCount), decided before anything is launched */
#define FL_ERR_LAUNCH (-5)           /* a kernel launch or a stream-ordered allocation was refused by the runtime */

/* built-in objectives (SURVEY.md section 8d synthetic inputs) */
#define FL_OBJ_QUARTIC 0    /* f = sum x_i^4                      (test/test.f90:630-663)        */
#define FL_OBJ_ROSENBROCK 1 /* f = sum_{i<n} 100 (x_{i+1}-x_i^2)^2 + (1-x_i)^2  (chained)       */
#define FL_OBJ_DIAGQUAD 2   /* f = 1/2 sum d_i x_i^2 - sum b_i x_i ; data d,b [batch][n]         */

/* per-problem exit status */
#define FL_STATUS_CONVERGED 0      /* g.g < Precision^2                   (NO.f90:612)          */
#define FL_STATUS_STEP_CONVERGED 1 /* p.p a^2 < MinStepLength^2 "step length has converged" (NO.f90:615) */
#define FL_STATUS_MAXIT 2          /* MaxIteration exceeded              (NO.f90:580)           */
#define FL_STATUS_NOT_FINITE 4     /* fused kernels: the objective returned NaN -- the problem stops where it is (the       */
                                   /* reference's line searchers would never return: their loops end on comparisons,       */
                                   /* NO.f90:1557-1579; by reverse communication the loop is the caller's)                 */
#define FL_STATUS_STALLED 5        /* fused kernels: the line search's zoom did not narrow its bracket in 65 536 consecutive  */
                                   /* trials -- the reference's zoom has no iteration limit and never returns on such a      */
                                   /* problem (NO.f90:1557-1579); the problem stops at the last trial point                  */
#define FL_STATUS_NOT_SOLVED (-1)  /* fl_multi_solve: the shard holding this problem failed (allocation, launch): its rows of x */
                                   /* and of the outputs are untouched; cooperative form (fl_cooperative_groups_for): the       */
                                   /* problem's workgroups were not resident together, x and the outputs mean nothing           */

#define FL_CG_DY 0 /* Dai-Yuan        NO.f90:352-372 */
#define FL_CG_PR 1 /* Polak-Ribiere+  NO.f90:373-393 */

#define FL_MAX_MEMORY 64 /* largest L-BFGS Memory held by the kernels */

/* The reference's optional arguments (NO.f90:41-51).  fl_default_options fills the
 * reference defaults; solver-specific default: ConjugateGradient uses
 * wolfe_c2 = 0.45 (NO.f90:229) -- fl_default_options(opt, FL_SOLVER_CG) does that. */
typedef struct fl_options {
    int32_t strong;          /* Strong         : 1 = strong Wolfe (default), 0 = Wolfe            */
    int32_t max_iteration;   /* MaxIteration   : default 1000                                     */
    double precision;        /* Precision      : default 1e-15 (squared internally, NO.f90:427)   */
    double min_step_length;  /* MinStepLength  : default 1e-15 (squared internally)               */
    double wolfe_c1;         /* WolfeConst1    : default 1e-4                                     */
    double wolfe_c2;         /* WolfeConst2    : default 0.9 (CG 0.45)                            */
    double increment;        /* Increment      : default 1.05 (line-search growth factor)         */
    int32_t memory;          /* L-BFGS Memory  : default 10, clamped to >= 1 (NO.f90:419)         */
    int32_t cg_method;       /* FL_CG_DY (default) | FL_CG_PR                                     */
    int32_t fused_f_fd;      /* 1 = behave as if the caller passed f_fd (main loops use the        */
                             /*     *_fdwithf line searchers, NO.f90:512-527); default 0          */
    int32_t clamp;           /* 1 = apply the fail-safe clamps on c1,c2 (NO.f90:83-86; default),   */
                             /* 0 = take them verbatim (ConjugateGradient_basic, NO.f90:2249)     */
    int32_t exact_step;      /* BFGS ExactStep: every how many steps the exact inverse Hessian is  */
                             /* recomputed (NO.f90:630-631, default 20; <= 0 never).  The batched  */
                             /* solvers use the built-in objective's analytic Hessian (the fdd     */
                             /* branch, NO.f90:675, 951); reverse communication runs <= 0 only     */
} fl_options;

#define FL_SOLVER_SD 0
#define FL_SOLVER_CG 1
#define FL_SOLVER_LBFGS 2
#define FL_SOLVER_BFGS 3
#define FL_SOLVER_NEWTON 4 /* NewtonRaphson with analytic Hessian, NO.f90:1026-1271 */

int fl_version(void);
void fl_default_options(fl_options *opt, int solver);

/* Reduction geometry the kernels use for dimension n: `threads` per problem (one
 * workgroup), `ept` elements per thread.  Sums are taken per thread over its
 * elements, then the fixed 64-lane tree of csrc/fl_reduce.hpp (lanes l and l+32,
 * then l and l+16, then mirror steps inside a row of 16), then waves left to right -- the order
 * tests replay on the CPU to compare bit for bit.  n <= 4096: the problem's vectors live
 * in registers (threads <= 512, ept <= 8).  n > 4096 (SD / CG / L-BFGS; BFGS with quasi-Newton
 * updates only up to 16384): the same machine with its vectors in HBM, threads = 1024,
 * ept = 2*ceil(ceil(n/2)/1024) (csrc/fl_big.hpp).
 * FL_ERR_UNSUPPORTED_SIZE beyond 2^27. */
int fl_reduction_geometry(int n, int *threads, int *ept);
/* The geometry of the FUSED kernel of one solver (fl_*_batched without constraints): as above except that
 * FL_SOLVER_SD / FL_SOLVER_CG run 512 < n <= 4096 with 16 elements per thread and half the waves (1 / 2 / 4 x 16: no
 * history to keep, the state fits) and FL_SOLVER_NEWTON 256 < n <= 512 with two waves x 4 (its Cholesky wants the threads).  threads*ept -- the padded length of every workspace row -- is the same for all solvers of
 * an n; the reverse-communication kernels (fl_rci_*) and the dense routines use fl_reduction_geometry's. */
int fl_reduction_geometry_for(int solver, int n, int *threads, int *ept);
/* The fused BFGS kernels (fl_bfgs_batched, BFGS inside fl_augmented_lagrangian_batched) apply their rank-2 updates DEFERRED for
 * n > 128: H is left alone for 8 iterations, the product H g is corrected with the pending updates' vectors, and every 8th
 * iteration they are folded into H in the order they occurred -- algebraically the reference's update (NO.f90:958-962), a third of
 * its HBM traffic.  Returns how many updates stay pending for dimension n (0 = each is applied at once: n <= 128, and by reverse
 * communication up to n = 4096 -- there H's bytes are a small part of a step and the deferred form measured slower).  A bit-exact
 * replay needs it (oracle update_form 100 + this). */
int fl_bfgs_deferred_updates(int n);
/* FEW problems of very large n (the reference's callers typically solve ONE problem of any dim): beyond n = 14336 and below one
 * problem per two compute units, fl_steepest_descent_batched / fl_conjugate_gradient_batched / fl_lbfgs_batched (objectives
 * FL_OBJ_QUARTIC, FL_OBJ_DIAGQUAD) and fl_user_solve (a streaming functor without NEIGHBOURS) share each problem among `groups`
 * workgroups -- one launch for the whole solve, a line-search trial at the chip's bandwidth instead of one CU's.  Each
 * workgroup owns a contiguous range of every thread's elements; every sum is then: the workgroups' sums in the usual order,
 * added left to right -- so the last bits of a result depend on `groups`, which this call reports for a batch on the
 * current device (1: none).  FL_COOP_GROUPS=<g> in the environment overrides the choice (1: never).  A problem whose
 * workgroups could not all be resident ends with FL_STATUS_NOT_SOLVED (the device was shared with other work: calls that run side
 * by side on one device from several streams or threads of YOURS should set FL_COOP_GROUPS=1; fl_multi_solve's own shards
 * size their groups for the device's whole load). */
int fl_cooperative_groups_for(int solver, int objective, int batch, int n);

/* The fused L-BFGS kernel keeps the newest pairs of its (s, y) ring on the chip (registers, then an LDS ring); this
 * returns how many for a built-in objective and dimension n (0 beyond n = 4096).  With C of them on the chip an
 * iteration with cnt pairs in the ring fetches max(0, 2(cnt-C)) + max(0, 2(cnt-C-2)) rows of 8*npad bytes from HBM
 * (the two oldest pairs are still in the row buffers at the turn-around and fetched once) and stores 2 -- the model bench.py
 * prints next to the measured traffic. */
int fl_lbfgs_onchip_pairs(int objective, int n);

/* Bytes of device workspace: FL_SOLVER_LBFGS -- the (s,y) history ring,
 * [batch][2*memory][padded n] fp64 (n > 4096: plus four vector rows per problem);
 * FL_SOLVER_BFGS -- the inverse Hessians [batch][n][padded n] plus, for n > 128, 16 rows per problem for the vectors of the
 * pending updates (fl_bfgs_deferred_updates); 0 for SD / CG (n > 4096: their four
 * vector rows come from the stream-ordered allocator inside the call). */
size_t fl_workspace_bytes(int solver, int batch, int n, int memory);
/* the same from an option block: BFGS with exact_step > 0 needs three matrices per problem (inverse
 * Hessian, Hessian / Cholesky factor, inverse factor), FL_SOLVER_NEWTON one */
size_t fl_workspace_bytes_for(int solver, int batch, int n, const fl_options *opt);

/* Batched solvers.  x_dev [batch][n] in/out (initial guess -> minimiser);
 * d_dev/b_dev [batch][n] objective data (FL_OBJ_DIAGQUAD only, else NULL);
 * outputs (each may be NULL): f_dev[batch] objective at exit, gg_dev[batch]
 * squared gradient norm at exit, iters_dev[batch] line searches performed,
 * status_dev[batch] FL_STATUS_*, nf_dev/ng_dev[batch] f / gradient evaluations
 * as the reference would have issued them. */
int fl_lbfgs_batched(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                     const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                     double *gg_dev, int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                     void *stream);
int fl_conjugate_gradient_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                  const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                  int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                  void *stream);
int fl_steepest_descent_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                                const double *b_dev, const fl_options *opt, double *f_dev, double *gg_dev,
                                int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev,
                                void *stream);

/* Dense BFGS without exact Hessian -- the reference's ExactStep <= 0 path (subroutine BFGS,
 * NO.f90:632-1022: first step 683-716, After_NoHessian 996-1015; C++ binding
 * cpp/NonlinearOptimization.hpp:326-342).  The inverse-Hessian update
 * H <- U^T (H U) + rho s s^T, U = I - rho y s^T (NO.f90:1010-1013) is evaluated in its
 * algebraically equal rank-2 form, H - rho q s^T - rho s q^T + (rho^2 y.q + rho) s s^T with q = H y,
 * as two streaming passes over H (24 n^2 bytes per iteration instead of 4 n^3 flops).
 * workspace: the column-major inverse Hessians, fl_workspace_bytes(FL_SOLVER_BFGS, batch, n, 0)
 * = batch * n * npad * 8 bytes.  max_iteration bounds the main loop (one more search
 * precedes it, NO.f90:689-701). */
int fl_bfgs_batched(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                    const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev, double *gg_dev,
                    int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream);

/* NewtonRaphson with the built-in objective's analytic Hessian (subroutine NewtonRaphson with fdd present,
 * NO.f90:1026-1271; C++ binding cpp/NonlinearOptimization.hpp:344-358): p = -H^{-1} g by Cholesky solve
 * (My_dposv, LinearAlgebra.f90:719-730), steepest-descent fallback when H is not positive definite
 * (NO.f90:1068-1075, 1233-1237).  workspace: fl_workspace_bytes_for(FL_SOLVER_NEWTON, ...). */
int fl_newton_raphson_batched(int objective, int batch, int n, double *x_dev, const double *d_dev,
                              const double *b_dev, const fl_options *opt, void *workspace_dev,
                              size_t workspace_bytes, double *f_dev, double *gg_dev, int32_t *iters_dev,
                              int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream);

/* LinearAlgebra primitives of the path for a batch of SPD matrices (column-major [batch][n][ld], ld as
 * fl_reduction_geometry's threads*ept, lower triangle referenced):
 *   fl_dposv_batched  <- My_dposv  LinearAlgebra.f90:719-730: A <- Cholesky factor, b [batch][n] <- A^{-1} b
 *   fl_dpotri_batched <- My_dpotri LinearAlgebra.f90:798-812 + dsyL2U 260-265: A <- A^{-1} (both triangles);
 *                        work_dev: one more [batch][n][ld] buffer
 * info_dev[batch]: 0 or the index of the first non-positive pivot.  Such a matrix does not disturb the others of the
 * batch; fl_dposv_batched leaves its b untouched and its A partially factorised (as far as the failure: with the blocked
 * path the block columns before the failing 64-column block, and that block's diagonal part up to the pivot; behind it
 * the Schur complement of the completed block steps, like a right-looking LAPACK dpotrf); fl_dpotri_batched leaves its A unspecified (the reference's My_dpotri skips dpotri then and
 * nothing reads the partial factor).  Any batch size (launches go in chunks of 65535 matrices where a grid dimension
 * carries the matrix index). */
/* Size: n < 512 runs one workgroup per matrix with sums in sequential order (what the oracle replays bit for bit);
 * from n = 512 on -- any n, also beyond 4096 -- fl_dposv_batched / fl_dpotri_batched run a blocked right-looking
 * Cholesky with many workgroups per matrix whose O(n^3) part is on the f64 matrix cores (csrc/fl_chol_blocked.hip);
 * results then agree with LAPACK to rounding, not with the sequential kernel bit for bit.  fl_set_chol_blocked_min_n
 * moves the threshold (returns the old one; also FL_CHOL_BLOCKED_MIN_N in the environment). */
int fl_set_chol_blocked_min_n(int n);
int fl_dposv_batched(int batch, int n, double *A_dev, double *b_dev, int32_t *info_dev, void *stream);
int fl_dpotri_batched(int batch, int n, double *A_dev, double *work_dev, int32_t *info_dev, void *stream);
/*   fl_dsysv_batched  <- My_dsysv  LinearAlgebra.f90:695-703 (LAPACK dsysv 'L'): symmetric INDEFINITE systems given by
 *                        their lower triangle (the KKT systems of LagrangianMultiplier, NO.f90:1984); A is destroyed,
 *                        b [batch][n] <- A^{-1} b; info: 0, or k+1 if no non-zero pivot exists at step k (b untouched).
 *                        Elimination with partial pivoting; n <= 4096. */
int fl_dsysv_batched(int batch, int n, double *A_dev, double *b_dev, int32_t *info_dev, void *stream);

/* LinearAlgebra.f90's My_dgemm / My_dgemm_T (182-196; cpp/FortranLibrary.hpp:50-63) on device pointers:
 * C(M,N) = op(A) B, column-major, alpha = 1, beta = 0; transA = 0: A is M x K (lda >= M), transA = 1: A is K x M
 * (lda >= K); B is K x N (ldb >= K); C ldc >= M.  f64 matrix cores (v_mfma_f64_16x16x4_f64), 128 x 128 tiles. */
int fl_dgemm(int transA, int M, int K, int N, const double *A_dev, int lda, const double *B_dev, int ldb, double *C_dev,
             int ldc, void *stream);
/* My_dsyev (LinearAlgebra.f90:879-887) by cyclic two-sided Jacobi: A_dev n x n column-major (lda = n), lower triangle
 * referenced, destroyed.  w_dev[n] <- the eigenvalues UNSORTED; jobz = 'V': the matching normalised eigenvectors are
 * the columns of the n x n matrix at (double *)workspace_dev + n*n.  *sweeps_out (host, may be NULL): sweeps used,
 * negative if max_sweeps were not enough.  The call synchronises the stream once per sweep (convergence test).
 * The legacy symbol __linearalgebra_MOD_my_dsyev sorts ascending and copies back to the host arrays. */
size_t fl_dsyev_workspace_bytes(int n);
int fl_dsyev_jacobi(char jobz, int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes,
                    int max_sweeps, int *sweeps_out, void *stream);
/* jobz = 'N' without the eigenvectors' cost (My_dsyev('N',...), LinearAlgebra.f90:879-887): Householder
 * tridiagonalisation, one launch per reflector, then the tridiagonal's eigenvalues by multisection with Sturm counts;
 * w_dev ascending, A_dev destroyed, n <= 6144, workspace fl_dsyev_workspace_bytes(n). */
int fl_dsyev_values(int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes, void *stream);
/* jobz = 'V' for one large matrix (My_dsyev('V',...), LinearAlgebra.f90:879-887; csrc/fl_eig_vectors.hip): the same
 * tridiagonalisation with the reflectors kept, the tridiagonal's eigenvectors by inverse iteration (one lane per
 * vector), made orthonormal all at once by Cholesky-QR on the f64 matrix cores, back-transformed with one eigenvector
 * per wave.  A_dev (lda = n, lower triangle referenced) <- the normalised eigenvectors (columns), w_dev ascending;
 * n <= 6144.  The basis is checked on the device (orthogonality and tridiagonal residuals to 512 eps) before it is
 * returned: FL_OK, or 1 = the check failed, A_dev / w_dev undefined -- run fl_dsyev_jacobi on a fresh copy, as the legacy
 * symbol does.  quality_host (may be NULL): max |Y Y^T - I|, max |T y - lambda y| / ||T||, Cholesky-QR passes.
 * Synchronises the stream. */
size_t fl_dsyev_vectors_workspace_bytes(int n);
int fl_dsyev_vectors(int n, double *A_dev, int lda, double *w_dev, void *workspace_dev, size_t workspace_bytes,
                     double *quality_host, void *stream);

/* The BFGS inverse-Hessian update AS THE REFERENCE WRITES IT: U = I - rho y s^T, rho = 1/(y.s),
 * H <- matmul(transpose(U), matmul(H, U)) + rho s s^T  (NO.f90:958-962; LinearAlgebra.f90:105-114
 * vector_direct_product) -- two dense n^3 products on the f64 matrix cores (v_mfma_f64_16x16x4_f64),
 * 4 n^3 flop per problem.  H_dev: [batch][n][ld] column-major inverse Hessians in the solver's layout
 * (ld = threads*ept of fl_reduction_geometry), updated in place; s_dev, y_dev [batch][n].
 * workspace: one n x ld product buffer (+ n + 1 doubles) per problem of a chunk; the batch is processed
 * in chunks of workspace_bytes / fl_bfgs_update_gemm_workspace_bytes(1, n) problems.
 * (fl_bfgs_batched itself uses the algebraically equal O(n^2) rank-2 form.) */
size_t fl_bfgs_update_gemm_workspace_bytes(int chunk, int n);
int fl_bfgs_update_gemm_batched(int batch, int n, double *H_dev, const double *s_dev, const double *y_dev,
                                void *workspace_dev, size_t workspace_bytes, void *stream);

/* Augmented Lagrangian for equality constraints (subroutine AugmentedLagrangian, NO.f90:2005-2241;
 * C++ binding cpp/NonlinearOptimization.hpp:367-392) around solver = FL_SOLVER_LBFGS (NO.f90:2150-2167),
 * FL_SOLVER_CG (NO.f90:2168-2185), FL_SOLVER_BFGS (NO.f90:2131-2148: quasi-Newton with opt->exact_step <= 0, or with the
 * exact inverse Hessian of L every exact_step iterations) or FL_SOLVER_NEWTON (NO.f90:2074-2130); the last two take the
 * analytic Hessian of the augmented Lagrangian as the reference's Ldd forms it (NO.f90:2229-2241; every n of the fused path, n <= 4096).
 * Workspace as for the inner solver alone (fl_workspace_bytes_for).  Built-in constraint family: m block spheres
 * c_j(x) = sum_{i in block j} x_i^2 - 1 over m consecutive blocks of n/m elements (m = 1 is the unit
 * sphere of the reference's test, test/test.f90:699-721); m <= 16, n % m == 0.
 * lambda_dev [batch][m] in/out (lambda0 -> final multipliers), miu0 as the reference (clamped to >= 1).
 * As in the reference, opt->precision is both the inner gradient tolerance and the outer ||c||
 * tolerance, opt->max_iteration bounds outer and inner loops, opt->increment is both the line-search
 * growth factor and the miu growth factor, and the inner solver always runs with f_fd present.
 * Outputs: f_dev = augmented Lagrangian at exit, cnorm2_dev = c.c at exit, iters_dev = inner
 * iterations (all outer rounds), outer_dev = outer iterations, status_dev = FL_STATUS_CONVERGED
 * (||c|| < Precision) or FL_STATUS_MAXIT, nf/ng = objective / gradient evaluations. */
int fl_augmented_lagrangian_batched(int solver, int objective, int batch, int n, int m, double *x_dev,
                                    const double *d_dev, const double *b_dev, double *lambda_dev, double miu0,
                                    const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                                    double *cnorm2_dev, int32_t *iters_dev, int32_t *outer_dev, int32_t *status_dev,
                                    int32_t *nf_dev, int32_t *ng_dev, void *stream);

/* How fl_augmented_lagrangian_batched will run `batch` problems on the current device (what bench.py reports next to its
 * timings): the number of launches ("stages"), and per stage the waves per problem and the number of unfinished problems at
 * which the stage hands over to the next one (0: runs to the end).  Where the kernel has helper-wave forms (L-BFGS / CG inside,
 * diagonal-quadratic or quartic objective, 128 < n <= 512, block widths 32 / 64 / 128) a batch that under-fills the device
 * starts with helper waves, and any batch ENDS with them: the unfinished problems pause at an outer iteration's boundary and
 * continue in a launch with more waves per problem.  Results do not depend on the plan (bit-identical). */
int fl_augmented_lagrangian_launch_plan(int solver, int objective, int batch, int n, int m, int *waves, int *pause_below, int max_stages);

/* ---- YOUR objective inside the fused kernels, compiled at run time (csrc/fl_user_rtc.hip) ----------------------------
 * The reference takes the objective as callbacks: subroutine f(fx,x,dim), fd(g,x,dim) (NO.f90:33-38).  A caller without hipcc
 * in its build -- Python, Fortran, C -- passes the objective as HIP SOURCE TEXT: the functor of include/fl_user_objective.hpp
 *     template <int NW, int EPT> struct <class_name> {
 *         static constexpr int LDS_DOUBLES = ...;                       // doubles of LDS scratch eval() wants
 *         __device__ void init(const fl::SolveArgs &A, int prob, double *lds);   // A.n, A.d, A.b, A.user: the data pointers
 *         __device__ void eval(const double (&x)[EPT], double (&g)[EPT], double &s0, double &s1, int n, double *lds);
 *         __device__ static double combine(double s0, double s1);       // f = combine(sum of s0, sum of s1)
 *     };
 * (layout of the thread's EPT elements and the helpers fl::load_user / fl::Geo: that header).  fl_user_compile builds
 * fl_solve_kernel<geometry of n, FL_OBJ_USER, solver> around it with hiprtc -- about a second; the kernel's own headers are
 * embedded in libFL.so, libhiprtc.so is opened on first use -- and fl_user_solve runs a batch with it: the fused kernel's
 * speed (the reverse-communication form is 22 x slower on the headline workload), its geometry and summation order -- a
 * functor restating a built-in objective reproduces it bit for bit.
 *   solver     FL_SOLVER_SD | CG | LBFGS | BFGS (quasi-Newton updates: exact_step is taken as 0)
 *   n          <= 4096: the class template above.  n > 4096 (the vectors live in HBM, 1024 threads per problem): class_name
 *              names a plain class with the STREAMING interface -- the objective is asked one element pair at a time, the
 *              solver does the loads, stores and fixed-order sums around it (include/fl_user_stream_objective.hpp):
 *                  struct <class_name> {
 *                      static constexpr bool NEIGHBOURS = ...;      // pair() reads x of other elements through `x`
 *                      __device__ void init(const fl::SolveArgs &A, int prob);
 *                      __device__ void pair(int e, int n, const double *x, double xa, double xb,   // elements e (even), e + 1
 *                                           double &ta, double &tb, double &ua, double &ub, double &ga, double &gb);
 *                      __device__ static double combine(double s0, double s1);   // s0 = sum of ta, tb; s1 = sum of ua, ub
 *                  };
 *              SD / CG / L-BFGS at any n, BFGS up to n = 16384 (its dense H); tune_like is not used; not inside the
 *              augmented Lagrangian.  A class restating a built-in objective reproduces the built-in kernel bit for bit here too.
 *   tune_like  whose register budget the kernel is tuned like: FL_OBJ_DIAGQUAD for an element-wise objective with at most two
 *              data vectors in registers, FL_OBJ_USER_TUNE_NONE (4) when in doubt (no occupancy caps: nothing can spill)
 *   log        (may be NULL) receives the compiler's messages, truncated to log_bytes
 * Returns FL_OK; FL_ERR_INVALID_ARGUMENT: the source does not compile (see the log); FL_ERR_LAUNCH: no run-time compiler.
 * fl_user_solve: x_dev [batch][n] in/out; data0_dev / data1_dev [batch][n] and params_dev arrive in the functor as A.d, A.b,
 * A.user; options, workspace (fl_workspace_bytes_for) and outputs as fl_lbfgs_batched.  FL_RTC_CACHE_DIR in the environment:
 * a directory where compiled code objects are kept between processes.
 * fl_user_compile_check compiles only (no device needed; arch e.g. "gfx950"): for a build or CI step. */
#define FL_OBJ_USER_TUNE_NONE 4
typedef struct fl_user_objective fl_user_objective;
int fl_user_compile(fl_user_objective **handle, const char *source, const char *class_name, int solver, int n, int tune_like,
                    char *log, size_t log_bytes);
int fl_user_compile_check(const char *source, const char *class_name, int solver, int n, int tune_like, const char *arch, char *log,
                          size_t log_bytes);
int fl_user_geometry(const fl_user_objective *handle, int *threads, int *ept);
int fl_user_solve(fl_user_objective *handle, int batch, double *x_dev, const double *data0_dev, const double *data1_dev,
                  const void *params_dev, const fl_options *opt, void *workspace_dev, size_t workspace_bytes, double *f_dev,
                  double *gg_dev, int32_t *iters_dev, int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream);
/* AugmentedLagrangian (NO.f90:2005-2241) around the caller's objective -- and, where constraints_class_name is given, the caller's
 * CONSTRAINTS: the reference's callbacks c(cx,x,M,N), cd(cdx,x,M,N) (NO.f90:1928-1934) as a second class template in the same source
 *     template <int NW, int EPT> struct <constraints_class_name> {
 *         __device__ void init(const fl::SolveArgs &A, int prob);                              // A.aug_m = M (<= 8), A.user
 *         __device__ void partial(const double (&x)[EPT], double (&cpart)[8], int n);           // c_j(x) = sum over threads of cpart[j] + offset(j)
 *         __device__ double offset(int j) const;                                                // the constant of c_j (added after the sum)
 *         __device__ void add_gradient(const double (&x)[EPT], const double (&v)[8], double (&g)[EPT], int n);  // g += sum_j v_j grad c_j
 *     };
 * (csrc/fl_device.hpp, FL_USER_CONSTRAINTS).  constraints_class_name NULL or "": the library's constraint family (m block
 * spheres, fl_augmented_lagrangian_batched; an objective tuned like FL_OBJ_DIAGQUAD then gets the tight objective-only loops of the
 * line search too).  solver: FL_SOLVER_LBFGS | FL_SOLVER_CG as the inner solver.  fl_user_solve_auglag: arguments as
 * fl_augmented_lagrangian_batched (m <= 8 with the caller's constraints). */
int fl_user_compile_auglag(fl_user_objective **handle, const char *source, const char *class_name, const char *constraints_class_name,
                           int solver, int n, int tune_like, char *log, size_t log_bytes);
int fl_user_compile_check_auglag(const char *source, const char *class_name, const char *constraints_class_name, int solver, int n,
                                 int tune_like, const char *arch, char *log, size_t log_bytes); /* compiles only, no device */
int fl_user_solve_auglag(fl_user_objective *handle, int batch, int m, double *x_dev, const double *data0_dev, const double *data1_dev,
                         const void *params_dev, double *lambda_dev, double miu0, const fl_options *opt, void *workspace_dev,
                         size_t workspace_bytes, double *f_dev, double *cnorm2_dev, int32_t *iters_dev, int32_t *outer_dev,
                         int32_t *status_dev, int32_t *nf_dev, int32_t *ng_dev, void *stream);
int fl_user_destroy(fl_user_objective *handle);

/* ---- all the GPUs of the node from one process --------------------------------------------------
 * The batched solvers above for HOST arrays, sharded over the visible devices by one host thread per shard (hipSetDevice,
 * upload, one launch of the fused kernel on the shard's own stream, the shard's rows of the results written straight
 * into the caller's arrays).  Problems are independent: no collective, no exchange beyond that write-back.
 *   solver     FL_SOLVER_SD | CG | LBFGS | BFGS | NEWTON;  objective, opt, outputs: as in the one-device entries
 *   aug_m > 0  AugmentedLagrangian around `solver` with aug_m block-sphere constraints (fl_augmented_lagrangian_batched):
 *              lambda_host [batch][aug_m] in (NULL: lambda0 = 0) / out, miu0, cnorm2_host, outer_host; gg_host unused
 *   nshards    <= 0: up to four shards per visible device (while a shard keeps >= 4096 problems: the shards of a device
 *              overlap each other's transfers and solves); more shards than devices share devices round-robin
 *   interleaved  0: contiguous blocks of ceil(batch / nshards) problems;  1: problem k -> shard k mod nshards
 * Every output pointer may be NULL.  Results do not depend on the sharding (one workgroup owns one problem either way).
 * FL_MULTI_PIN=1 in the environment page-locks the big host arrays (x, d, b) in place for the duration of the call
 * (hipHostRegister, best effort) so that the shards' copies go by DMA; off by default (measured: no gain, csrc/fl_multi.cpp).
 * Return: FL_OK, or the first failing shard's error; then status_host (if given) carries FL_STATUS_NOT_SOLVED in the rows
 * of the shards that failed -- their rows of x_host and of the other outputs are untouched, every other row is final. */
int fl_multi_device_count(void);
int fl_multi_solve(int solver, int objective, int batch, int n, double *x_host, const double *d_host, const double *b_host,
                   const fl_options *opt, int aug_m, double *lambda_host, double miu0, double *f_host, double *gg_host,
                   double *cnorm2_host, int32_t *iters_host, int32_t *outer_host, int32_t *status_host, int32_t *nf_host,
                   int32_t *ng_host, int nshards, int interleaved);

/* ---- reverse communication: objectives evaluated by the caller -------------------------------
 * The reference calls back into user code for every evaluation (subroutine f(fx,x,dim), fd(g,x,dim),
 * integer function f_fd(fx,g,x,dim): NO.f90:33-38).  For a batch on the GPU the same protocol is an
 * "ask / tell" loop: the solver machines live in device memory; every fl_rci_step
 *     1. takes the caller's evaluations f_dev[batch], g_dev[batch][n] at the points it asked for,
 *     2. advances every problem to its next request,
 *     3. overwrites x_dev[batch][n] with the next points and sets request_dev[batch]:
 *        bit 0 (1) evaluate f, bit 1 (2) evaluate grad f, bit 2 (4) the point is unchanged since the
 *        previous request (only the newly requested quantity is read), 0 = this problem has finished
 *        (x_dev holds its minimiser and is no longer touched).
 * The bits are exactly the reference's callback pattern (f only while Armijo fails, f' only once it
 * holds, f_fd when both are set and opt->fused_f_fd is on), so wrapping host callbacks reproduces the
 * reference's evaluation counts.  First call: x_dev = initial guesses, f_dev/g_dev may be NULL (afterwards only with a
 * solver that takes Hessians, on a step that delivers nothing else; otherwise FL_ERR_INVALID_ARGUMENT).
 * Arrays whose bit was not requested are not read.  solver: FL_SOLVER_SD | CG | LBFGS | BFGS
 * (BFGS: ExactStep <= 0 path).  The handle owns its device buffers (history ring / inverse Hessians,
 * parked machine state).
 * Hessians (integer function fdd(f''(x),x,dim), NO.f90:37): with FL_SOLVER_NEWTON, or FL_SOLVER_BFGS and
 * opt->exact_step > 0, a request may carry bit 4 (16): write f''(x) of the current (unchanged) x for that problem
 * into the handle's buffer (fl_rci_hessian_buffer: [batch][n][ld] column-major, problem k at k*n*ld for Newton
 * and at (3k+1)*n*ld for BFGS) and call fl_rci_step again; f_dev / g_dev are not read on that step. */
typedef struct fl_rci fl_rci;
int fl_rci_create(fl_rci **handle, int solver, int batch, int n, const fl_options *opt, void *stream);
int fl_rci_hessian_buffer(fl_rci *handle, double **hessian_dev, int *ld);
/* n > 4096 with few problems (batch < 128): the step kernel runs COOPERATIVELY -- several workgroups share one problem
 * (each owns a range of the vector, all run the same scalar machine and meet in every reduction), as many as stay resident
 * together (256 / batch) -- so that ONE problem of n = 10^6 occupies the chip instead of one CU.  The number in use (1 = not
 * cooperative); sums are then taken per workgroup and added left to right (bit-reproducible; oracle: flo_set_sum_groups).
 * FL_COOP_GROUPS in the environment at fl_rci_create overrides the choice (1 = off). */
int fl_rci_cooperative_groups(fl_rci *handle);
int fl_rci_step(fl_rci *handle, double *x_dev, const double *f_dev, const double *g_dev, int32_t *request_dev);
/* The same step with flags (FL_SOLVER_SD | CG | LBFGS).  FL_RCI_BOTH: the caller evaluated f AND grad f at every requested point, whatever the bits
 * asked for (the natural form of a torch / HIP objective, the reference's f_fd).  The kernel then answers a request for the
 * other quantity at the SAME point (bit 2: StrongWolfe asks for f, then for f' once Armijo holds, NO.f90:1483-1485) by
 * itself instead of handing it back, so the caller sees one round per distinct trial point -- like the fused kernels --
 * and the reference's evaluation counts (nf, ng) stay what they were. */
#define FL_RCI_BOTH 1
int fl_rci_step_flags(fl_rci *handle, double *x_dev, const double *f_dev, const double *g_dev, int32_t *request_dev, int flags);
/* The step for a caller that COMPACTS (FL_SOLVER_SD | CG | LBFGS): only the n_active problems listed in active_dev
 * [n_active] (problem ids, any order) are stepped, and the arrays of the step -- xc_dev [n_active][n] (in: the points just
 * evaluated, out: the next requested points), f_dev [n_active], g_dev [n_active][n], request_dev [n_active] -- are indexed
 * by the position in that list.  x_dev [batch][n] receives a problem's minimiser when it finishes (request 0).  Between
 * two calls the caller may drop finished problems from the list, moving the rows of xc_dev along; the tail of a batch
 * -- a few slow problems -- then costs evaluations of those problems only.  First call: every problem listed, xc_dev =
 * the initial guesses.  A finished problem that is still listed costs one word read. */
int fl_rci_step_compact(fl_rci *handle, double *x_dev, const int32_t *active_dev, int n_active, double *xc_dev,
                        const double *f_dev, const double *g_dev, int32_t *request_dev, int flags);
/* copies the per-problem outputs (device pointers, each may be NULL) after all requests are 0 */
int fl_rci_results(fl_rci *handle, double *f_dev, double *gg_dev, int32_t *iters_dev, int32_t *status_dev,
                   int32_t *nf_dev, int32_t *ng_dev);
int fl_rci_destroy(fl_rci *handle);

/* ---- AugmentedLagrangian (NO.f90:2005-2241) for a batch with the CALLER's objective AND constraints -------------
 * The reference's callbacks c(cx,x,M,N) and cd(cdx(N,M),x,M,N) (NO.f90:1928-1934) join the ask / tell loop: the handle
 * wraps the inner solver (FL_SOLVER_LBFGS, NO.f90:2150-2167, FL_SOLVER_CG, 2168-2185, or FL_SOLVER_BFGS with
 * opt->exact_step <= 0, 2131-2148: quasi-Newton updates, H rebuilt from a I in every outer round; always with f_fd
 * present like the reference, 2153) in the outer loop lambda <- lambda - miu c, miu <- miu * Increment (2155-2157), per problem and
 * inside the step kernel.  Requests carry two more bits: FL_RCI_REQ_C (32) -- with EVERY request: c_dev[batch][m] =
 * c(x) at the requested point -- and FL_RCI_REQ_CD (64) -- with every gradient request: cd_dev[batch][m][n], row j =
 * grad c_j(x) (the Fortran array cdx(N,M) as it lies in memory).  L = f - lambda.c + miu/2 c.c and
 * grad L = grad f + cd^T (miu c - lambda) (NO.f90:2198, 2205) are formed in the kernel; m <= 16.  Any n: beyond 4096 the
 * machine's vectors live in HBM (1024 threads per problem, fl_reduction_geometry; FL_SOLVER_BFGS to n = 16384), whole-batch steps.
 * lambda_dev [batch][m]: lambda0 on entry, the multipliers afterwards (updated in place from step to step; must stay
 * valid while the handle lives).  opt->precision is the inner gradient tolerance AND the outer ||c|| tolerance,
 * opt->max_iteration bounds both loops, opt->increment is the line-search growth factor AND the miu growth factor
 * (as in the reference).  fl_rci_results: f = the augmented Lagrangian at exit, iters = inner iterations of all outer
 * rounds, status = FL_STATUS_CONVERGED (||c|| < Precision) | FL_STATUS_MAXIT; fl_rci_results_auglag: c.c at exit and
 * the number of outer iterations.
 * NewtonRaphson (FL_SOLVER_NEWTON, NO.f90:2074-2130) and BFGS with opt->exact_step > 0 (2131-2148) as inner solvers ask
 * for the HESSIAN OF L (bit 4, FL_REQ_H = 16, at an unchanged point): the caller forms it as the reference's Ldd does
 * (NO.f90:2229-2241),  Ldd = f'' + sum_j c_j'' (miu c_j - lambda_j) + cd cd^T  -- lambda from lambda_dev, miu from
 * fl_rci_auglag_miu -- writes it with fl_rci_put_hessians (or into fl_rci_hessian_buffer) and steps again with the same f,
 * g, c, cd arrays; n <= 4096 for these two (dense Cholesky: the register path).  Without f'' / c'' the reference differentiates grad L by MKL's djacobi:
 * fl_fd_points / fl_fd_column do that for a batch with djacobi's step rule (2n gradient evaluations per Hessian). */
#define FL_RCI_REQ_C 32
#define FL_RCI_REQ_CD 64
int fl_rci_create_auglag(fl_rci **handle, int solver, int batch, int n, int m, double *lambda_dev, double miu0,
                         const fl_options *opt, void *stream);
int fl_rci_step_auglag(fl_rci *handle, double *x_dev, const double *f_dev, const double *g_dev, const double *c_dev,
                       const double *cd_dev, int32_t *request_dev);
int fl_rci_results_auglag(fl_rci *handle, double *cnorm2_dev, int32_t *outer_dev);
int fl_rci_auglag_miu(fl_rci *handle, double *miu_dev); /* [batch]: the penalty parameter of each problem's current outer round */
/* Hessians for the problems whose request carries bit 4 (request_dev == NULL: all): H_dev [batch][n][n] dense (symmetric, so
 * row- and column-major agree) -> the handle's padded buffer.  Works for fl_rci_create (f'') and fl_rci_create_auglag (Ldd). */
int fl_rci_put_hessians(fl_rci *handle, const double *H_dev, const int32_t *request_dev);
/* Central differences of a gradient for a batch with MKL djacobi's step rule (see fl_djacobi): for j = 0 .. n-1:
 * fl_fd_points(j): xp_dev / xm_dev [batch][n] <- x with coordinate j at x_j (1 +- eps), or x_j +- eps where |x_j| <= eps;
 * evaluate the gradients gp at xp, gm at xm; fl_fd_column(j): column j of every H_dev [batch][n][n] <- (gp - gm) * (0.5 / h_j). */
int fl_fd_points(int batch, int n, int j, double eps, const double *x_dev, double *xp_dev, double *xm_dev, void *stream);
int fl_fd_column(int batch, int n, int j, double eps, const double *x_dev, const double *gp_dev, const double *gm_dev,
                 double *H_dev, void *stream);

/* ---- central-difference Jacobian: the library's replacement of MKL's djacobi, which the reference calls for f'' when
 * the caller passes no fdd (NO.f90:676, 981, 1067, 1258: djacobi(fd_j,dim,dim,H,x,1d-8)) and for TrustRegion's Jacobian
 * (NO.f90:1779, 1833).  MKL's argument convention: fcn(m, n, x, f) with everything by reference, fjac(m, n) column-major,
 * returns 1501 (TR_SUCCESS) or 1502 (invalid argument); x is restored.  Step rule = djacobi's own, held to the real MKL
 * routine bit for bit (tests/golden/mkl_djacobi.npz):  |x_j| > eps: f at x_j (1 +- eps), h = eps x_j;  else f at
 * x_j +- eps, h = eps;  fjac(:,j) = (f_plus - f_minus) * (0.5 / h).  Host code, 2n calls of fcn. */
int fl_djacobi(void (*fcn)(const int *m, const int *n, const double *x, double *f), const int *n, const int *m,
               double *fjac, double *x, const double *eps);

/* ---- TrustRegion (NO.f90:1728-1906) for a batch, on the device, by reverse communication ------------------------
 * Solves f'(x) = 0 in the least-squares sense (M equations, N unknowns, M >= N; optional box low <= x <= up shared by
 * the batch) for `batch` independent problems.  The reference wraps MKL's closed dtrnlsp solver; this is the library's
 * own Levenberg-Marquardt iteration (own path; end points held to the real dtrnlsp's) behind the reference's stopping options
 * (MaxIteration, MaxStepIteration, Precision on ||f'(x)||_2, MinStepLength on the step).  Ask / tell:
 *   fl_trust_region_step(h, x_dev, r_dev, J_dev, request_dev): first call x_dev [batch][N] = starting points (r_dev,
 *   J_dev may be NULL); afterwards r_dev [batch][M] = f'(x) and J_dev [batch][N][M] = the M x N Jacobian, column-major
 *   (the Fortran array Jacobian(M,N)), evaluated at x_dev[k] where request_dev[k] asked for them: bit FL_TRS_REQ_R (1)
 *   residual, FL_TRS_REQ_J (2) Jacobian, FL_TRS_REQ_AGAIN (4) nothing to evaluate for this problem but it has not
 *   finished, 0 finished (x_dev[k] = the solution).  What the arrays hold for a problem that did NOT ask is ignored:
 *   J^T J and J^T r are kept per problem inside the handle and renewed only from a Jacobian the problem requested, so
 *   a caller may as well evaluate r and J for the whole batch at x_dev on every step.
 * fl_trust_region_results: ||f'(x)||_2, accepted iterations, stopping reason (1 MaxIteration, 2 MaxStepIteration,
 * 3 Precision met, 4 stationary point of |f'|^2, 5 MinStepLength). */
#define FL_TRS_REQ_R 1
#define FL_TRS_REQ_J 2
#define FL_TRS_REQ_AGAIN 4
typedef struct fl_trs fl_trs;
int fl_trust_region_create(fl_trs **handle, int batch, int M, int N, const double *low_dev, const double *up_dev,
                           int max_iteration, int max_step_iteration, double precision, double min_step_length, void *stream);
int fl_trust_region_step(fl_trs *handle, double *x_dev, const double *r_dev, const double *J_dev, int32_t *request_dev);
int fl_trust_region_results(fl_trs *handle, double *resnorm_dev, int32_t *iters_dev, int32_t *reason_dev);
int fl_trust_region_destroy(fl_trs *handle);

/* The L-BFGS two-loop recursion alone (Before(), NO.f90:586-608) for a batch:
 * p = -H_k g from a full ring of `memory` pairs.  hist_dev is the solver's
 * history layout [batch][2*memory][npad] (npad = threads*ept; pair i: s at
 * row 2i, y at row 2i+1), rho_dev [batch][memory], recent = newest slot.
 * g_dev, p_dev [batch][n].  Used by bench.py to measure the recursion against
 * the HBM roofline in isolation. */
int fl_lbfgs_two_loop_batched(int batch, int n, int memory, int recent, const double *hist_dev,
                              const double *rho_dev, const double *g_dev, double *p_dev, void *stream);

/* Synthetic inputs, generated on the device with Philox-4x32-10 (key = seed,
 * counter = (element pair, problem id)); tests replay the same generator with numpy.
 * out[batch][n] uniform in (lo, hi). */
int fl_synth_uniform(uint64_t seed, int batch, int n, double lo, double hi, double *out_dev, void *stream);
/* d[k][i] = 1 + (kappa_k - 1) i/(n-1), kappa_k log-uniform in [kappa_lo, kappa_hi] (one Philox draw per problem) */
int fl_synth_diag_spectrum(uint64_t seed, int batch, int n, double kappa_lo, double kappa_hi, double *d_dev,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif
