/*
 * fl_legacy.h -- the reference's own single-problem entry points, exported by libFL.so (MI355X)
 * under the mangled names the reference's C++ header binds, so existing callers of
 * `#include <FortranLibrary.hpp>` / `FL::NO::*` keep linking against -lFL unchanged.
 *
 * Replaces (file:line in /root/reference):
 *   __nonlinearoptimization_MOD_steepestdescent          cpp/NonlinearOptimization.hpp:279-292   (NO.f90:55)
 *   __nonlinearoptimization_MOD_conjugategradient_basic  cpp/NonlinearOptimization.hpp:294-307   (NO.f90:2249)
 *   __nonlinearoptimization_MOD_conjugategradient        cpp/NonlinearOptimization.hpp:309-324   (NO.f90:193)
 *   __nonlinearoptimization_MOD_bfgs                     cpp/NonlinearOptimization.hpp:326-342   (NO.f90:632)
 *   __nonlinearoptimization_MOD_newtonraphson            cpp/NonlinearOptimization.hpp:344-358   (NO.f90:1026)
 *   __nonlinearoptimization_MOD_augmentedlagrangian      cpp/NonlinearOptimization.hpp:367-392   (NO.f90:2005)
 *   __nonlinearoptimization_MOD_lbfgs                    (Fortran only in the reference)          (NO.f90:398)
 *   __nonlinearoptimization_MOD_lagrangianmultiplier     (Fortran only in the reference)          (NO.f90:1950)
 *   __nonlinearoptimization_MOD_{wolfe,strongwolfe}[_fdwithf]  (Fortran only in the reference)     (NO.f90:1286-1698)
 *   nonlinearoptimization_mp_*_                          the ifort manglings, hpp:11-123
 * Conventions are the reference's (cpp/README.md:11-18): every argument by reference, an absent
 * Fortran optional = NULL, logical = 4-byte integer (nonzero = true), character(*) = pointer plus a
 * hidden length appended by value.  Callbacks run on the HOST, exactly when the reference would call
 * them; all solver arithmetic runs on the GPU (reverse communication, fl_nlopt.h: fl_rci_*).
 * x is the only result (in/out), warnings go to stdout when Warning is true, like the reference.
 * Differences: an unknown Method prints the reference's message and returns instead of `stop`;
 * AugmentedLagrangian: every inner solver of the reference's menu ('LBFGS', 'ConjugateGradient', 'BFGS',
 * 'NewtonRaphson'; NO.f90:2074-2185); the wrappers L, Ld, L_Ld, Ldd that compose the caller's f, fd, fdd, c, cd, cdd
 * (NO.f90:2193-2240) run on the host next to those callbacks, every inner solve on the GPU.
 * NewtonRaphson and BFGS with ExactStep > 0 call the caller's fdd on the host and ship the Hessian to the GPU
 * (Cholesky solve / inverse there); without fdd the reference calls MKL djacobi: here fl_djacobi's central differences
 * of the caller's fd with djacobi's own step rule, 2n gradient calls per Hessian -- the real MKL routine's bits
 * (tests/golden/mkl_djacobi.npz), so e.g. BFGS with its defaults makes the reference's 844 f / 845 fd callbacks on the
 * Rosenbrock n = 10 probe.
 */
#ifndef FL_LEGACY_H
#define FL_LEGACY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void (*fl_f_cb)(double *fx, const double *x, const int *dim);             /* subroutine f(f(x),x,dim)  */
typedef void (*fl_fd_cb)(double *fdx, const double *x, const int *dim);           /* subroutine fd(f'(x),x,dim) */
typedef int (*fl_f_fd_cb)(double *fx, double *fdx, const double *x, const int *dim); /* integer function f_fd   */
typedef int (*fl_fdd_cb)(double *fddx, const double *x, const int *dim);          /* integer function fdd     */

#define FL_LEGACY_COMMON_ARGS                                                                                   \
    const int32_t *Strong, const int32_t *Warning, const int *MaxIteration, const double *Precision,            \
        const double *MinStepLength, const double *WolfeConst1, const double *WolfeConst2, const double *Increment

void __nonlinearoptimization_MOD_steepestdescent(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_f_fd_cb f_fd,
                                                 FL_LEGACY_COMMON_ARGS);
void __nonlinearoptimization_MOD_conjugategradient_basic(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim,
                                                         const char *Method, FL_LEGACY_COMMON_ARGS, int len_Method);
void __nonlinearoptimization_MOD_conjugategradient(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim,
                                                   const char *Method, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS,
                                                   int len_Method);
void __nonlinearoptimization_MOD_lbfgs(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, const int *Memory,
                                       fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);
void __nonlinearoptimization_MOD_bfgs(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_fdd_cb fdd,
                                      const int *ExactStep, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);

void __nonlinearoptimization_MOD_newtonraphson(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_fdd_cb fdd,
                                               fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);
void nonlinearoptimization_mp_newtonraphson_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_fdd_cb fdd,
                                             fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);
typedef void (*fl_c_cb)(double *cx, const double *x, const int *M, const int *N);    /* subroutine c(c(x),x,M,N)    */
typedef void (*fl_cd_cb)(double *cdx, const double *x, const int *M, const int *N);  /* subroutine cd(c'(x),x,M,N): N x M */
typedef int (*fl_cdd_cb)(double *cddx, const double *x, const int *M, const int *N); /* integer function cdd         */
void __nonlinearoptimization_MOD_augmentedlagrangian(fl_f_cb f, fl_fd_cb fd, fl_c_cb c, fl_cd_cb cd, double *x,
                                                     const int *N, const int *M, const char *UnconstrainedSolver,
                                                     const double *lambda0, const double *miu0, fl_fdd_cb fdd,
                                                     fl_cdd_cb cdd, const int *ExactStep, const int *Memory,
                                                     const char *Method, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS,
                                                     int len_UnconstrainedSolver, int len_Method);
void nonlinearoptimization_mp_augmentedlagrangian_(fl_f_cb f, fl_fd_cb fd, fl_c_cb c, fl_cd_cb cd, double *x,
                                                   const int *N, const int *M, const char *UnconstrainedSolver,
                                                   const double *lambda0, const double *miu0, fl_fdd_cb fdd,
                                                   fl_cdd_cb cdd, const int *ExactStep, const int *Memory,
                                                   const char *Method, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS,
                                                   int len_UnconstrainedSolver, int len_Method);

void nonlinearoptimization_mp_steepestdescent_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_f_fd_cb f_fd,
                                               FL_LEGACY_COMMON_ARGS);
void nonlinearoptimization_mp_conjugategradient_basic_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim,
                                                       const char *Method, FL_LEGACY_COMMON_ARGS, int len_Method);
void nonlinearoptimization_mp_conjugategradient_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim,
                                                 const char *Method, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS,
                                                 int len_Method);
void nonlinearoptimization_mp_lbfgs_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, const int *Memory,
                                     fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);
void nonlinearoptimization_mp_bfgs_(fl_f_cb f, fl_fd_cb fd, double *x, const int *dim, fl_fdd_cb fdd,
                                    const int *ExactStep, fl_f_fd_cb f_fd, FL_LEGACY_COMMON_ARGS);

/* The line searchers are public procedures of the reference module too (NO.f90:1286, 1373, 1462, 1582):
 * subroutine StrongWolfe(c1,c2,f,fd,x,a,p,fx,phid0,fdx,dim,Increment) etc.  In: x, trial step a, direction p,
 * fx = f(x), phid0 = f'(x).p; out: a, x = x + a p, fx = f(x), fdx = f'(x).  Increment may be NULL (1.05). */
void __nonlinearoptimization_MOD_wolfe(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd, double *x, double *a,
                                       const double *p, double *fx, const double *phid0, double *fdx, const int *dim,
                                       const double *Increment);
void __nonlinearoptimization_MOD_wolfe_fdwithf(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd,
                                               fl_f_fd_cb f_fd, double *x, double *a, const double *p, double *fx,
                                               const double *phid0, double *fdx, const int *dim,
                                               const double *Increment);
void __nonlinearoptimization_MOD_strongwolfe(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd, double *x,
                                             double *a, const double *p, double *fx, const double *phid0, double *fdx,
                                             const int *dim, const double *Increment);
void __nonlinearoptimization_MOD_strongwolfe_fdwithf(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd,
                                                     fl_f_fd_cb f_fd, double *x, double *a, const double *p,
                                                     double *fx, const double *phid0, double *fdx, const int *dim,
                                                     const double *Increment);
void nonlinearoptimization_mp_wolfe_(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd, double *x, double *a,
                                     const double *p, double *fx, const double *phid0, double *fdx, const int *dim,
                                     const double *Increment);
void nonlinearoptimization_mp_wolfe_fdwithf_(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd, fl_f_fd_cb f_fd,
                                             double *x, double *a, const double *p, double *fx, const double *phid0,
                                             double *fdx, const int *dim, const double *Increment);
void nonlinearoptimization_mp_strongwolfe_(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd, double *x,
                                           double *a, const double *p, double *fx, const double *phid0, double *fdx,
                                           const int *dim, const double *Increment);
void nonlinearoptimization_mp_strongwolfe_fdwithf_(const double *c1, const double *c2, fl_f_cb f, fl_fd_cb fd,
                                                   fl_f_fd_cb f_fd, double *x, double *a, const double *p, double *fx,
                                                   const double *phid0, double *fdx, const int *dim,
                                                   const double *Increment);

/* subroutine LagrangianMultiplier(fd,fdd,c,cd,cdd,x,lambda,N,M,Warning,MaxIteration,Precision)  NO.f90:1950-1993
 * (Fortran only in the reference).  Warning / MaxIteration / Precision may be NULL (true / 1000 / 1e-15). */
void __nonlinearoptimization_MOD_lagrangianmultiplier(fl_fd_cb fd, fl_fdd_cb fdd, fl_c_cb c, fl_cd_cb cd, fl_cdd_cb cdd,
                                                      double *x, double *lambda, const int *N, const int *M,
                                                      const int32_t *Warning, const int *MaxIteration,
                                                      const double *Precision);
void nonlinearoptimization_mp_lagrangianmultiplier_(fl_fd_cb fd, fl_fdd_cb fdd, fl_c_cb c, fl_cd_cb cd, fl_cdd_cb cdd,
                                                    double *x, double *lambda, const int *N, const int *M,
                                                    const int32_t *Warning, const int *MaxIteration,
                                                    const double *Precision);

/* TrustRegion / TrustRegion_basic (NO.f90:1728-1906, 2348-2423; hpp:358-366): f'(x) = 0 by minimising |f'(x)|^2.
 * The reference wraps MKL's closed dtrnlsp solver; here an own Levenberg-Marquardt iteration sits behind the same
 * interface (callbacks on the host, J^T J / damped normal equations on the GPU): same stationary points -- own path,
 * end points held to the real dtrnlsp's (tests/golden/mkl_trnlsp.npz).  subroutine fd(f'(x),x,M,N); integer function Jacobian(J(x),x,M,N), J is M x N. */
typedef void (*fl_residue_cb)(double *fdx, const double *x, const int *M, const int *N);
typedef int (*fl_jacobian_cb)(double *Jx, const double *x, const int *M, const int *N);
void __nonlinearoptimization_MOD_trustregion_basic(fl_residue_cb fd, fl_jacobian_cb Jacobian, double *x, const int *M,
                                                   const int *N, const int32_t *Warning, const int *MaxIteration,
                                                   const int *MaxStepIteration, const double *Precision,
                                                   const double *MinStepLength);
void nonlinearoptimization_mp_trustregion_basic_(fl_residue_cb fd, fl_jacobian_cb Jacobian, double *x, const int *M,
                                                 const int *N, const int32_t *Warning, const int *MaxIteration,
                                                 const int *MaxStepIteration, const double *Precision,
                                                 const double *MinStepLength);
/* Fortran's general routine: Jacobian, low, up and everything after them may be NULL (absent) */
void __nonlinearoptimization_MOD_trustregion(fl_residue_cb fd, double *x, const int *M, const int *N,
                                             fl_jacobian_cb Jacobian, const double *low, const double *up,
                                             const int32_t *Warning, const int *MaxIteration, const int *MaxStepIteration,
                                             const double *Precision, const double *MinStepLength);
void nonlinearoptimization_mp_trustregion_(fl_residue_cb fd, double *x, const int *M, const int *N,
                                           fl_jacobian_cb Jacobian, const double *low, const double *up,
                                           const int32_t *Warning, const int *MaxIteration, const int *MaxStepIteration,
                                           const double *Precision, const double *MinStepLength);

/* LinearAlgebra entry points the reference's C++ header binds (cpp/FortranLibrary.hpp:48-63; LinearAlgebra.f90:182-196,
 * 879-887).  Host arrays, column-major; computed by this library's own kernels on the GPU (the reference hands them to
 * MKL): fl_dgemm on the f64 matrix cores; My_dsyev 'N' by tridiagonalisation + multisection (fl_dsyev_values), 'V' by
 * tridiagonalisation + inverse iteration + Cholesky-QR + back-transformation (fl_dsyev_vectors; the basis is checked on
 * the device, cyclic Jacobi if the check fails; FL_DSYEV_JACOBI=1 in the environment forces Jacobi).  Like dsyev the
 * matrix is rescaled first when max |a_ij| is outside [1e-100, 1e100].  Same results to rounding, eigenvectors up to
 * sign (and up to a rotation inside numerically multiple eigenvalues). */
void __linearalgebra_MOD_my_dgemm(const double *A, const double *B, double *C, const int *M, const int *K, const int *N);
void __linearalgebra_MOD_my_dgemm_t(const double *A, const double *B, double *C, const int *M, const int *K, const int *N);
void __linearalgebra_MOD_my_dsyev(const char *jobtype, double *A, double *eigval, const int *N, int len_jobtype);
void linearalgebra_mp_my_dgemm_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N);
void linearalgebra_mp_my_dgemm_t_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N);
void linearalgebra_mp_my_dsyev_(const char *jobtype, double *A, double *eigval, const int *N, int len_jobtype);

/* Import-time symbols of the reference's Python package (FortranLibrary/General.py:4-16 probes
 * general_mp_showtime_ / __general_MOD_showtime when the package is imported, so `CDLL('libFL.so')` users keep
 * importing it against this library).  Host utilities, restated from source/General.f90:29-55:
 * ShowTime prints " yyyy year mm month dd day hh:mm:ss"; dScientificNotation: x_in = x_out * 10^i, 1 <= x_out < 10. */
void __general_MOD_showtime(void);
void general_mp_showtime_(void);
void __general_MOD_dscientificnotation(double *x, int *i);
void general_mp_dscientificnotation_(double *x, int *i);

#ifdef __cplusplus
}
#endif
#endif
