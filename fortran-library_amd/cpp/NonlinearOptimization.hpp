// FL::NO -- C++ front end of libFL.so (MI355X) for existing users of the reference's
// cpp/NonlinearOptimization.hpp: same namespace, function names, argument order and defaults
// (reference cpp/NonlinearOptimization.hpp:395-590), so `#include <FortranLibrary.hpp>` code
// recompiles unchanged.  The calls go to the mangled entry points of include/fl_legacy.h
// (host callbacks, solver arithmetic on the GPU).  New here: FL::NO::LBFGS (the reference's
// header never exposed it) and FL::NO::batched::* over device pointers.
#ifndef FL_AMD_NonlinearOptimization_hpp
#define FL_AMD_NonlinearOptimization_hpp

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fl_legacy.h"
#include "../../include/fl_nlopt.h"

namespace FL { namespace NO {

// callback shapes of the reference header: Fortran passes everything by reference
using f_t = void (*)(double &, const double *, const int &);
using fd_t = void (*)(double *, const double *, const int &);
using f_fd_t = int (*)(double &, double *, const double *, const int &);
using fdd_t = int (*)(double *, const double *, const int &);
using c_t = void (*)(double *, const double *, const int &, const int &);
using cd_t = void (*)(double *, const double *, const int &, const int &);
using cdd_t = int (*)(double *, const double *, const int &, const int &);

namespace detail {
// a Fortran logical travels as a 4-byte integer; the reference header sends -1 / 0 (cpp/README.md:15-18)
struct Common {
    int32_t strong, warning;
    int maxit;
    double precision, minstep, c1, c2, incr;
    Common(bool s, bool w, int it, double p, double ms, double a, double b, double inc)
        : strong(s ? -1 : 0), warning(w ? -1 : 0), maxit(it), precision(p), minstep(ms), c1(a), c2(b), incr(inc) {}
};
// references and pointers share a representation in every ABI libFL.so is built for; the C header spells the
// callbacks with pointers, the reference's C++ header with references
template <class To, class From> inline To as(From p) { return reinterpret_cast<To>(p); }
} // namespace detail

#define FL_NO_COMMON_ARGS                                                                                         \
    const bool &Strong = true, const bool &Warning = true, const int &MaxIteration = 1000,                        \
    const double &Precision = 1e-15, const double &MinStepLength = 1e-15, const double &WolfeConst1 = 1e-4
#define FL_NO_PASS(k) &k.strong, &k.warning, &k.maxit, &k.precision, &k.minstep, &k.c1, &k.c2, &k.incr

inline void SteepestDescent(f_t f, fd_t fd, f_fd_t f_fd, double *x, const int &dim, FL_NO_COMMON_ARGS,
                            const double &WolfeConst2 = 0.9, const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_steepestdescent(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim,
                                                detail::as<fl_f_fd_cb>(f_fd), FL_NO_PASS(k));
}

// without f_fd: the reference routes this overload to ConjugateGradient_basic (hpp:417-434)
inline void ConjugateGradient(f_t f, fd_t fd, double *x, const int &dim, const std::string &Method = "DY",
                              FL_NO_COMMON_ARGS, const double &WolfeConst2 = 0.45, const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_conjugategradient_basic(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim,
                                                        Method.c_str(), FL_NO_PASS(k), (int)Method.size());
}
inline void ConjugateGradient(f_t f, fd_t fd, f_fd_t f_fd, double *x, const int &dim,
                              const std::string &Method = "DY", FL_NO_COMMON_ARGS, const double &WolfeConst2 = 0.45,
                              const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_conjugategradient(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim,
                                                  Method.c_str(), detail::as<fl_f_fd_cb>(f_fd), FL_NO_PASS(k),
                                                  (int)Method.size());
}

// new: subroutine LBFGS(f,fd,x,dim,Memory,f_fd,...) NO.f90:398-400 (argument order follows BFGS below)
inline void LBFGS(f_t f, fd_t fd, f_fd_t f_fd, double *x, const int &dim, const int &Memory = 10, FL_NO_COMMON_ARGS,
                  const double &WolfeConst2 = 0.9, const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_lbfgs(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim, &Memory,
                                      detail::as<fl_f_fd_cb>(f_fd), FL_NO_PASS(k));
}

inline void BFGS(f_t f, fd_t fd, f_fd_t f_fd, fdd_t fdd, double *x, const int &dim, const int &ExactStep = 20,
                 FL_NO_COMMON_ARGS, const double &WolfeConst2 = 0.9, const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_bfgs(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim,
                                     detail::as<fl_fdd_cb>(fdd), &ExactStep, detail::as<fl_f_fd_cb>(f_fd),
                                     FL_NO_PASS(k));
}

inline void NewtonRaphson(f_t f, fd_t fd, f_fd_t f_fd, fdd_t fdd, double *x, const int &dim, FL_NO_COMMON_ARGS,
                          const double &WolfeConst2 = 0.9, const double &Increment = 1.05)
{
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_newtonraphson(detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), x, &dim,
                                              detail::as<fl_fdd_cb>(fdd), detail::as<fl_f_fd_cb>(f_fd), FL_NO_PASS(k));
}

// hpp:494-511: residue(f'(x), x, M, N), Jacobian(J(x), x, M, N); an own Levenberg-Marquardt iteration stands in for
// MKL's trnlsp (include/fl_legacy.h)
inline void TrustRegion(void (*residue)(double *, const double *, const int &, const int &),
                        void (*Jacobian)(double *, const double *, const int &, const int &), double *x, const int &M,
                        const int &N, const bool &Warning = true, const int &MaxIteration = 1000,
                        const int &MaxStepIteration = 100, const double &Precision = 1e-15,
                        const double &MinStepLength = 1e-15)
{
    const int32_t w = Warning ? -1 : 0;
    __nonlinearoptimization_MOD_trustregion_basic(detail::as<fl_residue_cb>(residue), detail::as<fl_jacobian_cb>(Jacobian), x,
                                                  &M, &N, &w, &MaxIteration, &MaxStepIteration, &Precision, &MinStepLength);
}

// lambda0 empty = zeros (hpp:575-578)
inline void AugmentedLagrangian(f_t f, fd_t fd, f_fd_t f_fd, fdd_t fdd, c_t c, cd_t cd, cdd_t cdd, double *x,
                                const int &N, const int &M, const std::string &UnconstrainedSolver = "BFGS",
                                std::vector<double> lambda0 = {}, const double &miu0 = 1.0, const int &ExactStep = 20,
                                const int &Memory = 10, const std::string &Method = "DY", FL_NO_COMMON_ARGS,
                                double WolfeConst2 = 0.9, const double &Increment = 1.05)
{
    if (lambda0.size() != (size_t)M) lambda0.assign((size_t)M, 0.0);
    detail::Common k(Strong, Warning, MaxIteration, Precision, MinStepLength, WolfeConst1, WolfeConst2, Increment);
    __nonlinearoptimization_MOD_augmentedlagrangian(
        detail::as<fl_f_cb>(f), detail::as<fl_fd_cb>(fd), detail::as<fl_c_cb>(c), detail::as<fl_cd_cb>(cd), x, &N, &M,
        UnconstrainedSolver.c_str(), lambda0.data(), &miu0, detail::as<fl_fdd_cb>(fdd), detail::as<fl_cdd_cb>(cdd),
        &ExactStep, &Memory, Method.c_str(), detail::as<fl_f_fd_cb>(f_fd), FL_NO_PASS(k),
        (int)UnconstrainedSolver.size(), (int)Method.size());
}

#undef FL_NO_COMMON_ARGS
#undef FL_NO_PASS

// Batches of independent problems with a built-in objective, everything resident on the device
// (include/fl_nlopt.h).  x_dev [batch][n] in/out; outputs may be null.  Returns FL_OK or an FL_ERR_* code.
namespace batched {
struct Results {
    double *f = nullptr, *gg = nullptr;
    int32_t *iters = nullptr, *status = nullptr, *nf = nullptr, *ng = nullptr;
};
inline fl_options Options(int solver)
{
    fl_options o;
    fl_default_options(&o, solver);
    return o;
}
inline int LBFGS(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                 const fl_options &opt, void *workspace_dev, size_t workspace_bytes, const Results &r = Results(),
                 void *stream = nullptr)
{
    return fl_lbfgs_batched(objective, batch, n, x_dev, d_dev, b_dev, &opt, workspace_dev, workspace_bytes, r.f,
                            r.gg, r.iters, r.status, r.nf, r.ng, stream);
}
inline int ConjugateGradient(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                             const fl_options &opt, const Results &r = Results(), void *stream = nullptr)
{
    return fl_conjugate_gradient_batched(objective, batch, n, x_dev, d_dev, b_dev, &opt, r.f, r.gg, r.iters,
                                         r.status, r.nf, r.ng, stream);
}
inline int SteepestDescent(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                           const fl_options &opt, const Results &r = Results(), void *stream = nullptr)
{
    return fl_steepest_descent_batched(objective, batch, n, x_dev, d_dev, b_dev, &opt, r.f, r.gg, r.iters, r.status,
                                       r.nf, r.ng, stream);
}
inline int BFGS(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                const fl_options &opt, void *workspace_dev, size_t workspace_bytes, const Results &r = Results(),
                void *stream = nullptr)
{
    return fl_bfgs_batched(objective, batch, n, x_dev, d_dev, b_dev, &opt, workspace_dev, workspace_bytes, r.f, r.gg,
                           r.iters, r.status, r.nf, r.ng, stream);
}
inline int NewtonRaphson(int objective, int batch, int n, double *x_dev, const double *d_dev, const double *b_dev,
                         const fl_options &opt, void *workspace_dev, size_t workspace_bytes,
                         const Results &r = Results(), void *stream = nullptr)
{
    return fl_newton_raphson_batched(objective, batch, n, x_dev, d_dev, b_dev, &opt, workspace_dev, workspace_bytes,
                                     r.f, r.gg, r.iters, r.status, r.nf, r.ng, stream);
}
} // namespace batched

}} // namespace FL::NO

#endif
