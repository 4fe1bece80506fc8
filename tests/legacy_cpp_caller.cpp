// A C++ caller written the way users of the reference write one: it declares the mangled Fortran symbols with
// C++ reference parameters exactly like the reference's header does (cpp/NonlinearOptimization.hpp:278-393 --
// declarations restated here, the header itself is not copied) and calls them like FL::NO::* would: every
// optional passed, logicals as -1 / 0, the hidden string length last.  Linked against libFL.so (MI355X).
// Mirrors the optimiser calls of the reference's test/test.cpp:84-125 (quartic, dim 10, "close to 0").
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

extern "C" {
void __nonlinearoptimization_MOD_steepestdescent(void (*f)(double &, const double *, const int &),
                                                 void (*fd)(double *, const double *, const int &), double *x,
                                                 const int &dim, int (*f_fd)(double &, double *, const double *, const int &),
                                                 const int32_t &Strong, const int32_t &Warning, const int &MaxIteration,
                                                 const double &Precision, const double &MinStepLength,
                                                 const double &WolfeConst1, const double &WolfeConst2, const double &Increment);
void __nonlinearoptimization_MOD_conjugategradient_basic(void (*f)(double &, const double *, const int &),
                                                         void (*fd)(double *, const double *, const int &), double *x,
                                                         const int &dim, const char *Method, const int32_t &Strong,
                                                         const int32_t &Warning, const int &MaxIteration,
                                                         const double &Precision, const double &MinStepLength,
                                                         const double &WolfeConst1, const double &WolfeConst2,
                                                         const double &Increment, int len_Method);
void __nonlinearoptimization_MOD_conjugategradient(void (*f)(double &, const double *, const int &),
                                                   void (*fd)(double *, const double *, const int &), double *x,
                                                   const int &dim, const char *Method,
                                                   int (*f_fd)(double &, double *, const double *, const int &),
                                                   const int32_t &Strong, const int32_t &Warning, const int &MaxIteration,
                                                   const double &Precision, const double &MinStepLength,
                                                   const double &WolfeConst1, const double &WolfeConst2,
                                                   const double &Increment, int len_Method);
void __nonlinearoptimization_MOD_bfgs(void (*f)(double &, const double *, const int &),
                                      void (*fd)(double *, const double *, const int &), double *x, const int &dim,
                                      int (*fdd)(double *, const double *, const int &), const int &ExactStep,
                                      int (*f_fd)(double &, double *, const double *, const int &), const int32_t &Strong,
                                      const int32_t &Warning, const int &MaxIteration, const double &Precision,
                                      const double &MinStepLength, const double &WolfeConst1, const double &WolfeConst2,
                                      const double &Increment);
void __nonlinearoptimization_MOD_augmentedlagrangian(
    void (*f)(double &, const double *, const int &), void (*fd)(double *, const double *, const int &),
    void (*c)(double *, const double *, const int &, const int &), void (*cd)(double *, const double *, const int &, const int &),
    double *x, const int &N, const int &M, const char *UnconstrainedSolver, const double *lambda0, const double &miu0,
    int (*fdd)(double *, const double *, const int &), int (*cdd)(double *, const double *, const int &, const int &),
    const int &ExactStep, const int &Memory, const char *Method, int (*f_fd)(double &, double *, const double *, const int &),
    const int32_t &Strong, const int32_t &Warning, const int &MaxIteration, const double &Precision,
    const double &MinStepLength, const double &WolfeConst1, const double &WolfeConst2, const double &Increment,
    int len_UnconstrainedSolver, int len_Method);
}

static void f(double &fx, const double *x, const int &dim)
{
    fx = 0.0;
    for (int i = 0; i < dim; i++) fx += x[i] * x[i] * x[i] * x[i];
}
static void fd(double *fdx, const double *x, const int &dim)
{
    for (int i = 0; i < dim; i++) fdx[i] = 4.0 * x[i] * x[i] * x[i];
}
static int f_fd(double &fx, double *fdx, const double *x, const int &dim)
{
    f(fx, x, dim);
    fd(fdx, x, dim);
    return 0;
}
static void constraint(double *cx, const double *x, const int &, const int &N)
{
    cx[0] = -1.0;
    for (int i = 0; i < N; i++) cx[0] += x[i] * x[i];
}
static void constraintd(double *cdx, const double *x, const int &, const int &N)
{
    for (int i = 0; i < N; i++) cdx[i] = 2.0 * x[i];
}
static double norm(const double *x, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return std::sqrt(s);
}

int main()
{
    const int dim = 10;
    double x[dim];
    auto start = [&]() { for (int i = 0; i < dim; i++) x[i] = 0.1 * (i + 1); };
    const int32_t T = -1, F = 0;
    start();
    __nonlinearoptimization_MOD_steepestdescent(f, fd, x, dim, nullptr, T, F, 300, 1e-15, 1e-15, 1e-4, 0.9, 1.05);
    std::printf("SD %.16e\n", norm(x, dim));
    start();
    __nonlinearoptimization_MOD_conjugategradient_basic(f, fd, x, dim, "DY", T, F, 1000, 1e-15, 1e-15, 1e-4, 0.45, 1.05, 2);
    std::printf("CG-DY-basic %.16e\n", norm(x, dim));
    start();
    __nonlinearoptimization_MOD_conjugategradient(f, fd, x, dim, "PR", f_fd, T, F, 1000, 1e-15, 1e-15, 1e-4, 0.45, 1.05, 2);
    std::printf("CG-PR-f_fd %.16e\n", norm(x, dim));
    start();
    __nonlinearoptimization_MOD_bfgs(f, fd, x, dim, nullptr, 0, f_fd, T, F, 1000, 1e-15, 1e-15, 1e-4, 0.9, 1.05);
    std::printf("BFGS0 %.16e\n", norm(x, dim));
    start();
    const std::string solver = "LBFGS", method = "DY";
    const double lambda0[1] = {0.0};
    __nonlinearoptimization_MOD_augmentedlagrangian(f, fd, constraint, constraintd, x, dim, 1, solver.c_str(), lambda0, 1.0,
                                                    nullptr, nullptr, 0, 10, method.c_str(), nullptr, T, F, 1000, 1e-8,
                                                    1e-15, 1e-4, 0.9, 1.05, (int)solver.size(), (int)method.size());
    std::printf("AugLag-LBFGS-norm-minus-1 %.16e\n", std::fabs(norm(x, dim) - 1.0));
    std::printf("Mission complete\n");
    return 0;
}
