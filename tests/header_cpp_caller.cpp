// A user program written against FL::NO (fortran-library_amd/cpp/FortranLibrary.hpp): the calls a user of the
// reference's header makes (test/test.cpp:84-125: quartic sum x^4, dim 10, "close to 0"; unit-sphere constraint for
// the augmented Lagrangian), with default arguments, plus the new FL::NO::LBFGS and a numerical-Hessian Newton run.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../fortran-library_amd/cpp/FortranLibrary.hpp"

static void f(double &fx, const double *x, const int &dim)
{
    fx = 0.0;
    for (int i = 0; i < dim; i++) fx += x[i] * x[i] * x[i] * x[i];
}
static void fd(double *g, const double *x, const int &dim)
{
    for (int i = 0; i < dim; i++) g[i] = 4.0 * x[i] * x[i] * x[i];
}
static int f_fd(double &fx, double *g, const double *x, const int &dim)
{
    f(fx, x, dim);
    fd(g, x, dim);
    return 0;
}
static int fdd(double *H, const double *x, const int &dim)
{
    for (int i = 0; i < dim * dim; i++) H[i] = 0.0;
    for (int i = 0; i < dim; i++) H[i * dim + i] = 12.0 * x[i] * x[i];
    return 0;
}
static void c(double *cx, const double *x, const int &, const int &N)
{
    cx[0] = -1.0;
    for (int i = 0; i < N; i++) cx[0] += x[i] * x[i];
}
static void cd(double *cdx, const double *x, const int &, const int &N)
{
    for (int i = 0; i < N; i++) cdx[i] = 2.0 * x[i];
}

static double norm(const std::vector<double> &x)
{
    double s = 0.0;
    for (double v : x) s += v * v;
    return std::sqrt(s);
}
static std::vector<double> start(int dim)
{
    std::vector<double> x(dim);
    for (int i = 0; i < dim; i++) x[i] = 0.1 * (i + 1);
    return x;
}

int main()
{
    const int dim = 10;
    std::vector<double> x;
    x = start(dim);
    FL::NO::SteepestDescent(f, fd, f_fd, x.data(), dim, true, false, 200);
    std::printf("SD %.3e\n", norm(x));
    x = start(dim);
    FL::NO::ConjugateGradient(f, fd, x.data(), dim);
    std::printf("CG_basic %.3e\n", norm(x));
    x = start(dim);
    FL::NO::ConjugateGradient(f, fd, f_fd, x.data(), dim, "PR");
    std::printf("CG_PR %.3e\n", norm(x));
    x = start(dim);
    FL::NO::LBFGS(f, fd, f_fd, x.data(), dim);
    std::printf("LBFGS %.3e\n", norm(x));
    x = start(dim);
    FL::NO::BFGS(f, fd, f_fd, fdd, x.data(), dim);
    std::printf("BFGS %.3e\n", norm(x));
    x = start(dim);
    FL::NO::BFGS(f, fd, f_fd, nullptr, x.data(), dim, 20, true, false); // numerical Hessian
    std::printf("BFGS_numH %.3e\n", norm(x));
    x = start(dim);
    FL::NO::NewtonRaphson(f, fd, f_fd, fdd, x.data(), dim, true, false);
    std::printf("Newton %.3e\n", norm(x));
    x = start(dim);
    FL::NO::AugmentedLagrangian(f, fd, f_fd, nullptr, c, cd, nullptr, x.data(), dim, 1, "LBFGS", {}, 1.0, 20, 10, "DY",
                                true, false, 100, 1e-10);
    std::printf("AugLag_unit_sphere %.3e\n", std::fabs(norm(x) - 1.0));
    std::printf("Mission complete\n");
    return 0;
}
