// fl_linalg.cpp -- the LinearAlgebra entry points the reference's C++ header binds next to the optimisers
// (cpp/FortranLibrary.hpp:48-63): My_dgemm / My_dgemm_T (LinearAlgebra.f90:182-196: dgemm calls) and My_dsyev
// (879-887: dsyev 'L') -- host arrays in, host arrays out, like the reference -- and the TrustRegion replacement.
// The reference hands the three to MKL; here they run on kernels of this library (csrc/fl_blas_kernels.hip: an f64-MFMA
// DGEMM and a cyclic Jacobi eigensolver).  No vendor BLAS / LAPACK is opened.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"

namespace {

struct DeviceBuffer {
    void *p = nullptr;
    explicit DeviceBuffer(size_t bytes) { (void)hipMalloc(&p, bytes ? bytes : 8); }
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// C(M,N) = op(A) B, column-major; transA: A is K x M
void gemm(bool transA, const double *A, const double *B, double *C, int M, int K, int N)
{
    int ndev = 0;
    if (M <= 0 || N <= 0 || K <= 0) return;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "FortranLibrary(MI355X) My_dgemm: no HIP device; C is unchanged\n");
        return;
    }
    const size_t a = sizeof(double) * (size_t)M * K, b = sizeof(double) * (size_t)K * N, c = sizeof(double) * (size_t)M * N;
    DeviceBuffer Ad(a), Bd(b), Cd(c);
    bool ok = Ad.p && Bd.p && Cd.p && hipMemcpy(Ad.p, A, a, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(Bd.p, B, b, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && fl_dgemm(transA ? 1 : 0, M, K, N, Ad.as<double>(), transA ? K : M, Bd.as<double>(), K, Cd.as<double>(), M,
                        nullptr) == FL_OK;
    ok = ok && hipMemcpy(C, Cd.p, c, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) My_dgemm: device error; C is undefined\n");
}

} // namespace

extern "C" {

// The library's replacement of MKL's djacobi (the reference: NO.f90:676, 981, 1067, 1258, 1779, 1833), with MKL's own
// argument convention -- fcn(m, n, x, f) by reference, fjac(m, n) column-major, the value TR_SUCCESS = 1501 returned --
// and MKL's step rule (fl_host.hpp).  Host code: no device is needed.
int fl_djacobi(void (*fcn)(const int *, const int *, const double *, double *), const int *n, const int *m, double *fjac,
               double *x, const double *eps)
{
    if (!fcn || !n || !m || !fjac || !x || !eps || *n <= 0 || *m <= 0 || !(*eps > 0.0)) return 1502; // TR_INVALID_OPTION
    std::vector<double> fp(*m), fm(*m);
    fl::central_difference_jacobian([&](const double *p, double *f) { fcn(m, n, p, f); }, *n, *m, fjac, x, *eps, fp.data(),
                                    fm.data());
    return 1501;
}

// subroutine My_dgemm(A,B,C,M,K,N): C = A . B (LinearAlgebra.f90:182-188)
void __linearalgebra_MOD_my_dgemm(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(false, A, B, C, *M, *K, *N);
}
// subroutine My_dgemm_T(A,B,C,M,K,N): C = A^T . B, A is K x M (LinearAlgebra.f90:190-196; FortranLibrary.hpp:50)
void __linearalgebra_MOD_my_dgemm_t(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(true, A, B, C, *M, *K, *N);
}
// subroutine My_dsyev(jobtype,A,eigval,N): eigenvalues ascending, A <- normalised eigenvectors for 'V'
// (LinearAlgebra.f90:879-887: dsyev(jobtype,'L',...); "A will be overwritten even for 'N' job")
void __linearalgebra_MOD_my_dsyev(const char *jobtype, double *A, double *eigval, const int *N, int /*len_jobtype*/)
{
    const int n = *N;
    int ndev = 0;
    if (n <= 0) return;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::fprintf(stderr, "FortranLibrary(MI355X) My_dsyev: no HIP device; A is unchanged\n");
        return;
    }
    const bool vec = jobtype && (*jobtype == 'V' || *jobtype == 'v');
    const size_t a = sizeof(double) * (size_t)n * n;
    // with eigenvectors: tridiagonalisation + inverse iteration + Cholesky-QR + back-transformation (fl_dsyev_vectors),
    // whose result is checked on the device; cyclic Jacobi if that check fails (or beyond its size, or on request:
    // FL_DSYEV_JACOBI=1 in the environment)
    const char *force = std::getenv("FL_DSYEV_JACOBI");
    bool fast_vec = vec && n <= 6144 && !(force && force[0] == '1');
    const size_t wsb = std::max(fl_dsyev_workspace_bytes(n), fast_vec ? fl_dsyev_vectors_workspace_bytes(n) : (size_t)0);
    const char *dbg = std::getenv("FL_DSYEV_DEBUG");
    const bool debug = dbg && dbg[0] == '1';
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    DeviceBuffer Ad(a), Wd(sizeof(double) * n), ws(wsb);
    const double t_alloc = now();
    int sweeps = 0;
    // dsyev scales a matrix whose norm is close to the under/overflow thresholds before it squares anything (dlascl);
    // here: by a power of two (exact) to max |a_ij| ~ 1 when that maximum is outside [1e-100, 1e100]
    double amax = 0.0;
    int scale_exp = 0; // the matrix is solved as A * 2^scale_exp (scalbn: exact, and no intermediate 2^e that could overflow -- a matrix of
                       // subnormals, max |a_ij| < 2^-1022, has e > 1023)
    for (int j = 0; j < n; ++j)
        for (int i = j; i < n; ++i) amax = std::max(amax, std::fabs(A[(size_t)j * n + i])); // 'L': the lower triangle
    std::vector<double> scaled;
    const double *src = A;
    if (std::isfinite(amax) && amax > 0.0 && (amax < 1e-100 || amax > 1e100)) {
        scale_exp = -std::ilogb(amax);
        scaled.resize((size_t)n * n);
        for (size_t k = 0; k < (size_t)n * n; ++k) scaled[k] = std::scalbn(A[k], scale_exp);
        src = scaled.data();
    }
    bool ok = Ad.p && Wd.p && ws.p && hipMemcpy(Ad.p, src, a, hipMemcpyHostToDevice) == hipSuccess;
    bool vectors_in_A = false;
    const double t_up = now();
    if (ok && fast_vec) {
        const int rc = fl_dsyev_vectors(n, Ad.as<double>(), n, Wd.as<double>(), ws.p, wsb, nullptr, nullptr);
        if (rc == FL_OK) vectors_in_A = true;
        else if (rc == 1) { // the basis did not pass its check: Jacobi on a fresh copy
            fast_vec = false;
            ok = hipMemcpy(Ad.p, src, a, hipMemcpyHostToDevice) == hipSuccess;
        } else ok = false;
    }
    // eigenvalues only: tridiagonalisation + multisection (ascending already); beyond its size: Jacobi
    const bool values_only = !vec && n <= 6144;
    if (vectors_in_A) {
    } else if (values_only) ok = ok && fl_dsyev_values(n, Ad.as<double>(), n, Wd.as<double>(), ws.p, wsb, nullptr) == FL_OK;
    else ok = ok && fl_dsyev_jacobi(vec ? 'V' : 'N', n, Ad.as<double>(), n, Wd.as<double>(), ws.p, wsb, 60, &sweeps, nullptr) == FL_OK;
    const double t_solved = now();
    std::vector<double> w(n), V;
    ok = ok && hipMemcpy(w.data(), Wd.p, sizeof(double) * n, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && vectors_in_A) // ascending already, the vectors in place: straight into the caller's array
        ok = hipMemcpy(A, Ad.p, a, hipMemcpyDeviceToHost) == hipSuccess;
    else if (ok && vec) {
        V.resize((size_t)n * n);
        ok = hipMemcpy(V.data(), ws.as<double>() + (size_t)n * n, a, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok) {
        std::fprintf(stderr, "FortranLibrary(MI355X) My_dsyev: device error; A, eigval are undefined\n");
        return;
    }
    if (sweeps < 0) std::fprintf(stderr, "FortranLibrary(MI355X) My_dsyev: Jacobi sweeps exhausted before convergence\n");
    if (debug)
        std::fprintf(stderr, "My_dsyev n=%d: alloc %.2f ms, scan + upload %.2f ms, device %.2f ms, download %.2f ms\n", n, t_alloc - t_begin,
                     t_up - t_alloc, t_solved - t_up, now() - t_solved);
    // eigenvalues in ascending order, eigenvectors follow (dsyev's contract, LinearAlgebra.f90:877)
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return w[x] < w[y]; });
    for (int k = 0; k < n; ++k) eigval[k] = scale_exp ? std::scalbn(w[order[k]], -scale_exp) : w[order[k]];
    if (vec && !vectors_in_A)
        for (int k = 0; k < n; ++k) std::copy(V.begin() + (size_t)order[k] * n, V.begin() + (size_t)(order[k] + 1) * n, A + (size_t)k * n);
}
// the ifort manglings (FortranLibrary.hpp:27-43)
void linearalgebra_mp_my_dgemm_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(false, A, B, C, *M, *K, *N);
}
void linearalgebra_mp_my_dgemm_t_(const double *A, const double *B, double *C, const int *M, const int *K, const int *N)
{
    gemm(true, A, B, C, *M, *K, *N);
}
void linearalgebra_mp_my_dsyev_(const char *jobtype, double *A, double *eigval, const int *N, int len)
{
    __linearalgebra_MOD_my_dsyev(jobtype, A, eigval, N, len);
}

} // extern "C"

// ------------------------------------------------------------------ TrustRegion
// subroutine TrustRegion(fd,x,M,N,Jacobian,low,up,Warning,MaxIteration,MaxStepIteration,Precision,MinStepLength)
// and TrustRegion_basic (NO.f90:1728-1906, 2348-2423; hpp:358-366): solve f'(x) = 0 by minimising |f'(x)|^2.
// The reference is a wrapper of MKL's closed RCI solver dtrnlsp -- there is no algorithm in the reference to restate
// (SURVEY.md 8f.4), so this is an own Levenberg-Marquardt iteration with Nielsen's damping update behind the same
// interface and stopping options: own path, end points held to the real dtrnlsp's (tests/test_mkl_pins.py).
// Callbacks on the host; J^T J, J^T f' through fl_dgemm and the damped normal equations through fl_dposv_batched on
// the GPU.  Bounds (low, up) are honoured by projecting every trial point.
typedef void (*res_cb)(double *, const double *, const int &, const int &);
typedef int (*jac_cb)(double *, const double *, const int &, const int &);

static void trust_region(res_cb fd, jac_cb Jacobian, double *x, int M, int N, const double *low, const double *up, int warn,
                         int maxit, int maxstepit, double precision, double minstep)
{
    int threads = 0, ept = 0;
    if (N <= 0 || M < N || N > 4096 || fl_reduction_geometry(N, &threads, &ept) != FL_OK) {
        std::fprintf(stderr, "FortranLibrary(MI355X) TrustRegion: needs 0 < N <= M, N <= 4096; x is unchanged\n");
        return;
    }
    const size_t ld = (size_t)threads * ept;
    std::vector<double> r(M), rn(M), J((size_t)M * N), A((size_t)N * N), g(N), d(N), xn(N), Ap((size_t)N * ld), rp(M), rm(M);
    double *Ad = nullptr, *bd = nullptr;
    int32_t *infod = nullptr;
    bool ok = hipMalloc((void **)&Ad, sizeof(double) * N * ld) == hipSuccess &&
              hipMalloc((void **)&bd, sizeof(double) * N) == hipSuccess &&
              hipMalloc((void **)&infod, sizeof(int32_t)) == hipSuccess;
    auto project = [&](double *v) {
        if (low && up)
            for (int i = 0; i < N; ++i) v[i] = v[i] < low[i] ? low[i] : (v[i] > up[i] ? up[i] : v[i]);
    };
    auto sq = [](const std::vector<double> &v) {
        double t = 0.0;
        for (double e : v) t += e * e;
        return t;
    };
    auto jacobian = [&](const double *xx) { // analytical, or central differences of f' (the reference: MKL djacobi)
        if (Jacobian) {
            (void)Jacobian(J.data(), xx, M, N);
            return;
        }
        std::vector<double> xp(xx, xx + N); // djacobi(fd_j,N,M,J,x,1d-8), NO.f90:1779: MKL's step rule (fl_host.hpp)
        fl::central_difference_jacobian([&](const double *p, double *f) { fd(f, p, M, N); }, N, M, J.data(), xp.data(), 1e-8,
                                        rp.data(), rm.data());
    };
    auto normal_equations = [&]() { // A = J^T J, g = J^T r on the GPU
        gemm(true, J.data(), J.data(), A.data(), N, M, N);
        gemm(true, J.data(), r.data(), g.data(), N, M, 1);
    };
    project(x);
    fd(r.data(), x, M, N);
    double f2 = sq(r);
    const double f2_0 = f2;
    jacobian(x);
    normal_equations();
    double mu = 0.0;
    for (int i = 0; i < N; ++i) mu = std::fmax(mu, A[(size_t)i * N + i]);
    mu = (mu > 0.0 ? mu : 1.0) * 1e-3;
    double nu = 2.0;
    int it = 0, reason = 1;
    bool stop = false;
    for (it = 1; ok && !stop && it <= maxit; ++it) {
        if (std::sqrt(f2) < precision) { reason = 3; break; } // || f'(x) ||_2 < Precision
        bool accepted = false;
        for (int step = 0; ok && step < maxstepit && !accepted; ++step) {
            std::fill(Ap.begin(), Ap.end(), 0.0);
            for (int j = 0; j < N; ++j)
                for (int i = 0; i < N; ++i) Ap[(size_t)j * ld + i] = A[(size_t)j * N + i] + (i == j ? mu : 0.0);
            for (int i = 0; i < N; ++i) d[i] = -g[i];
            int32_t info = 0;
            ok = hipMemcpy(Ad, Ap.data(), sizeof(double) * N * ld, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(bd, d.data(), sizeof(double) * N, hipMemcpyHostToDevice) == hipSuccess &&
                 fl_dposv_batched(1, N, Ad, bd, infod, nullptr) == FL_OK &&
                 hipMemcpy(&info, infod, sizeof info, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(d.data(), bd, sizeof(double) * N, hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) break;
            if (info == 0) {
                for (int i = 0; i < N; ++i) xn[i] = x[i] + d[i];
                project(xn.data());
                double s2 = 0.0, pred = 0.0;
                for (int i = 0; i < N; ++i) {
                    const double di = xn[i] - x[i];
                    s2 += di * di;
                    pred += di * (mu * di - g[i]);
                }
                if (std::sqrt(s2) < minstep) { reason = 5; stop = true; break; } // || s ||_2 < MinStepLength
                fd(rn.data(), xn.data(), M, N);
                const double f2n = sq(rn), rho = (f2 - f2n) / (pred > 0.0 ? pred : 1e-300);
                if (f2n < f2 && rho > 0.0) {
                    for (int i = 0; i < N; ++i) x[i] = xn[i];
                    r.swap(rn);
                    f2 = f2n;
                    const double t = 2.0 * rho - 1.0;
                    mu = mu * std::fmax(1.0 / 3.0, 1.0 - t * t * t);
                    nu = 2.0;
                    accepted = true;
                    continue;
                }
            }
            mu = mu * nu; // rejected (or not positive definite): shrink the trust region
            nu = 2.0 * nu;
        }
        if (stop) break;
        if (!accepted) { reason = 2; break; } // no acceptable step within MaxStepIteration: radius exhausted
        jacobian(x);
        normal_equations();
        double gmax = 0.0;
        for (int i = 0; i < N; ++i) gmax = std::fmax(gmax, std::fabs(g[i]));
        if (gmax == 0.0) { reason = 4; break; } // stationary point of the merit function
    }
    if (!ok) std::fprintf(stderr, "FortranLibrary(MI355X) TrustRegion: HIP error\n");
    if (ok && warn && reason != 3) { // the reference's report (NO.f90:1880-1904), reasons renumbered for this solver
        if (reason == 1) std::printf(" Failed trust region: max iteration exceeded!\n");
        else std::printf(" Trust region warning: stopped before || f'(x) ||_2 met the precision (reason %d)\n", reason);
        std::printf(" Final residual = %24.16E   (initial %24.16E)\n", std::sqrt(f2), std::sqrt(f2_0));
    }
    if (Ad) (void)hipFree(Ad);
    if (bd) (void)hipFree(bd);
    if (infod) (void)hipFree(infod);
}

extern "C" {

void __nonlinearoptimization_MOD_trustregion_basic(res_cb fd, jac_cb Jacobian, double *x, const int *M, const int *N,
                                                   const int32_t *Warning, const int *MaxIteration,
                                                   const int *MaxStepIteration, const double *Precision,
                                                   const double *MinStepLength)
{
    trust_region(fd, Jacobian, x, *M, *N, nullptr, nullptr, *Warning != 0, *MaxIteration, *MaxStepIteration, *Precision,
                 *MinStepLength);
}
void nonlinearoptimization_mp_trustregion_basic_(res_cb fd, jac_cb Jacobian, double *x, const int *M, const int *N,
                                                 const int32_t *Warning, const int *MaxIteration,
                                                 const int *MaxStepIteration, const double *Precision,
                                                 const double *MinStepLength)
{
    __nonlinearoptimization_MOD_trustregion_basic(fd, Jacobian, x, M, N, Warning, MaxIteration, MaxStepIteration, Precision,
                                                  MinStepLength);
}
// the general routine: every argument after N is optional (NULL = absent)
void __nonlinearoptimization_MOD_trustregion(res_cb fd, double *x, const int *M, const int *N, jac_cb Jacobian,
                                             const double *low, const double *up, const int32_t *Warning,
                                             const int *MaxIteration, const int *MaxStepIteration, const double *Precision,
                                             const double *MinStepLength)
{
    trust_region(fd, Jacobian, x, *M, *N, (low && up) ? low : nullptr, (low && up) ? up : nullptr,
                 Warning ? *Warning != 0 : 1, MaxIteration ? *MaxIteration : 1000, MaxStepIteration ? *MaxStepIteration : 100,
                 Precision ? *Precision : 1e-15, MinStepLength ? *MinStepLength : 1e-15);
}
void nonlinearoptimization_mp_trustregion_(res_cb fd, double *x, const int *M, const int *N, jac_cb Jacobian,
                                           const double *low, const double *up, const int32_t *Warning,
                                           const int *MaxIteration, const int *MaxStepIteration, const double *Precision,
                                           const double *MinStepLength)
{
    __nonlinearoptimization_MOD_trustregion(fd, x, M, N, Jacobian, low, up, Warning, MaxIteration, MaxStepIteration, Precision,
                                            MinStepLength);
}

} // extern "C"
