// fl_trust_region.hip -- TrustRegion (NO.f90:1728-1906: solve f'(x) = 0 in the least-squares sense, M equations, N
// unknowns, optional box low <= x <= up) for a BATCH of independent problems, on the device, by reverse communication.
//
// The reference is a wrapper of MKL's closed RCI solver dtrnlsp/dtrnlspbc -- there is no algorithm in the reference to
// restate -- so this is the library's own Levenberg-Marquardt iteration with Nielsen's damping update,
// the same one the one-problem legacy symbol runs on the host (csrc/fl_linalg.cpp), behind the reference's stopping
// options (MaxIteration, MaxStepIteration, Precision on ||f'(x)||_2, MinStepLength on ||s||_2): own path, END POINTS held to
// the real MKL dtrnlsp's (tests/golden/mkl_trnlsp.npz, tests/test_mkl_pins.py).  Per step and problem: A = J^T J, g = J^T r on the f64 matrix cores
// (fl_dgemm_strided), (A + mu I) d = -g by fl_dposv_batched, trial point x + d projected into the box; gain ratio
// rho = (|r|^2 - |r_new|^2) / (d.(mu d - g)) decides: accept (mu *= max(1/3, 1 - (2 rho - 1)^3), new Jacobian wanted)
// or reject (mu *= nu, nu *= 2, new trial from the same A, g: they are kept per problem and renewed only by a Jacobian
// the problem asked for).  The caller evaluates residuals and Jacobians for the
// whole batch wherever the request bits ask: FL_TRS_REQ_R (1) residual at x_dev[k], FL_TRS_REQ_J (2) Jacobian at
// x_dev[k], FL_TRS_REQ_AGAIN (4) nothing to evaluate for this problem, call again; 0 finished.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>

#include "../../include/fl_nlopt.h"
#include "fl_host.hpp"

namespace fl {

enum { TP_INIT = 0, TP_TRIAL = 1, TP_JAC = 2, TP_DONE = 3, TP_SOLVE = 4 };
// stopping reasons as csrc/fl_linalg.cpp reports them: 1 max iteration, 2 no acceptable step within MaxStepIteration,
// 3 ||f'(x)|| < Precision, 4 stationary point of the merit function, 5 ||s|| < MinStepLength
struct TrsState {
    double mu, nu, f2, f2_0, pred;
    int phase, it, stepit, reason, have_mu;
    int fresh; // this step consumed a Jacobian the problem had asked for (TP_INIT, TP_JAC): A = J^T J, g = J^T r are renewed
};

__device__ __forceinline__ double block_sum(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ double block_max(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, red[w]);
    return t;
}

// what arrived: residual at the trial point (TP_TRIAL), residual + Jacobian at x0 (TP_INIT), Jacobian (TP_JAC)
__global__ __launch_bounds__(256) void trs_take_kernel(int M, int N, int maxit, int maxstepit, double precision, TrsState *S,
                                                       double *xcur, const double *x_io, double *r, const double *r_in)
{
    __shared__ double red[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    TrsState st = S[k];
    if (st.phase == TP_DONE || st.phase == TP_SOLVE) return;
    const double *rin = r_in + (size_t)k * M;
    double *rk = r + (size_t)k * M;
    if (st.phase == TP_INIT) {
        double a = 0.0;
        for (int i = tid; i < M; i += 256) {
            rk[i] = rin[i];
            a += rin[i] * rin[i];
        }
        const double f2 = block_sum(a, red);
        if (tid == 0) {
            st.f2 = st.f2_0 = f2;
            st.phase = (sqrt(f2) < precision) ? TP_DONE : TP_SOLVE;
            st.reason = (st.phase == TP_DONE) ? 3 : 1;
            st.fresh = 1;
            S[k] = st;
        }
    } else if (st.phase == TP_TRIAL) {
        double a = 0.0;
        for (int i = tid; i < M; i += 256) a += rin[i] * rin[i];
        const double f2n = block_sum(a, red);
        const double rho = (st.f2 - f2n) / (st.pred > 0.0 ? st.pred : 1e-300);
        const bool accept = f2n < st.f2 && rho > 0.0; // uniform: every thread holds the same sums
        if (accept) {
            for (int i = tid; i < M; i += 256) rk[i] = rin[i];
            for (int i = tid; i < N; i += 256) xcur[(size_t)k * N + i] = x_io[(size_t)k * N + i];
        }
        if (tid == 0) {
            if (accept) {
                st.f2 = f2n;
                const double t = 2.0 * rho - 1.0;
                st.mu = st.mu * fmax(1.0 / 3.0, 1.0 - t * t * t);
                st.nu = 2.0;
                st.stepit = 0;
                ++st.it;
                if (sqrt(f2n) < precision) { st.phase = TP_DONE; st.reason = 3; }
                else if (st.it >= maxit) { st.phase = TP_DONE; st.reason = 1; }
                else st.phase = TP_JAC; // the Jacobian at the accepted point is wanted next
            } else {
                st.mu = st.mu * st.nu; // rejected: shrink the trust region
                st.nu = 2.0 * st.nu;
                ++st.stepit;
                if (st.stepit >= maxstepit) { st.phase = TP_DONE; st.reason = 2; }
                else st.phase = TP_SOLVE;
            }
            S[k] = st;
        }
    } else { // TP_JAC: the caller has written the Jacobian at xcur
        if (tid == 0) {
            st.phase = TP_SOLVE;
            st.fresh = 1;
            S[k] = st;
        }
    }
}

// A <- J^T J, g <- J^T r of this step -- only for the problems whose step consumed a Jacobian they had asked for.  A
// problem that retries with more damping (rejected trial, Cholesky failure) keeps the A, g of its current point: what
// the caller's J array holds for it meanwhile (nothing was requested) does not matter.
__global__ __launch_bounds__(256) void trs_commit_kernel(int N, TrsState *S, const double *A_new, const double *g_new, double *A,
                                                         double *g)
{
    const int k = blockIdx.x, tid = threadIdx.x;
    if (!S[k].fresh) return;
    const size_t nn = (size_t)N * N;
    for (size_t e = tid; e < nn; e += 256) A[(size_t)k * nn + e] = A_new[(size_t)k * nn + e];
    for (int i = tid; i < N; i += 256) g[(size_t)k * N + i] = g_new[(size_t)k * N + i];
    __syncthreads();
    if (tid == 0) S[k].fresh = 0;
}

// Ap = A + mu I in the padded layout of fl_dposv_batched, rhs = -g (identity / zero for problems that do not solve)
__global__ __launch_bounds__(256) void trs_build_kernel(int N, int ld, TrsState *S, const double *A, const double *g, double *Ap,
                                                        double *rhs)
{
    __shared__ double red[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    TrsState st = S[k];
    const bool solve = st.phase == TP_SOLVE;
    const double *Ak = A + (size_t)k * N * N, *gk = g + (size_t)k * N;
    if (solve && !st.have_mu) { // first damping from the scale of J^T J (Nielsen): 1e-3 max diag
        double m = 0.0;
        for (int i = tid; i < N; i += 256) m = fmax(m, Ak[(size_t)i * N + i]);
        m = block_max(m, red);
        st.mu = (m > 0.0 ? m : 1.0) * 1e-3;
        st.nu = 2.0;
        st.have_mu = 1;
        if (tid == 0) S[k] = st;
    }
    if (solve) { // stationary point of the merit function: g = 0
        double m = 0.0;
        for (int i = tid; i < N; i += 256) m = fmax(m, fabs(gk[i]));
        m = block_max(m, red);
        if (m == 0.0) {
            if (tid == 0) {
                st.phase = TP_DONE;
                st.reason = 4;
                S[k] = st;
            }
            for (int i = tid; i < N; i += 256) rhs[(size_t)k * N + i] = 0.0;
            for (size_t e = tid; e < (size_t)N * ld; e += 256) Ap[(size_t)k * N * ld + e] = ((int)(e / ld) == (int)(e % ld)) ? 1.0 : 0.0;
            return;
        }
    }
    for (size_t e = tid; e < (size_t)N * ld; e += 256) {
        const int c = (int)(e / ld), rr = (int)(e % ld);
        double v = 0.0;
        if (rr < N) v = solve ? Ak[(size_t)c * N + rr] + (rr == c ? st.mu : 0.0) : (rr == c ? 1.0 : 0.0);
        Ap[(size_t)k * N * ld + e] = v;
    }
    for (int i = tid; i < N; i += 256) rhs[(size_t)k * N + i] = solve ? -gk[i] : 0.0;
}

// the trial point (or the verdict that none is needed) and the request for the caller
__global__ __launch_bounds__(256) void trs_trial_kernel(int N, int maxstepit, double minstep, TrsState *S, const double *xcur,
                                                        const double *d, const double *g, const int32_t *info,
                                                        const double *low, const double *up, double *x_io, int32_t *request)
{
    __shared__ double red[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    TrsState st = S[k];
    const double *xc = xcur + (size_t)k * N, *dk = d + (size_t)k * N, *gk = g + (size_t)k * N;
    double *xo = x_io + (size_t)k * N;
    int rq = 0;
    if (st.phase == TP_SOLVE) {
        if (info[k] != 0) { // A + mu I not positive definite: more damping, try again
            st.mu = st.mu * st.nu;
            st.nu = 2.0 * st.nu;
            ++st.stepit;
            if (st.stepit >= maxstepit) { st.phase = TP_DONE; st.reason = 2; }
            else rq = 4;
        } else {
            double s2 = 0.0, pred = 0.0;
            for (int i = tid; i < N; i += 256) {
                double xn = xc[i] + dk[i];
                if (low && up) xn = xn < low[i] ? low[i] : (xn > up[i] ? up[i] : xn);
                const double di = xn - xc[i];
                xo[i] = xn;
                s2 += di * di;
                pred += di * (st.mu * di - gk[i]);
            }
            s2 = block_sum(s2, red);
            pred = block_sum(pred, red);
            if (sqrt(s2) < minstep) { st.phase = TP_DONE; st.reason = 5; }
            else {
                st.pred = pred;
                st.phase = TP_TRIAL;
                rq = 1;
            }
        }
    } else if (st.phase == TP_JAC) {
        for (int i = tid; i < N; i += 256) xo[i] = xc[i]; // the accepted point (it is what x_io holds already)
        rq = 2;
    } else if (st.phase == TP_INIT) {
        rq = 1 | 2;
    }
    if (st.phase == TP_DONE) {
        __syncthreads();
        for (int i = tid; i < N; i += 256) xo[i] = xc[i];
        rq = 0;
    }
    if (tid == 0) {
        S[k] = st;
        request[k] = rq;
    }
}
__global__ void trs_init_kernel(int batch, TrsState *S)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= batch) return;
    TrsState st;
    st.mu = st.pred = st.f2 = st.f2_0 = 0.0;
    st.nu = 2.0;
    st.phase = TP_INIT;
    st.it = st.stepit = st.have_mu = st.fresh = 0;
    st.reason = 1;
    S[k] = st;
}
__global__ void trs_project_kernel(int batch, int N, const double *low, const double *up, double *xcur, double *x_io)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)batch * N) return;
    const int i = (int)(e % N);
    double v = x_io[e];
    if (low && up) v = v < low[i] ? low[i] : (v > up[i] ? up[i] : v);
    xcur[e] = v;
    x_io[e] = v;
}
__global__ void trs_results_kernel(int batch, const TrsState *S, double *resnorm, int32_t *iters, int32_t *reason)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= batch) return;
    if (resnorm) resnorm[k] = sqrt(S[k].f2);
    if (iters) iters[k] = S[k].it;
    if (reason) reason[k] = S[k].reason;
}

} // namespace fl

struct fl_trs {
    int batch, M, N, ld, maxit, maxstepit, first;
    double precision, minstep;
    fl::TrsState *S;
    double *xcur, *r, *A, *g, *A_new, *g_new, *Ap, *d, *low, *up;
    int32_t *info;
    hipStream_t st;
};

extern "C" {

int fl_trust_region_destroy(fl_trs *h)
{
    if (!h) return FL_OK;
    void *bufs[] = {h->S, h->xcur, h->r, h->A, h->g, h->A_new, h->g_new, h->Ap, h->d, h->low, h->up, h->info};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete h;
    return FL_OK;
}

int fl_trust_region_create(fl_trs **out, int batch, int M, int N, const double *low_dev, const double *up_dev,
                           int max_iteration, int max_step_iteration, double precision, double min_step_length, void *stream)
{
    if (!out || batch <= 0 || N <= 0 || M < N) return FL_ERR_INVALID_ARGUMENT;
    int threads = 0, ept = 0;
    if (fl_reduction_geometry(N, &threads, &ept) != FL_OK) return FL_ERR_UNSUPPORTED_SIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FL_ERR_NO_DEVICE;
    fl_trs *h = new (std::nothrow) fl_trs();
    if (!h) return FL_ERR_WORKSPACE;
    *h = fl_trs{};
    h->batch = batch; h->M = M; h->N = N; h->ld = threads * ept;
    h->maxit = max_iteration; h->maxstepit = max_step_iteration;
    h->precision = precision; h->minstep = min_step_length;
    h->first = 1;
    h->st = static_cast<hipStream_t>(stream);
    const size_t B = (size_t)batch, n = (size_t)N;
    bool ok = hipMalloc((void **)&h->S, B * sizeof(fl::TrsState)) == hipSuccess &&
              hipMalloc((void **)&h->xcur, B * n * 8) == hipSuccess && hipMalloc((void **)&h->r, B * (size_t)M * 8) == hipSuccess &&
              hipMalloc((void **)&h->A, B * n * n * 8) == hipSuccess && hipMalloc((void **)&h->g, B * n * 8) == hipSuccess &&
              hipMalloc((void **)&h->A_new, B * n * n * 8) == hipSuccess && hipMalloc((void **)&h->g_new, B * n * 8) == hipSuccess &&
              hipMalloc((void **)&h->Ap, B * n * (size_t)h->ld * 8) == hipSuccess &&
              hipMalloc((void **)&h->d, B * n * 8) == hipSuccess && hipMalloc((void **)&h->info, B * 4) == hipSuccess;
    if (ok && low_dev && up_dev) {
        ok = hipMalloc((void **)&h->low, n * 8) == hipSuccess && hipMalloc((void **)&h->up, n * 8) == hipSuccess &&
             hipMemcpyAsync(h->low, low_dev, n * 8, hipMemcpyDeviceToDevice, h->st) == hipSuccess &&
             hipMemcpyAsync(h->up, up_dev, n * 8, hipMemcpyDeviceToDevice, h->st) == hipSuccess;
    }
    if (!ok) {
        fl_trust_region_destroy(h);
        return FL_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(fl::trs_init_kernel, dim3((batch + 255) / 256), dim3(256), 0, h->st, batch, h->S);
    *out = h;
    return fl::launch_status();
}

int fl_trust_region_step(fl_trs *h, double *x_dev, const double *r_dev, const double *J_dev, int32_t *request_dev)
{
    if (!h || !x_dev || !request_dev) return FL_ERR_INVALID_ARGUMENT;
    const int B = h->batch, M = h->M, N = h->N;
    if (h->first) { // x_dev = the starting points: project into the box, ask for residual and Jacobian there
        const size_t tot = (size_t)B * N;
        hipLaunchKernelGGL(fl::trs_project_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->st, B, N, h->low, h->up,
                           h->xcur, x_dev);
        h->first = 0;
    } else {
        if (!r_dev || !J_dev) return FL_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(fl::trs_take_kernel, dim3(B), dim3(256), 0, h->st, M, N, h->maxit, h->maxstepit, h->precision, h->S,
                           h->xcur, x_dev, h->r, r_dev);
        // J^T J and J^T r of what the caller's arrays hold, for every problem (one strided-batch product each); only the
        // problems that had ASKED for a Jacobian take them over (trs_commit_kernel) -- a caller is free to rewrite J for
        // every problem on every step, at whatever point x_dev holds for it
        int rc = fl_dgemm_strided(1, 0, N, M, N, 1.0, J_dev, M, (size_t)M * N, J_dev, M, (size_t)M * N, 0.0, h->A_new, N,
                                  (size_t)N * N, B, 0, h->st);
        if (rc != FL_OK) return rc;
        rc = fl_dgemm_strided(1, 0, N, M, 1, 1.0, J_dev, M, (size_t)M * N, h->r, M, (size_t)M, 0.0, h->g_new, N, (size_t)N, B, 0, h->st);
        if (rc != FL_OK) return rc;
        hipLaunchKernelGGL(fl::trs_commit_kernel, dim3(B), dim3(256), 0, h->st, N, h->S, h->A_new, h->g_new, h->A, h->g);
        hipLaunchKernelGGL(fl::trs_build_kernel, dim3(B), dim3(256), 0, h->st, N, h->ld, h->S, h->A, h->g, h->Ap, h->d);
        rc = fl_dposv_batched(B, N, h->Ap, h->d, h->info, h->st);
        if (rc != FL_OK) return rc;
    }
    hipLaunchKernelGGL(fl::trs_trial_kernel, dim3(B), dim3(256), 0, h->st, N, h->maxstepit, h->minstep, h->S, h->xcur, h->d, h->g,
                       h->info, h->low, h->up, x_dev, request_dev);
    return fl::launch_status();
}

int fl_trust_region_results(fl_trs *h, double *resnorm_dev, int32_t *iters_dev, int32_t *reason_dev)
{
    if (!h) return FL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(fl::trs_results_kernel, dim3((h->batch + 255) / 256), dim3(256), 0, h->st, h->batch, h->S, resnorm_dev,
                       iters_dev, reason_dev);
    return fl::launch_status();
}

} // extern "C"
