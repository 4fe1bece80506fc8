// fl_multi.cpp -- a batch of independent problems over ALL the GPUs of the node from ONE process: the drop-in boundary's
// multi-device entry ("single process with one thread per GPU suffices").  Host arrays in, host arrays
// out, like every entry point of the reference; one host thread per shard does hipSetDevice, uploads its problems,
// runs the fused batched solver on its own stream and writes its rows of the results straight into the caller's arrays
// -- the problems are independent, so there is no collective and no exchange beyond that write-back (the multi-process
// form, one rank per GPU with one RCCL gather, is FortranLibrary/distributed.py + bench.py).
//
// Shards: contiguous blocks of ceil(batch / nshards) problems, or interleaved (problem k -> shard k mod nshards, which
// spreads any trend of the iteration counts along the batch evenly over the devices).  nshards <= 0: up to four shards per
// visible device (see fl_multi_solve); nshards > devices: shards share devices round-robin.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../include/fl_nlopt.h"

namespace {

struct Job {
    int solver, objective, batch, n, aug_m, nshards, interleaved;
    double *x;
    const double *d, *b;
    const fl_options *opt;
    double *lambda, miu0;
    double *f, *gg, *cnorm2;
    int32_t *iters, *outer, *status, *nf, *ng;
};

struct Dev { // device buffers of one shard, released on every path out
    std::vector<void *> p;
    template <class T> T *get(size_t count)
    {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
        p.push_back(q);
        return static_cast<T *>(q);
    }
    ~Dev()
    {
        for (void *q : p) (void)hipFree(q);
    }
};

// rows first, first + step, ... (count of them) of a host array [batch][width] <-> a dense device array [count][width]
template <class T> bool rows_h2d(T *dst, const T *src, int first, int step, int count, int width, hipStream_t st)
{
    if (!src || count == 0) return true;
    return hipMemcpy2DAsync(dst, (size_t)width * sizeof(T), src + (size_t)first * width, (size_t)step * width * sizeof(T),
                            (size_t)width * sizeof(T), count, hipMemcpyHostToDevice, st) == hipSuccess;
}
template <class T> bool rows_d2h(T *dst, const T *src, int first, int step, int count, int width, hipStream_t st)
{
    if (!dst || count == 0) return true;
    return hipMemcpy2DAsync(dst + (size_t)first * width, (size_t)step * width * sizeof(T), src, (size_t)width * sizeof(T),
                            (size_t)width * sizeof(T), count, hipMemcpyDeviceToHost, st) == hipSuccess;
}

int run_shard(const Job &J, int shard, int device)
{
    const int S = J.nshards, n = J.n;
    int first, step, count;
    if (J.interleaved) {
        first = shard;
        step = S;
        count = shard < J.batch ? (J.batch - shard + S - 1) / S : 0;
    } else {
        const int per = (J.batch + S - 1) / S;
        first = std::min(J.batch, shard * per);
        step = 1;
        count = std::min(J.batch, first + per) - first;
    }
    if (count <= 0) return FL_OK;
    if (hipSetDevice(device) != hipSuccess) return FL_ERR_NO_DEVICE;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return FL_ERR_LAUNCH;
    int rc = FL_OK;
    {
        Dev D;
        const size_t N = (size_t)count * n;
        double *x = D.get<double>(N), *d = J.d ? D.get<double>(N) : nullptr, *b = J.b ? D.get<double>(N) : nullptr;
        double *f = D.get<double>(count), *gg = D.get<double>(count);
        int32_t *it = D.get<int32_t>(count), *stt = D.get<int32_t>(count), *nf = D.get<int32_t>(count), *ng = D.get<int32_t>(count);
        double *lam = J.aug_m ? D.get<double>((size_t)count * J.aug_m) : nullptr, *cn = J.aug_m ? D.get<double>(count) : nullptr;
        int32_t *outer = J.aug_m ? D.get<int32_t>(count) : nullptr;
        const size_t wsb = fl_workspace_bytes_for(J.solver, count, n, J.opt);
        void *ws = wsb ? D.get<char>(wsb) : nullptr;
        bool ok = x && f && gg && it && stt && nf && ng && (!J.d || d) && (!J.b || b) && (!wsb || ws) &&
                  (!J.aug_m || (lam && cn && outer));
        if (!ok) rc = FL_ERR_WORKSPACE;
        if (ok) {
            ok = rows_h2d(x, J.x, first, step, count, n, st) && rows_h2d(d, J.d, first, step, count, n, st) &&
                 rows_h2d(b, J.b, first, step, count, n, st);
            if (ok && J.aug_m) {
                if (J.lambda) ok = rows_h2d(lam, J.lambda, first, step, count, J.aug_m, st);
                else ok = hipMemsetAsync(lam, 0, (size_t)count * J.aug_m * sizeof(double), st) == hipSuccess; // lambda0 = 0
            }
            if (!ok) rc = FL_ERR_LAUNCH;
        }
        if (ok) {
            if (J.aug_m)
                rc = fl_augmented_lagrangian_batched(J.solver, J.objective, count, n, J.aug_m, x, d, b, lam, J.miu0, J.opt, ws, wsb, f, cn,
                                                     it, outer, stt, nf, ng, st);
            else if (J.solver == FL_SOLVER_SD)
                rc = fl_steepest_descent_batched(J.objective, count, n, x, d, b, J.opt, f, gg, it, stt, nf, ng, st);
            else if (J.solver == FL_SOLVER_CG)
                rc = fl_conjugate_gradient_batched(J.objective, count, n, x, d, b, J.opt, f, gg, it, stt, nf, ng, st);
            else if (J.solver == FL_SOLVER_BFGS)
                rc = fl_bfgs_batched(J.objective, count, n, x, d, b, J.opt, ws, wsb, f, gg, it, stt, nf, ng, st);
            else if (J.solver == FL_SOLVER_NEWTON)
                rc = fl_newton_raphson_batched(J.objective, count, n, x, d, b, J.opt, ws, wsb, f, gg, it, stt, nf, ng, st);
            else
                rc = fl_lbfgs_batched(J.objective, count, n, x, d, b, J.opt, ws, wsb, f, gg, it, stt, nf, ng, st);
        }
        if (rc == FL_OK) { // this shard's rows of the results, straight into the caller's arrays
            ok = rows_d2h(J.x, x, first, step, count, n, st) && rows_d2h(J.f, f, first, step, count, 1, st) &&
                 rows_d2h(J.iters, it, first, step, count, 1, st) && rows_d2h(J.status, stt, first, step, count, 1, st) &&
                 rows_d2h(J.nf, nf, first, step, count, 1, st) && rows_d2h(J.ng, ng, first, step, count, 1, st);
            if (ok && J.aug_m)
                ok = rows_d2h(J.lambda, lam, first, step, count, J.aug_m, st) && rows_d2h(J.cnorm2, cn, first, step, count, 1, st) &&
                     rows_d2h(J.outer, outer, first, step, count, 1, st);
            else if (ok)
                ok = rows_d2h(J.gg, gg, first, step, count, 1, st);
            if (!ok) rc = FL_ERR_LAUNCH;
        }
        if (hipStreamSynchronize(st) != hipSuccess && rc == FL_OK) rc = FL_ERR_LAUNCH;
    } // (buffers released after the stream has drained)
    (void)hipStreamDestroy(st);
    return rc;
}

} // namespace

extern "C" {

int fl_multi_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return ndev;
}

int fl_multi_solve(int solver, int objective, int batch, int n, double *x_host, const double *d_host, const double *b_host,
                   const fl_options *opt, int aug_m, double *lambda_host, double miu0, double *f_host, double *gg_host,
                   double *cnorm2_host, int32_t *iters_host, int32_t *outer_host, int32_t *status_host, int32_t *nf_host,
                   int32_t *ng_host, int nshards, int interleaved)
{
    if (!x_host || !opt || batch <= 0 || n <= 0 || aug_m < 0) return FL_ERR_INVALID_ARGUMENT;
    if (objective == FL_OBJ_DIAGQUAD && (!d_host || !b_host)) return FL_ERR_INVALID_ARGUMENT;
    const int ndev = fl_multi_device_count();
    if (ndev <= 0) return FL_ERR_NO_DEVICE;
    // default: four shards per device while a shard keeps at least 4096 problems -- the shards of a device run on their own
    // threads and streams, so one shard's transfers overlap another's solve (headline batch on one GPU: 225 -> 196 ms)
    int per_dev = 1;
    if (nshards <= 0) per_dev = std::max(1, std::min(4, batch / (ndev * 4096)));
    int S = nshards > 0 ? nshards : ndev * per_dev;
    S = std::min(S, batch);
    int prev = 0;
    (void)hipGetDevice(&prev);
    Job J{solver, objective, batch, n, aug_m, S, interleaved != 0, x_host, d_host, b_host, opt, lambda_host, miu0,
          f_host, gg_host, cnorm2_host, iters_host, outer_host, status_host, nf_host, ng_host};
    std::vector<int> rc(S, FL_OK);
    std::vector<std::thread> th;
    th.reserve(S);
    for (int s = 0; s < S; ++s) th.emplace_back([&, s] { rc[s] = run_shard(J, s, s % ndev); });
    for (auto &t : th) t.join();
    (void)hipSetDevice(prev);
    for (int s = 0; s < S; ++s)
        if (rc[s] != FL_OK) return rc[s];
    return FL_OK;
}

} // extern "C"
