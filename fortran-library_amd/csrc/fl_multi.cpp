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
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/fl_nlopt.h"

extern "C" void fl_internal_set_concurrent_batch(int problems); // (csrc/fl_solver_kernels.hip; not part of the public ABI)

namespace {

// The caller's big arrays are pageable memory: the runtime stages every copy through its own pinned buffers (2.1 GB of the
// headline batch: 43 of 198 ms, profiles/r03/host_arrays.txt).  FL_MULTI_PIN=1 in the environment registers them
// (page-locked in place, hipHostRegister) for the duration of the call, so that the shards' 2D copies go by DMA straight
// from / to them -- best effort: a range that cannot be registered is copied the old way.  OFF by default: measured on the
// headline batch (profiles/r04/host_arrays.txt) the registration of 1.6 GB costs what the faster copies save (default
// shards 201.5 against 205.3 ms, one shard 224.3 against 227.4, eight shards 204.4 against 198.1): the call is bound by the
// 2.1 GB crossing PCIe (~40 ms at 55 GB/s), most of which the shards of a device already overlap with each other's solves.
struct Pinned {
    std::vector<void *> ranges;
    void add(const void *p, size_t bytes)
    {
        if (!p || bytes < ((size_t)8 << 20)) return; // (small arrays: the registration costs more than it saves)
        if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterDefault) == hipSuccess) ranges.push_back(const_cast<void *>(p));
        else (void)hipGetLastError();
    }
    ~Pinned()
    {
        for (void *p : ranges) (void)hipHostUnregister(p);
    }
};

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

// shard s of J: its first problem and how many it holds (rows first, first + step, ...; step = nshards when interleaved)
int shard_first(const Job &J, int s)
{
    if (J.interleaved) return s;
    const int per = (J.batch + J.nshards - 1) / J.nshards;
    return std::min(J.batch, s * per);
}
int shard_count(const Job &J, int s)
{
    if (J.interleaved) return s < J.batch ? (J.batch - s + J.nshards - 1) / J.nshards : 0;
    const int per = (J.batch + J.nshards - 1) / J.nshards, first = shard_first(J, s);
    return std::min(J.batch, first + per) - first;
}

int run_shard(const Job &J, int shard, int device)
{
    const int S = J.nshards, n = J.n;
    const int first = shard_first(J, shard), step = J.interleaved ? S : 1, count = shard_count(J, shard);
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
    Pinned pin;
    {
        const char *e = std::getenv("FL_MULTI_PIN");
        if (e && e[0] == '1') {
            const size_t bytes = (size_t)batch * n * sizeof(double);
            pin.add(x_host, bytes);
            pin.add(d_host, bytes);
            pin.add(b_host, bytes);
        }
    }
    std::vector<int> rc(S, FL_ERR_LAUNCH); // (a shard whose thread could not even be started stays "failed")
    std::vector<std::thread> th;
    th.reserve(S);
    // how many problems each device holds at once: the shards of a device run concurrently, so a kernel's choices by batch
    // size (helper waves: fl_solver_kernels.hip, select_replicas) must see the device's load, not the shard's
    std::vector<int> dev_load(ndev, 0);
    for (int s = 0; s < S; ++s) dev_load[s % ndev] += shard_count(J, s);
    for (int s = 0; s < S; ++s) {
        try {
            th.emplace_back([&, s] {
                fl_internal_set_concurrent_batch(dev_load[s % ndev]);
                rc[s] = run_shard(J, s, s % ndev);
                fl_internal_set_concurrent_batch(0);
            });
        } catch (...) { // (std::system_error: no more threads) -- the shards already started finish, the rest are reported
            break;
        }
    }
    for (auto &t : th) t.join();
    (void)hipSetDevice(prev);
    int first_error = FL_OK;
    for (int s = 0; s < S; ++s) {
        if (rc[s] == FL_OK) continue;
        if (first_error == FL_OK) first_error = rc[s];
        // which rows hold results and which still hold the initial guesses: the failed shards' status rows say so
        if (status_host) {
            const int first = J.interleaved ? s : shard_first(J, s), step = J.interleaved ? S : 1, count = shard_count(J, s);
            for (int k = 0; k < count; ++k) status_host[(size_t)first + (size_t)k * step] = FL_STATUS_NOT_SOLVED;
        }
    }
    return first_error;
}

} // extern "C"
