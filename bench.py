#!/usr/bin/env python3
"""bench.py -- batched line-search optimisers on MI355X: iterations/sec, rooflines, CPU baseline, parity.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run (one rank per GPU, RCCL) -- started without a launcher it spawns one itself, as a child
process, before anything touches the GPU.  Rank 0 prints ONE JSON line.

HEADLINE (the top-level keys of the line).  A "step" is one complete pass of the hot path over one batch: every
problem of the batch is solved from its initial guess to convergence by ONE launch of the fused solver kernel
(fl_lbfgs_batched), then -- when N > 1 -- the converged results (x*, f*, iterations, status) are gathered to rank 0
with one RCCL gather per array (buffers allocated before the timed region).  The workload is the north-star
configuration: L-BFGS, Memory = 10, n = 1024 convex diagonal quadratics of BASELINE.json's config-3 family (kappa
log-uniform in [10, 1000], b ~ U(-1,1), x0 = 0, Precision = 1e-6: the tightest gradient tolerance every problem of
this family can meet with an fp64 objective-value line search), inputs generated on the device (Philox, stream =
global problem id) and resident in HBM before the timed region.
  --scaling weak   (default) 65536 problems PER GPU;
  --scaling strong --batch problems in ALL, rank r owns the problems k = r mod N (or, --contiguous, a block).
With N > 1 BOTH legs are measured in the one invocation: the one --scaling names is the top-level record, the
other is reported under "other_leg" with its own `scaling`.
value = L-BFGS iterations (line searches) of all ranks / sec.

roofline (the fused solver kernel is the only kernel in the timed region; its duration comes from HIP events on the
launch stream):
  * traffic   = bytes through the L2's memory side per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB from separate
                rocprofv3 --pmc passes of this very workload (MI355X_MICROARCH.md, HBM: FETCH_SIZE counts 64 of the
                128 bytes of a wide read on gfx950); taken from profiles/traffic.json only while its record is of
                the same kernel sources and workload;
  * achieved  = traffic / kernel time (without a valid record: the model below / kernel time), frac = achieved /
                8 TB/s -- a bound that binds: <= 1 by construction;
  * model     = the bytes the shipped kernel asks the L2 for: with C pairs of the ring on the chip
                (fl_lbfgs_onchip_pairs) an iteration loads max(0,2(cnt-C)) + max(0,2(cnt-C-2)) rows, stores 2;
  * algorithmic_bw = SURVEY.md 8(d)'s streaming figure, (4*cnt+2)*8n per iteration and problem, / kernel time: a
                rate, not a utilisation (rows served from registers / LDS never cross the pins);
  * trial_phase = strong-Wolfe trials per iteration (live); vector instructions / wave cycles per trial (record).
cpu_baseline: the CPU oracle (reference summation order), OpenMP over problems on all host cores, on a bounded
sample of rank 0's batch; parity: the same sample against the GPU result, 256 problems bit for bit against the
oracle in the kernels' summation order, and the north-star minimiser bar on the sub-family where it is defined.

"configs" (N = 1): the other BASELINE.json configurations, each one launch of its fused kernel on its full size,
each with ms, its metric, `roofline`, `cpu_baseline` (oracle, >= 2 s sample) and `parity` (bit-exact subset + the
objective tolerance).  Rooflines: C4 = HBM from its keyed PMC record; C4_gemm = f64 MFMA (flops / time, live); C2,
C3, C5 keep their vectors on the chip and stream (next to) nothing: their bound is the SIMDs' vector issue rate --
`bound: "valu_issue"`, achieved = vector issue slots used per second (SQ_ACTIVE_INST_VALU of the keyed record /
live kernel time), peak = 1024 SIMDs x shader clock / 4.  With N > 1: BASELINE's multi-GPU configs C3 and C5,
sharded over the ranks (strong), one gather each.
`--only-config cX` runs one configuration alone (the rocprofv3 passes of tools/profile.sh).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
MFMA_F64_PEAK_TF = 78.6  # dense FP64 matrix peak, AMD datasheet
N_SIMD = 1024            # 256 CUs x 4 SIMDs
SHADER_CLOCK_HZ = 2.4e9  # peak engine clock; a PMC record carries the clock it measured (GRBM_GUI_ACTIVE / time)
SEED = 20261003
CONFIG_KEYS = ("c2", "c3", "c4", "c4gemm", "c5")


def kernel_source_hash():
    """identifies the kernels a PMC record was taken from: the sources the device code is compiled from"""
    import hashlib
    h = hashlib.sha256()
    for f in ("fl_device.hpp", "fl_reduce.hpp", "fl_linesearch.hpp", "fl_dense.hpp", "fl_solver_launch.hpp",
              "fl_bfgs_gemm.hip"):
        h.update(open(os.path.join(ROOT, "fortran-library_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def host_cores():
    """CPU cores this process may really use: min(online, affinity mask, cgroup quota)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            c = min(c, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return c


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_record(key, wl):
    """(record, stale): the PMC record of profiles/traffic.json for `key` if it is of this workload -- `record` when it
    was taken from the kernel sources of this tree, `stale` when from others"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except (OSError, ValueError):
        return None, None
    if not rec or any(rec.get(k) != v for k, v in wl.items()):
        return None, None
    if rec.get("kernel_source_hash") == kernel_source_hash():
        return rec, None
    return None, rec


def valu_roofline(rec, stale, kern_ms, kernel):
    """vector-issue roofline of a kernel that streams nothing: issue slots used per second against 1024 SIMDs x clock / 4"""
    r = rec or stale
    out = {"bound": "valu_issue", "kernel": kernel, "unit": "G issue slots/s", "kernel_ms": kern_ms, "traffic": None}
    if not r or not r.get("SQ_ACTIVE_INST_VALU"):
        out.update({"achieved": None, "peak": N_SIMD * SHADER_CLOCK_HZ / 4 / 1e9, "frac": None,
                    "achieved_source": "no PMC record for this workload (tools/profile.sh + tools/pmc_summary.py --record)"})
        return out
    clk = r.get("shader_clock_hz") or SHADER_CLOCK_HZ
    ach = r["SQ_ACTIVE_INST_VALU"] / (kern_ms * 1e-3) / 1e9
    peak = N_SIMD * clk / 4 / 1e9
    out.update({"achieved": ach, "peak": peak, "frac": ach / peak,
                "achieved_source": ("SQ_ACTIVE_INST_VALU of the keyed PMC record / live kernel time"
                                    + ("" if rec else " -- record taken from OTHER kernel sources: re-profile")),
                "shader_clock_hz": clk, "valu_insts_per_launch": r.get("SQ_INSTS_VALU"),
                "waves_waiting_fraction": (r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"]) if r.get("SQ_WAIT_ANY") and r.get("SQ_WAVE_CYCLES") else None,
                "mean_waves_per_simd": (4.0 * r["SQ_WAVE_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * N_SIMD)) if r.get("GRBM_GUI_ACTIVE") and r.get("SQ_WAVE_CYCLES") else None,
                "memory_side_bytes_per_launch": r.get("traffic_bytes_per_launch"),
                "hbm_frac": (r["traffic_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if r.get("traffic_bytes_per_launch") else None,
                "source": r.get("source")})
    return out


# ------------------------------------------------------------------------------------------------ context
class Ctx:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("FL_BENCH_ONE_DEVICE"):  # rehearsal: every rank on GPU 0 (needs --backend gloo)
            local = 0
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        # FL_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, gatherer, collectives of the report) with a
        # world of one -- the only way to run that path through RCCL on a one-GPU box (tests/test_gpu_bench.py)
        self.multi = self.world > 1 or bool(os.environ.get("FL_BENCH_FORCE_DIST"))
        self.backend = args.backend
        if self.multi:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(args.backend, rank=self.rank, world_size=self.world)
        self.cdev = self.dev if args.backend == "nccl" else torch.device("cpu")
        self.cores = host_cores()

    def sync(self, gat=None):
        if gat is not None:
            gat.finish()
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def allmax(self, v):
        t = self.torch.tensor([v], dtype=self.torch.float64, device=self.cdev)
        if self.multi:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_scalar(self, v, dtype=None):
        torch = self.torch
        t = torch.tensor([v], dtype=dtype or torch.float64, device=self.cdev)
        if not self.multi:
            return [t.item()]
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.item() for o in out]

    def close(self):
        if self.multi:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_launches(ctx, fn, reps, warm=1):
    """average HIP-event duration of fn() (one kernel launch on the current stream) over reps launches"""
    torch = ctx.torch
    out = None
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return out, sum(ms) / len(ms)


def cpu_sample(call, S0, Smax, seconds):
    """time the oracle: pilot on S0 problems, then a sample sized for `seconds` of wall time (at most Smax problems).
    call(S) -> result dict; returns (result, S, wall seconds)"""
    t = time.perf_counter()
    r = call(S0)
    pilot = max(time.perf_counter() - t, 1e-3)
    S = int(min(Smax, max(S0, S0 * seconds / pilot)))
    if S == S0 and pilot >= 0.8 * seconds:
        return r, S0, pilot
    t = time.perf_counter()
    r = call(S)
    return r, S, time.perf_counter() - t


def bits_equal(a, b):
    import numpy as np
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype == np.float64:
        return bool(np.array_equal(a.view(np.uint64), b.view(np.uint64)))
    return bool(np.array_equal(a, b))


# ------------------------------------------------------------------------------------------------ headline
def headline_leg(ctx, args, NLO, scaling, interleaved, full_report):
    """one leg (weak or strong) of the headline workload: W warm-up steps, K timed steps between barriers, max over ranks.
    Returns (record, state) on rank 0 -- state holds what the parity / CPU legs need -- and (None, None) elsewhere."""
    import numpy as np
    torch, dist = ctx.torch, ctx.dist
    from FortranLibrary import distributed as D
    dev, world, rank = ctx.dev, ctx.world, ctx.rank
    if args.workload == "lbfgs_rosen256":
        n, objective = 256, NLO.ROSENBROCK
        batch = 4096 if args.batch == 65536 else args.batch
        opt_kw = dict(Precision=1e-10, MaxIteration=3000, Memory=args.memory)
    else:
        n, objective, batch = args.n, NLO.DIAGQUAD, args.batch
        opt_kw = dict(Precision=1e-6, MaxIteration=3000, Memory=args.memory)
    m = args.memory
    if scaling == "strong":
        Bglobal = batch
        idx = D.shard_indices(Bglobal, rank, world, interleaved).to(dev)
        B = int(idx.numel())
    else:
        Bglobal, idx, B = batch * world, None, batch

    def synth(Bgen, seed):  # synthetic inputs, generated on the device, resident in HBM
        xs = torch.zeros(Bgen, n, dtype=torch.float64, device=dev)
        ds = bs = None
        if objective == NLO.DIAGQUAD:
            ds = torch.empty(Bgen, n, dtype=torch.float64, device=dev)
            bs = torch.empty(Bgen, n, dtype=torch.float64, device=dev)
            NLO.synth_diag_spectrum(seed, ds, 10.0, 1000.0)
            NLO.synth_uniform(seed, bs, -1.0, 1.0)
        else:
            NLO.synth_uniform(seed, xs, 0.9, 1.1)  # x0 = 1 + 0.1 u
        return xs, ds, bs

    if idx is None:
        x0, d, b = synth(B, SEED + rank)
    else:  # the generator's stream is the problem id: generate the global batch, keep this rank's problems
        xg, dg, bg = synth(Bglobal, SEED)
        x0 = xg[idx].contiguous()
        d = dg[idx].contiguous() if dg is not None else None
        b = bg[idx].contiguous() if bg is not None else None
        del xg, dg, bg
        torch.cuda.empty_cache()
    x = torch.empty_like(x0)
    ws = NLO.workspace(B, n, m, dev)
    opts = NLO.default_options(NLO.LBFGS_, **opt_kw)
    gat = None
    if ctx.multi:  # the exchange's buffers exist before the timed region
        gat = D.Gatherer(Bglobal, {"x": ((n,), torch.float64), "f": ((), torch.float64), "iters": ((), torch.int32),
                                   "status": ((), torch.int32)}, dev, dst=0,
                         interleaved=(scaling == "strong" and interleaved))
    ev = []

    def step(record):
        x.copy_(x0)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = NLO.LBFGS(objective, x, d, b, workspace_=ws, options=opts)
        e1.record()
        if record:
            ev.append((e0, e1))
        if gat is not None:  # the single exchange of the path: converged results to rank 0 over xGMI
            # (enqueued on RCCL's stream: it overlaps the next step's solve; sync() waits for the last one)
            gat.gather({"x": x, "f": out["f"], "iters": out["iters"], "status": out["status"]}, overlap=True)
        return out

    out = None
    for _ in range(args.warmup):
        out = step(False)
    ctx.sync(gat)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    ctx.sync(gat)
    dt = ctx.allmax(time.perf_counter() - t0)
    per_rank_iters = [int(v) for v in ctx.allgather_scalar(int(out["iters"].to(torch.int64).sum().item()), torch.int64)]
    kern_ms = sum(a.elapsed_time(bb) for a, bb in ev) / max(1, len(ev))
    per_rank_kernel_ms = [float(v) for v in ctx.allgather_scalar(kern_ms)]
    gather_ms = None
    if ctx.multi:  # the exchange alone, outside the timed region: barrier, gather, synchronise
        ctx.sync(gat)
        tg = time.perf_counter()
        gat.gather({"x": x, "f": out["f"], "iters": out["iters"], "status": out["status"]})
        ctx.sync(gat)
        gather_ms = (time.perf_counter() - tg) * 1e3
    if rank != 0:
        return None, None
    total = sum(per_rank_iters)
    scaling_note = ("weak: %d problems per GPU" % B) if scaling == "weak" else (
        "strong: %d problems in all, %s shards" % (Bglobal, "interleaved" if interleaved else "contiguous"))
    rec = {
        "metric": "lbfgs_iterations_per_sec",
        "value": total * args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"L-BFGS m={m}, {scaling_note}, n={n}, "
                               + ("convex diagonal quadratics kappa in [10,1000], Precision 1e-6"
                                  if objective == NLO.DIAGQUAD else "chained Rosenbrock x0=1+0.1u, Precision 1e-10"),
                   "batch_per_gpu": B, "global_batch": Bglobal, "n": n, "memory": m, "solver": "LBFGS",
                   "line_search": "StrongWolfe",
                   "exchange": "gather x*,f*,iters,status to rank 0" if ctx.multi else "none"},
        "params_per_sec": total * args.steps / dt * n,
        "iterations_per_step": total,
        "ranks": {"backend": ctx.backend if ctx.multi else "none", "world_size": world,
                  "iterations_per_rank": per_rank_iters, "kernel_ms_per_rank": per_rank_kernel_ms,
                  "gather_ms": gather_ms, "exchange_overlaps_next_solve": bool(ctx.multi)},
        "converged_fraction": float((out["status"] == 0).double().mean().item()),
    }
    if not full_report:
        return rec, None
    st = dict(n=n, m=m, B=B, objective=objective, opts=opts, opt_kw=opt_kw, x0=x0, d=d, b=b, x=x, ws=ws, out=out,
              kern_ms=kern_ms)
    return rec, st


def headline_roofline(ctx, args, NLO, st):
    torch = ctx.torch
    n, m, B, objective, out, kern_ms = st["n"], st["m"], st["B"], st["objective"], st["out"], st["kern_ms"]
    it = out["iters"].to(torch.int64)
    k = torch.clamp(it - 1, min=0)  # two-loops performed per problem
    T_, E_ = NLO.reduction_geometry(n)
    row_bytes = 8 * T_ * E_

    def sum_over_two_loops(fn):
        """sum over the two-loops j = 1..k of every problem of fn(cnt), cnt = min(j, m)"""
        tot = 0
        for c in range(1, m + 1):
            times = torch.where(k >= c, torch.ones_like(k), torch.zeros_like(k)) if c < m else torch.clamp(k - m + 1, min=0)
            tot += int(times.sum().item()) * fn(c)
        return tot

    C_on = NLO.lbfgs_onchip_pairs(objective, n)
    algo_bytes = float(sum_over_two_loops(lambda c: (4 * c + 2) * 8 * n))
    writes = 2 if m > C_on else 0
    model_bytes = float(sum_over_two_loops(lambda c: (max(0, 2 * (c - C_on)) + max(0, 2 * (c - C_on - 2)) + writes)
                                           * row_bytes))
    trials = int(out["nf"].to(torch.int64).sum().item())
    wl_key = {"workload": args.workload, "batch_per_gpu": B, "n": n, "memory": m, "precision": st["opts"].precision}
    pmc, stale = pmc_record(args.workload, wl_key)
    traffic = float(pmc["traffic_bytes_per_launch"]) if pmc else None
    traffic_source = "pmc traffic" if pmc else "minimum-traffic model"
    moved = traffic if traffic is not None else model_bytes
    if pmc is None and stale and stale.get("model_bytes_per_launch"):
        # the share of the L2 requests that reaches the memory side is a property of the working set, not of the
        # instruction stream: scale the live model by the recorded ratio rather than present L2 requests as HBM bytes
        moved = model_bytes * float(stale["traffic_bytes_per_launch"]) / float(stale["model_bytes_per_launch"])
        traffic_source = ("minimum-traffic model x the traffic / model ratio of a PMC record taken from other kernel "
                          "sources (re-run tools/profile.sh + tools/pmc_summary.py --record)")
    achieved = moved / (kern_ms * 1e-3) / 1e9
    waves_per_problem = T_ // 64
    trial_phase = {"trials_per_launch": trials, "trials_per_iteration": trials / max(1, int(it.sum().item()))}
    if pmc and pmc.get("SQ_INSTS_VALU"):
        # SQ_* are sums over all waves; a trial is executed by every wave of its problem
        trial_phase["valu_insts_per_wave_and_trial_upper_bound"] = pmc["SQ_INSTS_VALU"] / (pmc["trials_per_launch"] * waves_per_problem)
        trial_phase["wave_cycles_per_trial_upper_bound"] = 4.0 * pmc["SQ_WAVE_CYCLES"] / (pmc["trials_per_launch"] * waves_per_problem)
        trial_phase["valu_busy_fraction"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_ACTIVE_INST_VALU") else None
        trial_phase["waiting_fraction"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAIT_ANY") else None
        trial_phase["note"] = ("upper bounds: all vector instructions / wave cycles of the launch divided by its trials "
                               "(the two-loop recursion's share is included); source: " + ", ".join(pmc.get("source", [])))
    return {"bound": "hbm", "kernel": "fl_solve_kernel<NW,EPT,OBJ,LBFGS> (fused solver)",
            "achieved": achieved, "achieved_source": traffic_source,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel_ms": kern_ms,
            "model_bytes_per_launch": model_bytes, "model_bw": model_bytes / (kern_ms * 1e-3) / 1e9,
            "traffic_over_model": (traffic / model_bytes) if traffic is not None else None,
            "onchip_pairs": C_on,
            "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bw": algo_bytes / (kern_ms * 1e-3) / 1e9,
            "trial_phase": trial_phase,
            "l2_hit_rate": (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]))
            if pmc and pmc.get("TCC_HIT_sum") else None,
            "note": "frac = bytes that crossed the L2's memory side (PMC: Infinity-Cache hits included) / kernel "
                    "time / 8 TB/s; model = bytes the kernel requests from L2 (ring rows not kept on the chip); "
                    "algorithmic_bw = SURVEY 8d streaming figure (4*cnt+2)*8n per iteration, credited for rows "
                    "served on chip, so it may exceed the peak and is not a utilisation"}


def headline_two_loop(ctx, NLO, st):
    """the stand-alone two-loop recursion on the solver's own final history"""
    torch = ctx.torch
    n, m, B, ws, dev = st["n"], st["m"], st["B"], st["ws"], ctx.dev
    T, E = NLO.reduction_geometry(n)
    npad = T * E
    hist = ws[: B * 2 * m * npad].view(B, 2 * m, npad)
    sub = min(B, 16384)
    rho = 1.0 / (hist[:sub, 0::2, :] * hist[:sub, 1::2, :]).sum(dim=2)
    rho = torch.where(torch.isfinite(rho), rho, torch.ones_like(rho))
    rho_all = torch.ones(B, m, dtype=torch.float64, device=dev)
    rho_all[:sub] = rho
    g = torch.empty(B, n, dtype=torch.float64, device=dev)
    NLO.synth_uniform(SEED + 99, g, -1.0, 1.0)
    p = torch.empty_like(g)
    _, tl_ms = timed_launches(ctx, lambda: NLO.two_loop(hist, rho_all, g, p, m, m - 1), 10, warm=2)
    tl_bytes = float(B) * (4 * m + 2) * 8 * n      # SURVEY 8d accounting
    tl_moved = float(B) * (4 * m - 2 + 2) * 8 * n  # what the kernel moves: the oldest pair is fetched once
    return {"kernel": "two_loop_kernel<NW,EPT>", "achieved": tl_moved / (tl_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": tl_moved / (tl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": tl_ms,
            "moved_bytes_per_launch": tl_moved, "algorithmic_bytes_per_launch": tl_bytes,
            "algorithmic_bw": tl_bytes / (tl_ms * 1e-3) / 1e9,
            "note": "stand-alone micro-kernel (the product runs the recursion inside the fused kernel); "
                    "22.5 GB per launch re-read ten times in a row: partly Infinity-Cache resident"}


def headline_cpu_and_parity(ctx, args, NLO, st, res):
    """CPU baseline (oracle, reference order) on a bounded sample of the same workload + the parity block"""
    import numpy as np
    import oracle_lib as O
    torch = ctx.torch
    n, m, B, objective, opts, opt_kw = st["n"], st["m"], st["B"], st["objective"], st["opts"], st["opt_kw"]
    x0, d, b, x, out = st["x0"], st["d"], st["b"], st["x"], st["out"]
    cores = ctx.cores
    oo = O.defaults(precision=opts.precision, maxit=opts.max_iteration, memory=m)
    kind = O.DIAGQUAD if objective == NLO.DIAGQUAD else O.ROSENBROCK

    def call(S, **kw):
        return O.solve_batch(O.LBFGS, kind, x0[:S].cpu().numpy(), d=d[:S].cpu().numpy() if d is not None else None,
                             b=b[:S].cpu().numpy() if b is not None else None, opts=kw.pop("opts", oo), nthreads=cores, **kw)

    if args.cpu_sample > 0:
        S = min(B, args.cpu_sample)
        t1 = time.perf_counter()
        ref = call(S, sum_mode=O.SEQ)
        cdt = time.perf_counter() - t1
    else:  # pilot on 4 problems per core, then a sample sized for ~12 s of wall time
        ref, S, cdt = cpu_sample(lambda s: call(min(B, s), sum_mode=O.SEQ), min(B, 4 * cores), B, 12.0)
        S = min(B, S)
    res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / cdt, "unit": "iterations/s",
                           "cores": int(ref["threads"]), "kind": "port",
                           "sample": f"first {S} problems of rank 0's batch, oracle in reference summation "
                                     f"order, OpenMP one problem per thread, {cdt:.1f} s wall; host: {cpu_model()}"}
    gx = x[:S].cpu().numpy()
    gf = out["f"][:S].cpu().numpy()
    ferr = np.abs(gf - ref["f"]) / np.maximum(np.abs(ref["f"]), 1e-10)
    xerr = np.linalg.norm(gx - ref["x"], axis=1) / np.maximum(1.0, np.linalg.norm(ref["x"], axis=1))
    par = {"sample": S, "tolerance": {"f_rel": 1e-10, "x": 1e-8},
           "final_f_rel_err_max": float(ferr.max()),
           "f_within_tolerance_fraction": float((ferr <= 1e-10).mean()),
           "benched_precision_minimiser": {
               "minimiser_err_max": float(xerr.max()), "x_within_1e-8_fraction": float((xerr <= 1e-8).mean()),
               "note": "context, not the bar: at the benched Precision = 1e-6 the reference-order CPU run and the GPU run "
                       "stop at different points of the set ||g|| < 1e-6, which with kappa up to 1e3 is ~1e-6 wide in x; "
                       "the 1e-8 bar is evaluated in `minimiser_bar` below, where a minimiser is defined that sharply"}}
    # the same problems against the oracle in the kernels' summation order: every bit must agree
    T_, E_ = NLO.reduction_geometry(n)
    SB = min(S, 256)
    tre = call(SB, sum_mode=O.TREE, threads=T_, ept=E_)
    bit = {"problems": SB, "x": bits_equal(gx[:SB], tre["x"]), "f": bits_equal(gf[:SB], tre["f"]),
           "iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"])}
    par["bit_exact_vs_oracle_kernel_order"] = bit
    par["ok_f"] = bool(ferr.max() <= 1e-10)
    par["ok_bitexact"] = bool(bit["x"] and bit["f"] and bit["iterations"])
    par["ok"] = par["ok_f"] and par["ok_bitexact"]
    if objective == NLO.DIAGQUAD:
        # The north-star bars (f: 1e-10 relative, minimiser: 1e-8) where a minimiser is defined that sharply: the
        # problems of the benched batch with kappa <= 100 (d's last entry is kappa), solved again by both sides at a
        # gradient tolerance nearly all of them attain.  x <= 1e-8 is REQUIRED there (parity.ok).
        tp = args.bar_precision
        kap = d[:, n - 1]
        sel = torch.nonzero(kap <= 100.0).flatten()[: args.bar_problems]
        NB = int(sel.numel())
        xs_, ds_, bs_ = x0[sel].contiguous(), d[sel].contiguous(), b[sel].contiguous()
        xt = xs_.clone()
        ot = NLO.LBFGS(objective, xt, ds_, bs_, options=NLO.default_options(NLO.LBFGS_, **dict(opt_kw, Precision=tp)))
        torch.cuda.synchronize()
        dh, bh = ds_.cpu().numpy(), bs_.cpu().numpy()
        rt = O.solve_batch(O.LBFGS, kind, xs_.cpu().numpy(), d=dh, b=bh,
                           opts=O.defaults(precision=tp, maxit=opts.max_iteration, memory=m), sum_mode=O.SEQ, nthreads=cores)
        gxt, gft, gst = xt.cpu().numpy(), ot["f"].cpu().numpy(), ot["status"].cpu().numpy()
        ferr_t = np.abs(gft - rt["f"]) / np.maximum(np.abs(rt["f"]), 1e-10)
        xerr_t = np.linalg.norm(gxt - rt["x"], axis=1) / np.maximum(1.0, np.linalg.norm(rt["x"], axis=1))
        both = (gst == 0) & (rt["status"] == 0)
        xstar = bh / dh
        nrm = np.maximum(1.0, np.linalg.norm(xstar, axis=1))
        bar = {"family": "the problems of the benched batch with kappa <= 100", "problems": NB, "precision": tp,
               "gpu_met_the_gradient_test_fraction": float((gst == 0).mean()),
               "cpu_met_the_gradient_test_fraction": float((rt["status"] == 0).mean()),
               "both_met_the_gradient_test_fraction": float(both.mean()),
               "f_rel_err_max": float(ferr_t.max()),
               "x_err_max_where_both_met_the_gradient_test": float(xerr_t[both].max()) if both.any() else None,
               "x_within_1e-8_fraction_where_both_met_the_gradient_test": float((xerr_t[both] <= 1e-8).mean()) if both.any() else None,
               "x_err_max_all": float(xerr_t.max()),
               "gpu_to_exact_minimiser_err_max": float((np.linalg.norm(gxt - xstar, axis=1) / nrm).max()),
               "cpu_reference_order_to_exact_minimiser_err_max": float((np.linalg.norm(rt["x"] - xstar, axis=1) / nrm).max())}
        need = max(4, NB // 50)
        gpu_exact = bar["gpu_to_exact_minimiser_err_max"]
        bar["comparable_problems"] = int(both.sum())
        bar["comparable_problems_required"] = need
        bar["rule"] = ("ok_x = at least 2 % of the sub-family is comparable (both sides met the gradient test), every comparable "
                       "minimiser agrees within 1e-8, and no GPU minimiser is farther than 3e-7 -- the reference's own "
                       "reproducibility, SURVEY.md section 6 -- from the exact one b/d.  Why not 95 %: the REFERENCE-ORDER run "
                       "itself meets ||g|| < Precision on cpu_met_the_gradient_test_fraction of these problems only (its "
                       "left-to-right sums of 1024 terms blur the objective differences the line search steers by; the kernels' "
                       "tree sums do not, see gpu_met_...), and at a Precision loose enough for 95 % of it the set ||g|| < "
                       "Precision is itself wider than 1e-8 in x")
        bar["ok_x"] = bool(both.sum() >= need and xerr_t[both].max() <= 1e-8 and gpu_exact <= 3e-7)
        bar["ok_f"] = bool(ferr_t.max() <= 1e-10)
        par["minimiser_bar"] = bar
        par["ok_x"] = bar["ok_x"]
        par["ok"] = bool(par["ok"] and bar["ok_x"] and bar["ok_f"])
        # context: the whole family (kappa up to 1e3) at the same tolerance -- how far is either side from b / d?
        SC = min(S, 256)
        xc = x0[:SC].clone()
        oc = NLO.LBFGS(objective, xc, d[:SC].contiguous(), b[:SC].contiguous(),
                       options=NLO.default_options(NLO.LBFGS_, **dict(opt_kw, Precision=tp)))
        torch.cuda.synchronize()
        dc, bc = d[:SC].cpu().numpy(), b[:SC].cpu().numpy()
        rc = O.solve_batch(O.LBFGS, kind, x0[:SC].cpu().numpy(), d=dc, b=bc,
                           opts=O.defaults(precision=tp, maxit=opts.max_iteration, memory=m), sum_mode=O.SEQ, nthreads=cores)
        xs2 = bc / dc
        nr2 = np.maximum(1.0, np.linalg.norm(xs2, axis=1))
        par["whole_family_context"] = {
            "problems": SC, "precision": tp, "kappa": "[10, 1000]",
            "both_met_the_gradient_test": int(((oc["status"].cpu().numpy() == 0) & (rc["status"] == 0)).sum()),
            "gpu_to_exact_minimiser_err_max": float((np.linalg.norm(xc.cpu().numpy() - xs2, axis=1) / nr2).max()),
            "cpu_reference_order_to_exact_minimiser_err_max": float((np.linalg.norm(rc["x"] - xs2, axis=1) / nr2).max()),
            "note": "with kappa up to 1e3 an objective-value line search in fp64 stalls before ||g|| reaches this tolerance and "
                    "both sides stop on MinStepLength / MaxIteration: the reference's own answer is then defined only as "
                    "sharply as these two distances to the exact minimiser b/d show"}
    res["parity"] = par
    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]


# ------------------------------------------------------------------------------------------------ the other BASELINE configs
def _quad(ctx, NLO, B, n, klo, khi, seed=SEED):
    torch = ctx.torch
    d = torch.empty(B, n, dtype=torch.float64, device=ctx.dev)
    b = torch.empty(B, n, dtype=torch.float64, device=ctx.dev)
    NLO.synth_diag_spectrum(seed, d, klo, khi)
    NLO.synth_uniform(seed, b, -1.0, 1.0)
    return d, b


def _shard(ctx, B):
    """this rank's problems of a global batch of B (interleaved) -- all of them at N = 1"""
    from FortranLibrary import distributed as D
    if ctx.world == 1:
        return None
    return D.shard_indices(B, ctx.rank, ctx.world, True).to(ctx.dev)


def _f_parity(gf, rf, floor=1e-10):
    import numpy as np
    err = np.abs(gf - rf) / np.maximum(np.abs(rf), floor)
    return float(err.max()), float((err <= 1e-10).mean())


def config_c2(ctx, NLO, prof, cpu_seconds):
    """BASELINE config 2: batch 4096 independent Rosenbrock n=256, L-BFGS m=10, one GPU"""
    import numpy as np
    torch = ctx.torch
    B, n, m = 4096, 256, 10
    x0 = torch.empty(B, n, dtype=torch.float64, device=ctx.dev)
    NLO.synth_uniform(SEED, x0, 0.9, 1.1)
    ws = NLO.workspace(B, n, m, ctx.dev)
    x = torch.empty_like(x0)
    kw = dict(Precision=1e-10, MaxIteration=3000, Memory=m)

    def run():
        x.copy_(x0)
        return NLO.LBFGS(NLO.ROSENBROCK, x, workspace_=ws, **kw)
    out, ms = timed_launches(ctx, run, 1 if prof else 5, warm=0 if prof else 1)
    it = out["iters"].to(torch.int64)
    wl = {"config": "c2", "batch": B, "n": n, "memory": m, "precision": 1e-10}
    rec, stale = pmc_record("c2", wl)
    res = {"workload": "C2: L-BFGS m=10, chained Rosenbrock n=256, x0 = 1 + 0.1 u, batch 4096, Precision 1e-10", "ms": ms,
           "metric": "lbfgs_iterations_per_sec", "value": float(it.sum()) / ms * 1e3, "unit": "iterations/s",
           "iterations": int(it.sum()), "trials": int(out["nf"].to(torch.int64).sum()),
           "converged_fraction": float((out["status"] == 0).double().mean()),
           "max_abs_x_minus_1": float((x - 1).abs().max()), "pmc_key": wl,
           "roofline": valu_roofline(rec, stale, ms, "fl_solve_kernel<1,4,ROSENBROCK,LBFGS>")}
    if prof or cpu_seconds <= 0:
        return res
    import oracle_lib as O
    oo = O.defaults(precision=1e-10, maxit=3000, memory=m)
    xh = x0.cpu().numpy()
    ref, S, dt = cpu_sample(lambda s: O.solve_batch(O.LBFGS, O.ROSENBROCK, xh[:min(s, B)], opts=oo, nthreads=ctx.cores),
                            4 * ctx.cores, B, cpu_seconds)
    S = min(S, B)
    res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / dt, "unit": "iterations/s", "cores": int(ref["threads"]), "kind": "port",
                           "sample": f"first {S} problems, oracle in reference summation order, {dt:.1f} s wall; host: {cpu_model()}"}
    fmax = float(np.max(np.abs(out["f"][:S].cpu().numpy() - ref["f"][:S])))  # f* = 0: absolute floor 1e-20
    T, E = NLO.reduction_geometry(n)
    SB = 64
    tre = O.solve_batch(O.LBFGS, O.ROSENBROCK, xh[:SB], opts=oo, sum_mode=O.TREE, threads=T, ept=E, nthreads=ctx.cores)
    bit = {"problems": SB, "x": bits_equal(x[:SB].cpu().numpy(), tre["x"]), "f": bits_equal(out["f"][:SB].cpu().numpy(), tre["f"]),
           "iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"]),
           "f_evaluations": bits_equal(out["nf"][:SB].cpu().numpy(), tre["nf"])}
    xerr = np.linalg.norm(x[:S].cpu().numpy() - ref["x"][:S], axis=1) / np.maximum(1.0, np.linalg.norm(ref["x"][:S], axis=1))
    res["parity"] = {"sample": S, "final_f_abs_err_max_vs_reference_order": fmax, "tolerance": {"f_abs_floor": 1e-20, "x": 1e-8},
                     "minimiser_err_max_vs_reference_order": float(xerr.max()),
                     "bit_exact_vs_oracle_kernel_order": bit,
                     "ok": bool(fmax <= 1e-20 and xerr.max() <= 1e-8 and all(v for k, v in bit.items() if k != "problems"))}
    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def config_c3(ctx, NLO, prof, cpu_seconds):
    """BASELINE config 3: batch 65536 convex quadratics n=1024, Dai-Yuan CG, sharded over the GPUs"""
    import numpy as np
    torch = ctx.torch
    Bg, n = 65536, 1024
    dg, bg = _quad(ctx, NLO, Bg, n, 10.0, 1000.0)
    idx = _shard(ctx, Bg)
    d, b = (dg, bg) if idx is None else (dg[idx].contiguous(), bg[idx].contiguous())
    del dg, bg
    B = d.shape[0]
    x = torch.zeros(B, n, dtype=torch.float64, device=ctx.dev)
    kw = dict(Precision=1e-6, MaxIteration=3000)

    def run():
        x.zero_()
        return NLO.ConjugateGradient(NLO.DIAGQUAD, x, d, b, **kw)
    res = _sharded_or_single(ctx, NLO, run, prof, Bg, n, lambda o: {"x": x, "f": o["f"], "iters": o["iters"], "status": o["status"]})
    if res is None:
        return None
    out, ms = res.pop("_out"), res["ms"]
    it = out["iters"].to(torch.int64)
    wl = {"config": "c3", "batch": Bg, "n": n, "precision": 1e-6}
    rec, stale = pmc_record("c3", wl)
    res.update({"workload": "C3: ConjugateGradient (Dai-Yuan), convex diagonal quadratics n=1024, kappa in [10,1000], batch 65536"
                            + (f" sharded over {ctx.world} GPUs (interleaved)" if ctx.world > 1 else "") + ", Precision 1e-6",
                "metric": "cg_iterations_per_sec", "unit": "iterations/s",
                "converged_fraction_rank0": float((out["status"] == 0).double().mean()), "pmc_key": wl})
    if ctx.world == 1:
        res["trials"] = int((out["nf"].to(torch.int64) + out["ng"].to(torch.int64)).sum())
        res["roofline"] = valu_roofline(rec, stale, ms, "fl_solve_kernel<1,16,DIAGQUAD,CG>")
    if prof or cpu_seconds <= 0 or ctx.world > 1:
        return res
    import oracle_lib as O
    oo = O.defaults(precision=1e-6, maxit=3000, c2=0.45)
    SM = 8192
    dh, bh = d[:SM].cpu().numpy(), b[:SM].cpu().numpy()
    ref, S, dt = cpu_sample(lambda s: O.solve_batch(O.CG, O.DIAGQUAD, np.zeros((min(s, SM), n)), d=dh[:min(s, SM)], b=bh[:min(s, SM)],
                                                    opts=oo, nthreads=ctx.cores), 4 * ctx.cores, SM, cpu_seconds)
    S = min(S, SM)
    res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / dt, "unit": "iterations/s", "cores": int(ref["threads"]), "kind": "port",
                           "sample": f"first {S} problems, oracle in reference summation order, {dt:.1f} s wall; host: {cpu_model()}"}
    fmax, ffrac = _f_parity(out["f"][:S].cpu().numpy(), ref["f"][:S])
    T, E = NLO.reduction_geometry(n, NLO.CG)
    SB = 64
    tre = O.solve_batch(O.CG, O.DIAGQUAD, np.zeros((SB, n)), d=dh[:SB], b=bh[:SB], opts=oo, sum_mode=O.TREE, threads=T, ept=E,
                        nthreads=ctx.cores)
    bit = {"problems": SB, "x": bits_equal(x[:SB].cpu().numpy(), tre["x"]), "f": bits_equal(out["f"][:SB].cpu().numpy(), tre["f"]),
           "iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"])}
    res["parity"] = {"sample": S, "final_f_rel_err_max_vs_reference_order": fmax, "f_within_1e-10_fraction": ffrac,
                     "tolerance": {"f_rel": 1e-10}, "bit_exact_vs_oracle_kernel_order": bit,
                     "ok": bool(fmax <= 1e-10 and bit["x"] and bit["f"] and bit["iterations"])}
    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def _sharded_or_single(ctx, NLO, run, prof, Bg, n, results_of):
    """time run() (one launch solving this rank's shard).  N = 1: HIP events, averaged.  N > 1: W = 1 warm-up, then 2
    steps of solve + gather between barriers, max over ranks.  Returns the record on rank 0 (with "_out")."""
    torch = ctx.torch
    if ctx.world == 1 and not ctx.multi:
        out, ms = timed_launches(ctx, run, 1 if prof else 3, warm=0 if prof else 1)
        it = int(out["iters"].to(torch.int64).sum())
        return {"ms": ms, "value": it / ms * 1e3, "iterations": it, "_out": out}
    from FortranLibrary import distributed as D
    out = run()
    shapes = {k: (tuple(v.shape[1:]), v.dtype) for k, v in results_of(out).items()}
    gat = D.Gatherer(Bg, shapes, ctx.dev, dst=0, interleaved=True)
    ctx.sync(gat)
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
        gat.gather(results_of(out), overlap=True)
    ctx.sync(gat)
    dt = ctx.allmax(time.perf_counter() - t0)
    its = [int(v) for v in ctx.allgather_scalar(int(out["iters"].to(torch.int64).sum().item()), torch.int64)]
    if ctx.rank != 0:
        return None
    return {"ms": dt / steps * 1e3, "value": sum(its) * steps / dt, "iterations": sum(its), "iterations_per_rank": its,
            "n_gpus": ctx.world, "scaling": "strong", "steps": steps, "_out": out}


def config_c4(ctx, NLO, prof, cpu_seconds):
    """BASELINE config 4: batch 1024 problems n=4096, full dense BFGS, a fixed 20 iterations, one GPU"""
    import numpy as np
    torch = ctx.torch
    B, n, K = 1024, 4096, 20
    d, b = _quad(ctx, NLO, B, n, 10.0, 100.0)
    x = torch.zeros(B, n, dtype=torch.float64, device=ctx.dev)
    ws = NLO.bfgs_workspace(B, n, ctx.dev)
    kw = dict(Precision=1e-12, MaxIteration=K - 1, ExactStep=0)

    def run():
        x.zero_()
        return NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, **kw)
    out, ms = timed_launches(ctx, run, 1 if prof else 2, warm=0 if prof else 1)
    it = out["iters"].to(torch.int64)
    upd = torch.clamp(it - 1, min=0)
    # SURVEY 8d, rank-2 form: per update 8n^2 (q = H y) + 16n^2 (read H, write H'); the first update only writes 8n^2
    algo = float((upd * 24 * n * n + (it > 0) * 8 * n * n).sum())
    # what the deferred form moves (n > 1024): one 8 n^2 read pass per update after the first J = 8 (those run on the implicit
    # H = a I), the first fold writes 8 n^2, every later fold reads and writes 16 n^2
    J = 8
    folds = torch.div(upd, J, rounding_mode="floor")
    model = float((torch.clamp(upd - J, min=0) * 8 * n * n + (folds > 0) * 8 * n * n + torch.clamp(folds - 1, min=0) * 16 * n * n).sum())
    wl = {"config": "c4", "batch": B, "n": n, "iterations": K}
    rec, stale = pmc_record("c4", wl)
    traffic = float(rec["traffic_bytes_per_launch"]) if rec else None
    moved = traffic if traffic is not None else (model * float(stale["traffic_bytes_per_launch"]) / float(stale["model_bytes_per_launch"])
                                                 if stale and stale.get("model_bytes_per_launch") else model)
    src = "pmc traffic" if rec else ("byte model x the traffic / model ratio of a PMC record taken from other kernel sources" if stale else "byte model of the deferred rank-2 form")
    ach = moved / (ms * 1e-3) / 1e9
    res = {"workload": f"C4: dense BFGS (ExactStep=0), diagonal quadratics n=4096, kappa in [10,100], batch {B}, a fixed {K} iterations",
           "ms": ms, "metric": "bfgs_iterations_per_sec", "value": float(it.sum()) / ms * 1e3, "unit": "iterations/s",
           "iterations": int(it.sum()), "seconds_per_solve": ms * 1e-3, "inverse_hessian_bytes": B * n * n * 8,
           "update_form": "rank-2, deferred: updates kept as vectors, folded into H every 8th iteration (DESIGN.md 4.2)", "pmc_key": wl,
           "roofline": {"bound": "hbm", "kernel": "fl_solve_kernel<8,8,DIAGQUAD,BFGS,0,0>", "achieved": ach, "achieved_source": src,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": ms,
                        "model_bytes_per_launch": model, "traffic_over_model": (traffic / model) if traffic else None,
                        "algorithmic_bytes_per_launch": algo, "algorithmic_bw": algo / (ms * 1e-3) / 1e9,
                        "note": "algorithmic = SURVEY 8d's 24 n^2 B per update of the immediate rank-2 form (a rate: the deferred "
                                "form moves a third of it); model = bytes of the deferred form; traffic = PMC",
                        "source": rec.get("source") if rec else None}}
    if prof or cpu_seconds <= 0:
        return res
    import oracle_lib as O
    Bc = 2 * ctx.cores
    oo = O.defaults(precision=1e-12, maxit=K - 1, exact_step=0)
    dh, bh = d[:Bc].cpu().numpy(), b[:Bc].cpu().numpy()
    t = time.perf_counter()
    ref = O.solve_batch(O.BFGS, O.DIAGQUAD, np.zeros((Bc, n)), d=dh, b=bh, opts=oo, bfgs_form=1, nthreads=ctx.cores)
    dt = time.perf_counter() - t
    res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / dt, "unit": "iterations/s", "cores": int(ref["threads"]), "kind": "port",
                           "sample": f"first {Bc} problems, oracle in the O(n^2) rank-2 form (the reference's two n^3 matmuls per "
                                     f"iteration would take ~1000x longer), reference summation order, {dt:.1f} s wall; host: {cpu_model()}"}
    fmax, ffrac = _f_parity(out["f"][:Bc].cpu().numpy(), ref["f"], 1e-300)
    T, E = NLO.reduction_geometry(n)
    SB = min(Bc, 16)
    tre = O.solve_batch(O.BFGS, O.DIAGQUAD, np.zeros((SB, n)), d=dh[:SB], b=bh[:SB], opts=oo, bfgs_form=108, sum_mode=O.TREE,
                        threads=T, ept=E, nthreads=ctx.cores)
    bit = {"problems": SB, "x": bits_equal(x[:SB].cpu().numpy(), tre["x"]), "f": bits_equal(out["f"][:SB].cpu().numpy(), tre["f"]),
           "iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"])}
    res["parity"] = {"sample": Bc, "final_f_rel_err_max_vs_reference_order": fmax, "tolerance": {"f_rel": 1e-10},
                     "bit_exact_vs_oracle_kernel_order": bit, "ok": bool(fmax <= 1e-10 and bit["x"] and bit["f"] and bit["iterations"])}
    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def config_c4gemm(ctx, NLO, prof, cpu_seconds):
    """BASELINE config 4's update as the reference writes it: H <- U^T (H U) + rho s s^T, two n^3 products on the f64 matrix cores"""
    import ctypes as C
    import numpy as np
    torch = ctx.torch
    B, n = 16, 4096
    T, E = NLO.reduction_geometry(n)
    ld = T * E
    gen = torch.Generator(device=ctx.dev)
    gen.manual_seed(SEED)
    H = torch.zeros(B, n, ld, dtype=torch.float64, device=ctx.dev)
    H[:, torch.arange(n), torch.arange(n)] = 1.0
    H[:, :, :n] += 0.01 * torch.randn(B, n, n, dtype=torch.float64, device=ctx.dev, generator=gen)
    s = torch.randn(B, n, dtype=torch.float64, device=ctx.dev, generator=gen)
    y = s * (1 + torch.rand(B, n, dtype=torch.float64, device=ctx.dev, generator=gen))
    H0 = H[:1, :1024, :1024].clone() if not prof else None
    ws = NLO.bfgs_update_gemm(H, s, y)
    _, ms = timed_launches(ctx, lambda: NLO.bfgs_update_gemm(H, s, y, workspace_=ws), 1 if prof else 3, warm=0 if prof else 1)
    flop = 4.0 * n ** 3 * B
    tf = flop / ms / 1e9
    res = {"workload": f"C4_gemm: the BFGS update as written (NO.f90:958-962: two n^3 matmuls), n=4096, {B} problems per call "
                       "(fl_bfgs_update_gemm_batched, v_mfma_f64_16x16x4_f64)", "ms": ms,
           "metric": "bfgs_updates_per_sec", "value": B / ms * 1e3, "unit": "updates/s", "ms_per_update": ms / B,
           "roofline": {"bound": "mfma", "kernel": "bfgs_gemm_kernel<2,4> (two launches per call: H U, then U^T (H U) + rho s s^T)",
                        "achieved": tf, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TF, "traffic": None,
                        "flops_per_call": flop, "kernel_ms": ms,
                        "note": "4 n^3 flop per update / HIP-event time of the call (both product launches and the fragment setup)"}}
    if prof or cpu_seconds <= 0:
        return res
    import oracle_lib as O
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_bfgs_update.argtypes = [C.c_int, dp, dp, dp, C.c_int]
    lib.flo_set_sum_mode(O.SEQ, 64, 2)
    # bounded CPU sample of the same update: n = 1024 (1/64 of the flops), the oracle's two sequential matmuls, one core
    nc = 1024
    Hc = np.ascontiguousarray(H0[0].cpu().numpy())
    sc, yc = s[0, :nc].cpu().numpy().copy(), y[0, :nc].cpu().numpy().copy()
    ref = Hc.copy()
    t = time.perf_counter()
    lib.flo_bfgs_update(nc, ref.ctypes.data_as(dp), sc.ctypes.data_as(dp), yc.ctypes.data_as(dp), 0)
    dt = time.perf_counter() - t
    cpu_tf = 4.0 * nc ** 3 / dt / 1e12
    res["cpu_baseline"] = {"value": cpu_tf, "unit": "TFLOP/s", "cores": 1, "kind": "port",
                           "sample": f"one update at n = {nc} (1/64 of the flops of n = 4096): the oracle's two sequential matmuls "
                                     f"(NO.f90:958-962 as written), {dt:.1f} s wall; host: {cpu_model()}"}
    # parity: the same n = 1024 update on the GPU (MFMA summation order: tolerance relative to the products' magnitude)
    T1, E1 = NLO.reduction_geometry(nc)
    Hg = torch.zeros(1, nc, T1 * E1, dtype=torch.float64, device=ctx.dev)
    Hg[0, :, :nc] = H0[0]
    NLO.bfgs_update_gemm(Hg, s[:1, :nc].contiguous(), y[:1, :nc].contiguous())
    torch.cuda.synchronize()
    err = float(np.max(np.abs(Hg[0, :, :nc].cpu().numpy() - ref)) / (np.abs(ref).max() * nc))
    res["parity"] = {"n": nc, "max_abs_err_over_max_abs_times_n": err, "tolerance": 1e-12,
                     "note": "matrix cores accumulate with fused multiply-adds in their own k order: tolerance, not bits "
                             "(tests/test_gpu_bfgs_gemm.py holds n = 4096 to a committed oracle digest)", "ok": bool(err <= 1e-12)}
    res["speedup_vs_cpu_baseline"] = tf / cpu_tf
    return res


def config_c5(ctx, NLO, prof, cpu_seconds):
    """BASELINE config 5: augmented Lagrangian around L-BFGS, batch 8192, n=512, 8 equality constraints, sharded over the GPUs"""
    import numpy as np
    torch = ctx.torch
    Bg, n, M, m = 8192, 512, 8, 10
    dg, bg = _quad(ctx, NLO, Bg, n, 2.0, 10.0)
    xg = torch.empty(Bg, n, dtype=torch.float64, device=ctx.dev)
    NLO.synth_uniform(SEED + 7, xg, 0.05, 0.15)
    idx = _shard(ctx, Bg)
    d, b, x0 = (dg, bg, xg) if idx is None else (dg[idx].contiguous(), bg[idx].contiguous(), xg[idx].contiguous())
    del dg, bg, xg
    B = d.shape[0]
    x = torch.empty_like(x0)
    ws = NLO.workspace(B, n, m, ctx.dev)

    def run():
        x.copy_(x0)
        return NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, d, b, UnconstrainedSolver="LBFGS", workspace_=ws, Precision=1e-10, Memory=m)
    res = _sharded_or_single(ctx, NLO, run, prof, Bg, n, lambda o: {"x": x, "f": o["f"], "iters": o["iters"], "status": o["status"],
                                                                    "lambda": o["lambda"], "cnorm2": o["cnorm2"]})
    if res is None:
        return None
    out, ms = res.pop("_out"), res["ms"]
    wl = {"config": "c5", "batch": Bg, "n": n, "constraints": M, "memory": m, "precision": 1e-10}
    rec, stale = pmc_record("c5", wl)
    nfp = out["nf"].double()
    res.update({"workload": "C5: AugmentedLagrangian around L-BFGS m=10, diagonal quadratics n=512 (kappa in [2,10]), 8 block-sphere "
                            "equality constraints, batch 8192" + (f" sharded over {ctx.world} GPUs (interleaved)" if ctx.world > 1 else "")
                            + ", Precision 1e-10",
                "metric": "inner_lbfgs_iterations_per_sec", "unit": "iterations/s", "seconds_per_solve": ms * 1e-3,
                "outer_iterations_mean_rank0": float(out["outer"].double().mean()),
                "objective_evaluations_rank0": int(out["nf"].to(torch.int64).sum()),
                "objective_evaluations_per_problem_min_mean_max": [float(nfp.min()), float(nfp.mean()), float(nfp.max())],
                "converged_fraction_rank0": float((out["status"] == 0).double().mean()),
                "constraint_norm_max_rank0": float(out["cnorm2"].max().sqrt()), "pmc_key": wl})
    # how the library ran this rank's share (include/fl_nlopt.h: fl_augmented_lagrangian_launch_plan): helper waves by batch size
    # for the start, the unfinished problems handed to launches with more waves per problem -- results do not depend on it
    plan = NLO.augmented_lagrangian_launch_plan(NLO.LBFGS_, NLO.DIAGQUAD, B, n, M)
    res["launch_plan"] = {"problems_on_this_rank": B, "stages": [{"waves_per_problem": w, "hands_over_when_unfinished_at_most": p} for w, p in plan],
                          "note": "bit-identical to one launch of one wave per problem (tests/test_gpu_helpers.py)"}
    if ctx.world == 1:
        res["roofline"] = valu_roofline(rec, stale, ms, "fl_solve_kernel<1,8,DIAGQUAD,LBFGS,AUG> + fl_solve_rep_kernel<2 / 4 waves> (the stages of one solve)")
        res["roofline"]["note"] = ("the longest problem does ~5x the mean's objective evaluations; the plain kernel runs while the chip is "
                                   "full of problems, the tail moves to launches with 2 and 4 waves per problem (launch_plan); counters = "
                                   "the sum over the stages' kernels")
    if prof or cpu_seconds <= 0 or ctx.world > 1:
        return res
    import oracle_lib as O
    oo = O.defaults(precision=1e-10, memory=m)
    SM = 1024
    xh, dh, bh = x0[:SM].cpu().numpy(), d[:SM].cpu().numpy(), b[:SM].cpu().numpy()
    ref, S, dt = cpu_sample(lambda s: O.auglag_batch(O.LBFGS, O.DIAGQUAD, xh[:min(s, SM)], M, d=dh[:min(s, SM)], b=bh[:min(s, SM)],
                                                     opts=oo, nthreads=ctx.cores), 4 * ctx.cores, SM, cpu_seconds)
    S = min(S, SM)
    res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / dt, "unit": "iterations/s", "cores": int(ref["threads"]), "kind": "port",
                           "sample": f"first {S} problems, oracle in reference summation order, {dt:.1f} s wall; host: {cpu_model()}"}
    fx = (0.5 * (d[:S] * x[:S] * x[:S]).sum(1) - (b[:S] * x[:S]).sum(1)).cpu().numpy()
    fmax, ffrac = _f_parity(fx, ref["f"][:S])
    T, E = NLO.reduction_geometry(n)
    SB = 32
    tre = O.auglag_batch(O.LBFGS, O.DIAGQUAD, xh[:SB], M, d=dh[:SB], b=bh[:SB], opts=oo, sum_mode=O.TREE, threads=T, ept=E,
                         nthreads=ctx.cores)
    bit = {"problems": SB, "x": bits_equal(x[:SB].cpu().numpy(), tre["x"]), "lambda": bits_equal(out["lambda"][:SB].cpu().numpy(), tre["lam"]),
           "inner_iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"]),
           "outer_iterations": bits_equal(out["outer"][:SB].cpu().numpy(), tre["outer"]),
           "objective_evaluations": bits_equal(out["nf"][:SB].cpu().numpy(), tre["nf"])}
    res["parity"] = {"sample": S, "final_f_rel_err_max_vs_reference_order": fmax, "f_within_1e-10_fraction": ffrac,
                     "tolerance": {"f_rel": 1e-10, "constraint_norm": 1e-10}, "bit_exact_vs_oracle_kernel_order": bit,
                     "ok": bool(fmax <= 1e-10 and res["constraint_norm_max_rank0"] <= 1e-10 and all(v for k, v in bit.items() if k != "problems"))}
    res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    return res


def rtc_leg(ctx, NLO):
    """The caller's objective handed over as SOURCE TEXT (fl_user_compile: hiprtc at run time, csrc/fl_user_rtc.hip) against the
    built-in objective on the headline family -- 16 384 problems, L-BFGS m = 10, n = 1024: compile time, solve time, bits.  The
    reference takes the objective as callbacks (NO.f90:33-38); reverse communication (fl_rci_*) is the other way to pass one."""
    torch = ctx.torch
    try:
        import user_sources as US
    except ImportError:
        return {"skipped": "tests/user_sources.py not found"}
    B, n, m = 16384, 1024, 10
    d, b = _quad(ctx, NLO, B, n, 10.0, 1000.0)
    x = torch.zeros(B, n, dtype=torch.float64, device=ctx.dev)
    ws = NLO.workspace(B, n, m, ctx.dev)
    t = time.perf_counter()
    obj = NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", n, solver=NLO.LBFGS_, tune_like=NLO.DIAGQUAD)
    compile_s = time.perf_counter() - t

    def builtin():
        x.zero_()
        return NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)

    def compiled():
        x.zero_()
        return obj.solve(x, d, b, None, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)
    ob, mb = timed_launches(ctx, builtin, 3)
    xb = x.clone()
    ou, mu = timed_launches(ctx, compiled, 3)
    same = bool(torch.equal(x, xb) and all(torch.equal(ou[k], ob[k]) for k in ("f", "iters", "nf", "ng", "status")))
    it = float(ou["iters"].to(torch.int64).sum())
    return {"workload": f"L-BFGS m={m}, diagonal quadratics n={n}, {B} problems; the objective given as HIP source text", "compile_seconds": compile_s,
            "ms_compiled": mu, "ms_builtin": mb, "iterations_per_s_compiled": it / mu * 1e3, "speed_vs_builtin": mb / mu,
            "same_bits_as_builtin": same, "note": "timed with the memset of x inside (both legs alike)"}


def bfgs_mid_leg(ctx, NLO, cpu_seconds):
    """BFGS -- the reference's DEFAULT solver (NO.f90:632-1022) -- at a size its users have: n = 1024, 4096 problems, ExactStep = 0,
    a fixed 20 iterations like BASELINE config 4.  Since round 4 the fused kernels keep 8 rank-2 updates pending and fold them
    into H every 8th iteration for every n > 128 (fl_bfgs_deferred_updates): 10 n^2 bytes per iteration in the steady state where
    the update applied at once moves 24 n^2.  Parity: the oracle's deferred form in the kernel's summation order, bit for bit."""
    import numpy as np
    torch = ctx.torch
    B, n, K = 4096, 1024, 20
    d, b = _quad(ctx, NLO, B, n, 10.0, 100.0)
    x = torch.zeros(B, n, dtype=torch.float64, device=ctx.dev)
    ws = NLO.bfgs_workspace(B, n, ctx.dev)
    kw = dict(Precision=1e-12, MaxIteration=K - 1, ExactStep=0)

    def run():
        x.zero_()
        return NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, **kw)
    out, ms = timed_launches(ctx, run, 3)
    it = out["iters"].to(torch.int64)
    upd = torch.clamp(it - 1, min=0)
    J = NLO.bfgs_deferred_updates(n)
    folds = torch.div(upd, J, rounding_mode="floor")
    moved = float((torch.clamp(upd - J, min=0) * 8 * n * n + (folds > 0) * 8 * n * n + torch.clamp(folds - 1, min=0) * 16 * n * n).sum())
    algo = float((upd * 24 * n * n + (it > 0) * 8 * n * n).sum())
    r = {"workload": f"BFGS (ExactStep=0), diagonal quadratics n={n}, kappa in [10,100], batch {B}, a fixed {K} iterations",
         "ms": ms, "iterations_per_s": float(it.sum()) / ms * 1e3, "updates_kept_pending": J,
         "roofline": {"bound": "hbm", "kernel": "fl_solve_kernel<2,8,DIAGQUAD,BFGS,0,0>", "achieved": moved / ms / 1e6, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": moved / ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "achieved_source": "byte model of the deferred rank-2 form",
                      "algorithmic_bytes_per_launch": algo, "algorithmic_bw": algo / ms / 1e6,
                      "note": "algorithmic = 24 n^2 B per update of the form applied at once (rounds 1-3 at this n: 385 ms)"}}
    if cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        Bc = 2 * ctx.cores
        oo = O.defaults(precision=1e-12, maxit=K - 1, exact_step=0)
        dh, bh = d[:Bc].cpu().numpy(), b[:Bc].cpu().numpy()
        t = time.perf_counter()
        ref = O.solve_batch(O.BFGS, O.DIAGQUAD, np.zeros((Bc, n)), d=dh, b=bh, opts=oo, bfgs_form=1, nthreads=ctx.cores)
        dt = time.perf_counter() - t
        r["cpu_baseline"] = {"value": float(ref["iters"].sum()) / dt, "unit": "iterations/s", "cores": int(ref["threads"]), "kind": "port",
                             "sample": f"first {Bc} problems, oracle in the O(n^2) rank-2 form, reference summation order, {dt:.2f} s wall"}
        T, E = NLO.reduction_geometry(n)
        SB = 8
        tre = O.solve_batch(O.BFGS, O.DIAGQUAD, np.zeros((SB, n)), d=dh[:SB], b=bh[:SB], opts=oo, bfgs_form=100 + J, sum_mode=O.TREE, threads=T,
                            ept=E, nthreads=ctx.cores)
        fmax, _ = _f_parity(out["f"][:Bc].cpu().numpy(), ref["f"], 1e-300)
        bit = {"problems": SB, "x": bits_equal(x[:SB].cpu().numpy(), tre["x"]), "f": bits_equal(out["f"][:SB].cpu().numpy(), tre["f"]),
               "iterations": bits_equal(out["iters"][:SB].cpu().numpy(), tre["iters"])}
        r["parity"] = {"final_f_rel_err_max_vs_reference_order": fmax, "bit_exact_vs_oracle_kernel_order": bit,
                       "ok": bool(fmax <= 1e-10 and bit["x"] and bit["f"] and bit["iterations"])}
        r["speedup_vs_cpu_baseline"] = r["iterations_per_s"] / r["cpu_baseline"]["value"]
    return r


def one_problem_leg(ctx, NLO, cpu_seconds):
    """The reference's typical call -- ONE problem (LBFGS takes one x: NO.f90:398-625) -- at n = 2^20: the fused solve shares the
    problem among `groups` workgroups (csrc/fl_big.hpp, the cooperative form: one launch, a trial at many CUs' bandwidth plus a
    barrier) against one workgroup (FL_COOP_GROUPS=1: one CU).  Parity: the oracle in the kernel's order with `groups`, bit for bit."""
    import numpy as np
    torch = ctx.torch
    n, m = 1 << 20, 10
    d, b = _quad(ctx, NLO, 1, n, 10.0, 100.0)
    # (Precision 1e-4: |grad f|^2 < 1e-8 from 3.5e5 at x = 0.  1e-6 would ask for 1e-12 -- below the rounding noise of sums of
    # 2^20 terms around f = -4e4, where it is luck which summation order gets through the last line searches: DESIGN.md 4.5b)
    kw = dict(Precision=1e-4, MaxIteration=60, Memory=m)
    ws = NLO.workspace(1, n, m, ctx.dev)
    x = torch.zeros(1, n, dtype=torch.float64, device=ctx.dev)

    def solve():
        x.zero_()
        return NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, **kw)
    keep = os.environ.get("FL_COOP_GROUPS")
    try:
        os.environ.pop("FL_COOP_GROUPS", None)
        G = NLO.cooperative_groups(NLO.LBFGS_, NLO.DIAGQUAD, 1, n)
        out, ms = timed_launches(ctx, solve, 3)
        xg = x.cpu().numpy().copy()
        os.environ["FL_COOP_GROUPS"] = "1"
        out1, ms1 = timed_launches(ctx, solve, 1)
    finally:
        if keep is None:
            os.environ.pop("FL_COOP_GROUPS", None)
        else:
            os.environ["FL_COOP_GROUPS"] = keep
    nf, it = int(out["nf"][0]), int(out["iters"][0])
    # bytes one evaluation moves (x0, p read; x, g written; d, b read) and one two-loop recursion (2m pairs read twice, p written)
    traffic = nf * 6 * n * 8 + it * (4 * m + 2) * n * 8
    r = {"workload": f"ONE problem, L-BFGS m={m}, diagonal quadratic n=2^20 (kappa in [10,100]), Precision 1e-4, at most 60 iterations",
         "workgroups_per_problem": G, "ms": ms, "ms_one_workgroup": ms1, "speedup": ms1 / ms, "iterations": it, "objective_evaluations": nf,
         "status": int(out["status"][0]), "vector_traffic_GBps": traffic / ms / 1e6,
         "us_per_reduction_step": ms * 1e3 / max(nf + it * (2 * m + 2), 1),
         "note": "the sums' order -- and the last bits of x -- depend on workgroups_per_problem (fl_cooperative_groups_for reports it)"}
    if cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        T, E = NLO.reduction_geometry(n, NLO.LBFGS_)
        oo = O.defaults(precision=1e-4, maxit=60)
        oo.memory = m
        O.lib().flo_set_sum_groups(G)
        t = time.perf_counter()
        try:
            o = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((1, n)), d=d.cpu().numpy(), b=b.cpu().numpy(), opts=oo, sum_mode=O.TREE, threads=T,
                              ept=E, nthreads=1)
        finally:
            O.lib().flo_set_sum_groups(1)
        r["cpu_oracle_seconds_one_core"] = time.perf_counter() - t
        r["parity"] = {"bit_identical_x": bool(np.array_equal(xg.view(np.uint64), o["x"].view(np.uint64))),
                       "objective_evaluations": bool(int(o["nf"][0]) == nf), "iterations": bool(int(o["iters"][0]) == it)}
        r["parity"]["ok"] = all(r["parity"].values())
    return r


CONFIGS = {"c2": config_c2, "c3": config_c3, "c4": config_c4, "c4gemm": config_c4gemm, "c5": config_c5}
MULTI_GPU_CONFIGS = ("c3", "c5")  # BASELINE.json: "sharded 1/2/4/8 GPUs", "8xMI355X"


def run_configs(ctx, NLO, names, prof, cpu_seconds):
    torch = ctx.torch
    out = {}
    for name in names:
        if ctx.world > 1 and name not in MULTI_GPU_CONFIGS:
            continue
        t = time.perf_counter()
        r = CONFIGS[name](ctx, NLO, prof, cpu_seconds)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if r is not None:
            r["wall_s"] = time.perf_counter() - t
            out[{"c4gemm": "C4_gemm"}.get(name, name.upper())] = r
    return out


# ------------------------------------------------------------------------------------------------ main
def spawn_launcher_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (nothing has touched
    the GPU yet -- no exec after GPU initialisation, ever) and leave with its exit code"""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU (weak) / in all (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--contiguous", action="store_true", help="strong scaling: contiguous blocks instead of problem k -> rank k mod N")
    ap.add_argument("--interleaved", action="store_true", help="(default for strong scaling; kept for older command lines)")
    ap.add_argument("--single-leg", action="store_true", help="N > 1: only the leg --scaling names")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--memory", type=int, default=10)
    ap.add_argument("--workload", default="lbfgs_quad1024", choices=["lbfgs_quad1024", "lbfgs_rosen256"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="problems timed on the CPU (-1 auto, 0 skip)")
    ap.add_argument("--no-two-loop", action="store_true")
    ap.add_argument("--configs", default="all", help="other BASELINE configurations to run after the headline: all | none | c2,c3,...")
    ap.add_argument("--config-cpu-seconds", type=float, default=3.0, help="CPU sample per configuration (0: no CPU leg)")
    ap.add_argument("--only-config", default=None, choices=list(CONFIG_KEYS), help="run ONE configuration alone (profiling passes)")
    ap.add_argument("--profile", action="store_true", help="with --only-config: one launch, no warm-up, no CPU leg")
    ap.add_argument("--bar-precision", type=float, default=3e-8, help="gradient tolerance of parity.minimiser_bar")
    ap.add_argument("--bar-problems", type=int, default=4096)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI); gloo only to "
                    "rehearse the N > 1 orchestration with several ranks on ONE GPU (FL_BENCH_ONE_DEVICE=1)")
    args = ap.parse_args()
    spawn_launcher_if_needed(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch  # (before libFL.so: both bring a HIP runtime along, and torch's has to be the one the process loads first)
    import FortranLibrary.NonlinearOptimization as NLO
    ctx = Ctx(args)

    if args.only_config:
        r = CONFIGS[args.only_config](ctx, NLO, args.profile, 0.0 if args.profile else args.config_cpu_seconds)
        if ctx.rank == 0:
            print(json.dumps({"only_config": args.only_config, **(r or {})}))
            sys.stdout.flush()
        ctx.close()
        return

    interleaved = not args.contiguous
    res, st = headline_leg(ctx, args, NLO, args.scaling, interleaved, full_report=True)
    other = None
    if ctx.world > 1 and not args.single_leg:
        # release the first leg's buffers on every rank before the second one allocates its own
        keep = st
        oth = "strong" if args.scaling == "weak" else "weak"
        other, _ = headline_leg(ctx, args, NLO, oth, interleaved, full_report=False)
        st = keep
    if ctx.rank == 0:
        res["roofline"] = headline_roofline(ctx, args, NLO, st)
        if other is not None:
            res["other_leg"] = other
        if not args.no_two_loop:
            res["two_loop"] = headline_two_loop(ctx, NLO, st)
        if args.cpu_sample != 0 and ctx.world == 1:  # reported at N = 1 only (the other ranks would wait at the barrier)
            headline_cpu_and_parity(ctx, args, NLO, st, res)
    st = None
    torch.cuda.empty_cache()

    names = [] if args.configs == "none" else (list(CONFIG_KEYS) if args.configs == "all" else [c.strip().lower() for c in args.configs.split(",") if c.strip()])
    if args.workload != "lbfgs_quad1024":
        names = []
    if names:
        cfgs = run_configs(ctx, NLO, names, False, args.config_cpu_seconds)
        if ctx.rank == 0:
            res["configs"] = cfgs

    if names and ctx.world == 1 and ctx.rank == 0:
        res["user_objective_compiled_at_run_time"] = rtc_leg(ctx, NLO)
        res["one_problem_of_a_million_unknowns"] = one_problem_leg(ctx, NLO, args.config_cpu_seconds)
        res["bfgs_default_solver_n1024"] = bfgs_mid_leg(ctx, NLO, args.config_cpu_seconds)

    bad = False
    if ctx.rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
        bad = ("parity" in res and not res["parity"]["ok"]) or any(
            "parity" in c and not c["parity"]["ok"] for c in res.get("configs", {}).values())
    ctx.close()
    if bad:
        sys.stderr.write("bench.py: PARITY VIOLATION (objective / minimiser tolerance or bit-exactness against the oracle)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
