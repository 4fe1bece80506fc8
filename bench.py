#!/usr/bin/env python3
"""bench.py -- batched L-BFGS on MI355X: iterations/sec, HBM roofline, CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
by torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one complete pass of the hot path over one batch: every problem of the batch is
solved from its initial guess to convergence by ONE launch of the fused solver kernel
(fl_lbfgs_batched), then -- when N > 1 -- the converged results (x*, f*, iterations, status)
are gathered to rank 0 with one RCCL gather per array (buffers allocated before the timed region).
The workload is the north-star configuration: L-BFGS, Memory = 10, n = 1024 convex diagonal
quadratics of BASELINE.json's config-3 family (kappa log-uniform in [10, 1000], b ~ U(-1,1),
x0 = 0, Precision = 1e-6: the tightest gradient tolerance every problem of this family can meet with
an fp64 objective-value line search), inputs generated on the device (Philox, stream = global
problem id) and resident in HBM before the timed region.
  --scaling weak   (default) 65536 problems PER GPU;
  --scaling strong --batch problems in ALL, rank r owns a contiguous block (or, --interleaved,
                   the problems k = r mod N) -- BASELINE config 3 "sharded 1/2/4/8 GPUs".
value = L-BFGS iterations (line searches) of all ranks / sec.

roofline (the fused solver kernel is the only kernel in the timed region; its duration comes from
HIP events on the launch stream):
  * traffic   = bytes through the L2's memory side per launch, (2*FETCH_SIZE + WRITE_SIZE) KiB from
                separate rocprofv3 --pmc passes of this very workload (MI355X_MICROARCH.md, HBM:
                FETCH_SIZE counts 64 of the 128 bytes of a wide read on gfx950); taken from
                profiles/traffic.json only while its record is of the same kernel sources and workload;
  * achieved  = traffic / kernel time (when there is no valid record: the model below / kernel time),
                frac = achieved / 8 TB/s -- a bound that binds: <= 1 by construction;
  * model     = the bytes the shipped kernel asks the L2 for: with C pairs of the ring on the chip
                (fl_lbfgs_onchip_pairs) an iteration loads max(0,2(cnt-C)) + max(0,2(cnt-C-2)) rows and
                stores 2.  traffic_over_model is what of it crossed to the memory side: 1 - (L2 hits), the rows
                around the turn-around of the recursion being re-read within a few microseconds; > 1 would be
                re-fetching;
  * algorithmic_bw = SURVEY.md 8(d)'s streaming figure, (4*cnt+2)*8n per iteration and problem,
                / kernel time: what a kernel WITHOUT on-chip reuse would have to move at this speed.  It may
                exceed the HBM peak -- the rows served from registers / LDS never cross the pins -- and is
                therefore reported as a rate, not as a fraction of the roofline;
  * trial_phase = the other limiter: strong-Wolfe trials per iteration (live), and from the PMC record
                the vector instructions and wave cycles per trial.
"two_loop" reports the stand-alone two-loop kernel (4m rows fetched + 1 read + 1 written).
cpu_baseline: the CPU oracle (reference summation order), OpenMP over problems on all host cores, on
a bounded sample of rank 0's batch; parity: the same sample against the GPU result.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
SEED = 20261003


def kernel_source_hash():
    """identifies the kernel a PMC record was taken from: the sources the fused solver kernel is compiled from"""
    import hashlib
    h = hashlib.sha256()
    for f in ("fl_device.hpp", "fl_reduce.hpp", "fl_linesearch.hpp", "fl_dense.hpp", "fl_solver_launch.hpp"):  # (device code only)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU (weak) / in all (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--interleaved", action="store_true", help="strong scaling: problem k -> rank k mod N")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--memory", type=int, default=10)
    ap.add_argument("--workload", default="lbfgs_quad1024", choices=["lbfgs_quad1024", "lbfgs_rosen256"])
    ap.add_argument("--cpu-sample", type=int, default=-1, help="problems timed on the CPU (-1 auto, 0 skip)")
    ap.add_argument("--no-two-loop", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL over xGMI); gloo only to "
                    "rehearse the N > 1 orchestration with several ranks on ONE GPU (FL_BENCH_ONE_DEVICE=1)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import FortranLibrary.NonlinearOptimization as NLO

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: for N > 1 launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}")
    if os.environ.get("FL_BENCH_ONE_DEVICE"):  # rehearsal: every rank on GPU 0 (needs --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # FL_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, gatherer, collectives of the report) with a
    # world of one -- the only way to run that path through RCCL on a one-GPU box (tests/test_gpu_bench.py)
    multi = world > 1 or bool(os.environ.get("FL_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.workload == "lbfgs_rosen256":
        args.n, objective = 256, NLO.ROSENBROCK
        if args.batch == 65536:
            args.batch = 4096
        opt_kw = dict(Precision=1e-10, MaxIteration=3000, Memory=args.memory)
    else:
        objective = NLO.DIAGQUAD
        opt_kw = dict(Precision=1e-6, MaxIteration=3000, Memory=args.memory)
    n, m = args.n, args.memory
    from FortranLibrary import distributed as D
    # ---- who owns what.  weak: every rank its own args.batch problems; strong: args.batch problems in all
    if args.scaling == "strong":
        Bglobal = args.batch
        idx = D.shard_indices(Bglobal, rank, world, args.interleaved).to(dev)
        B = int(idx.numel())
    else:
        Bglobal, idx, B = args.batch * world, None, args.batch

    # ---- synthetic inputs, generated on the device, resident in HBM
    def synth(Bgen, seed):
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
    if multi:  # the exchange's buffers exist before the timed region
        gat = D.Gatherer(Bglobal, {"x": ((n,), torch.float64), "f": ((), torch.float64), "iters": ((), torch.int32),
                                   "status": ((), torch.int32)}, dev, dst=0,
                         interleaved=(args.scaling == "strong" and args.interleaved))

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

    def sync():
        if gat is not None:
            gat.finish()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    sync()
    dt = time.perf_counter() - t0
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    my_iters = out["iters"].to(torch.int64).sum().reshape(1).to(cdev)
    my_ms = torch.tensor([sum(a.elapsed_time(bb) for a, bb in ev) / max(1, len(ev))], dtype=torch.float64, device=cdev)
    rank_iters, rank_ms = [my_iters], [my_ms]
    gather_ms = None
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        rank_iters = [torch.zeros_like(my_iters) for _ in range(world)]
        rank_ms = [torch.zeros_like(my_ms) for _ in range(world)]
        dist.all_gather(rank_iters, my_iters)
        dist.all_gather(rank_ms, my_ms)
        # the exchange alone, outside the timed region: barrier, gather, synchronise
        sync()
        tg = time.perf_counter()
        gat.gather({"x": x, "f": out["f"], "iters": out["iters"], "status": out["status"]})
        sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
    dt = float(tmax.item())
    per_rank_iters = [int(t.item()) for t in rank_iters]
    per_rank_kernel_ms = [float(t.item()) for t in rank_ms]
    total_iters_per_step = sum(per_rank_iters)

    if rank != 0:
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- rank 0: roofline of the solver kernel (HIP events on the launch stream)
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
    kern_ms = sum(a.elapsed_time(bb) for a, bb in ev) / len(ev)
    status = out["status"].cpu().numpy()
    trials = int(out["nf"].to(torch.int64).sum().item())
    wl_key = {"workload": args.workload, "batch_per_gpu": B, "n": n, "memory": m, "precision": opts.precision}
    pmc = None  # HBM-side bytes and SQ counters per launch, recorded from separate rocprofv3 passes (tools/pmc_summary.py)
    stale = None  # a record of the same workload taken from other kernel sources: only its traffic / model ratio is used
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
        if rec and all(rec.get(kk) == vv for kk, vv in wl_key.items()):
            if rec.get("kernel_source_hash") == kernel_source_hash():
                pmc = rec
            else:
                stale = rec
    except (OSError, ValueError, KeyError):
        pass
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
    trial_phase = {"trials_per_launch": trials, "trials_per_iteration": trials / max(1, int(it.sum().item())),
                   "kernel_us_per_iteration_and_resident_problem": None}
    if pmc and pmc.get("SQ_INSTS_VALU"):
        # SQ_* are sums over all waves; a trial is executed by every wave of its problem
        trial_phase["valu_insts_per_wave_and_trial_upper_bound"] = pmc["SQ_INSTS_VALU"] / (pmc["trials_per_launch"] * waves_per_problem)
        trial_phase["wave_cycles_per_trial_upper_bound"] = 4.0 * pmc["SQ_WAVE_CYCLES"] / (pmc["trials_per_launch"] * waves_per_problem)
        trial_phase["valu_busy_fraction"] = pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_ACTIVE_INST_VALU") else None
        trial_phase["waiting_fraction"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAIT_ANY") else None
        trial_phase["note"] = ("upper bounds: all vector instructions / wave cycles of the launch divided by its trials "
                               "(the two-loop recursion's share is included); source: " + ", ".join(pmc.get("source", [])))

    scaling_note = ("weak: %d problems per GPU" % B) if args.scaling == "weak" else (
        "strong: %d problems in all, %s blocks" % (Bglobal, "interleaved" if args.interleaved else "contiguous"))
    res = {
        "metric": "lbfgs_iterations_per_sec",
        "value": total_iters_per_step * args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"L-BFGS m={m}, {scaling_note}, n={n}, "
                               + ("convex diagonal quadratics kappa in [10,1000], Precision 1e-6"
                                  if objective == NLO.DIAGQUAD else "chained Rosenbrock x0=1+0.1u, Precision 1e-10"),
                   "batch_per_gpu": B, "global_batch": Bglobal, "n": n, "memory": m, "solver": "LBFGS",
                   "line_search": "StrongWolfe",
                   "exchange": "gather x*,f*,iters,status to rank 0" if multi else "none"},
        "params_per_sec": total_iters_per_step * args.steps / dt * n,
        "iterations_per_step": total_iters_per_step,
        "ranks": {"backend": args.backend if multi else "none", "world_size": world,
                  "iterations_per_rank": per_rank_iters, "kernel_ms_per_rank": per_rank_kernel_ms,
                  "gather_ms": gather_ms, "exchange_overlaps_next_solve": bool(multi)},
        "converged_fraction": float((status == 0).mean()),
        "roofline": {"bound": "hbm", "kernel": "fl_solve_kernel<NW,EPT,OBJ,LBFGS> (fused solver)",
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
                             "served on chip, so it may exceed the peak and is not a utilisation"},
    }

    # ---- stand-alone two-loop recursion on the solver's own final history
    if not args.no_two_loop:
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
        for _ in range(2):
            NLO.two_loop(hist, rho_all, g, p, m, m - 1)
        torch.cuda.synchronize()
        reps = 10
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            NLO.two_loop(hist, rho_all, g, p, m, m - 1)
        e1.record()
        torch.cuda.synchronize()
        tl_ms = e0.elapsed_time(e1) / reps
        tl_bytes = float(B) * (4 * m + 2) * 8 * n      # SURVEY 8d accounting
        tl_moved = float(B) * (4 * m - 2 + 2) * 8 * n  # what the kernel moves: the oldest pair is fetched once
        res["two_loop"] = {"kernel": "two_loop_kernel<NW,EPT>", "achieved": tl_moved / (tl_ms * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": tl_moved / (tl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": tl_ms,
                           "moved_bytes_per_launch": tl_moved, "algorithmic_bytes_per_launch": tl_bytes,
                           "algorithmic_bw": tl_bytes / (tl_ms * 1e-3) / 1e9,
                           "note": "stand-alone micro-kernel (the product runs the recursion inside the fused kernel); "
                                   "22.5 GB per launch re-read ten times in a row: partly Infinity-Cache resident"}

    # ---- CPU baseline: the oracle on a bounded sample of the same workload, all host cores
    if args.cpu_sample != 0 and world == 1:  # reported at N = 1 only (the other ranks would wait at the barrier)
        import oracle_lib as O
        cores = host_cores()
        oo = O.defaults(precision=opts.precision, maxit=opts.max_iteration, memory=m)
        kind = O.DIAGQUAD if objective == NLO.DIAGQUAD else O.ROSENBROCK
        if args.cpu_sample > 0:
            S = min(B, args.cpu_sample)
        else:  # pilot on 4 problems per core, then size the sample for ~15 s of wall time
            S0 = min(B, 4 * cores)
            t1 = time.perf_counter()
            O.solve_batch(O.LBFGS, kind, x0[:S0].cpu().numpy(), d=d[:S0].cpu().numpy() if d is not None else None,
                          b=b[:S0].cpu().numpy() if b is not None else None, opts=oo, sum_mode=O.SEQ, nthreads=cores)
            pilot = max(time.perf_counter() - t1, 1e-3)
            S = int(min(B, max(S0, S0 * 15.0 / pilot)))
        xs = x0[:S].cpu().numpy()
        ds = d[:S].cpu().numpy() if d is not None else None
        bs = b[:S].cpu().numpy() if b is not None else None
        t1 = time.perf_counter()
        ref = O.solve_batch(O.LBFGS, kind, xs, d=ds, b=bs, opts=oo, sum_mode=O.SEQ, nthreads=cores)
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": float(ref["iters"].sum()) / cdt, "unit": "iterations/s",
                               "cores": int(ref["threads"]), "kind": "port",
                               "sample": f"first {S} problems of rank 0's batch, oracle in reference summation "
                                         f"order, OpenMP one problem per thread, {cdt:.1f} s wall; host: {cpu_model()}"}
        gx = x[:S].cpu().numpy()
        gf = out["f"][:S].cpu().numpy()
        den = np.maximum(np.abs(ref["f"]), 1e-10)
        ferr = np.abs(gf - ref["f"]) / den
        xerr = np.linalg.norm(gx - ref["x"], axis=1) / np.maximum(1.0, np.linalg.norm(ref["x"], axis=1))
        res["parity"] = {"final_f_rel_err_max": float(ferr.max()), "minimiser_err_max": float(xerr.max()),
                         "sample": S, "tolerance": {"f_rel": 1e-10, "x": 1e-8},
                         "f_within_tolerance_fraction": float((ferr <= 1e-10).mean()),
                         "x_within_tolerance_fraction": float((xerr <= 1e-8).mean()),
                         "minimiser_note": "at the benched Precision = 1e-6 the reference-order CPU run and the GPU run stop "
                                           "at different points of the set ||g|| < 1e-6, which with kappa up to 1e3 is ~1e-6 "
                                           "wide in x: the 1e-8 bar is therefore evaluated in `tight` below, at the tightest "
                                           "gradient tolerance the family attains"}
        # the same problems against the oracle in the kernels' summation order: every bit must agree
        SB = min(S, 256)
        tre = O.solve_batch(O.LBFGS, kind, xs[:SB], d=ds[:SB] if ds is not None else None,
                            b=bs[:SB] if bs is not None else None, opts=oo, sum_mode=O.TREE, threads=T_, ept=E_,
                            nthreads=cores)
        bit = {"problems": SB,
               "x": bool(np.array_equal(gx[:SB].view(np.uint64), tre["x"].view(np.uint64))),
               "f": bool(np.array_equal(gf[:SB].view(np.uint64), tre["f"].view(np.uint64))),
               "iterations": bool(np.array_equal(out["iters"][:SB].cpu().numpy(), tre["iters"]))}
        res["parity"]["bit_exact_vs_oracle_kernel_order"] = bit
        # the north-star bars (f: 1e-10 relative, minimiser: 1e-8) where a minimiser is defined that sharply: the same
        # first problems solved again by both sides at Precision = 1e-9 (unreachable for part of the family: those stop
        # on MinStepLength or MaxIteration, on either side, and are counted separately)
        tp = 1e-9
        xt = x0[:SB].clone()
        ot = NLO.LBFGS(objective, xt, d[:SB].contiguous() if d is not None else None,
                       b[:SB].contiguous() if b is not None else None,
                       options=NLO.default_options(NLO.LBFGS_, **dict(opt_kw, Precision=tp)))
        torch.cuda.synchronize()
        rt = O.solve_batch(O.LBFGS, kind, xs[:SB], d=ds[:SB] if ds is not None else None,
                           b=bs[:SB] if bs is not None else None,
                           opts=O.defaults(precision=tp, maxit=opts.max_iteration, memory=m), sum_mode=O.SEQ, nthreads=cores)
        gxt, gft, gst = xt.cpu().numpy(), ot["f"].cpu().numpy(), ot["status"].cpu().numpy()
        ferr_t = np.abs(gft - rt["f"]) / np.maximum(np.abs(rt["f"]), 1e-10)
        xerr_t = np.linalg.norm(gxt - rt["x"], axis=1) / np.maximum(1.0, np.linalg.norm(rt["x"], axis=1))
        both = (gst == 0) & (rt["status"] == 0)
        res["parity"]["tight"] = {
            "precision": tp, "problems": SB, "both_sides_met_the_gradient_test": int(both.sum()),
            "f_within_1e-10_fraction": float((ferr_t <= 1e-10).mean()),
            "x_within_1e-8_fraction": float((xerr_t <= 1e-8).mean()),
            "x_within_1e-8_fraction_where_both_met_the_gradient_test": float((xerr_t[both] <= 1e-8).mean()) if both.any() else None,
            "f_rel_err_max": float(ferr_t.max()), "x_err_max": float(xerr_t.max())}
        if objective == NLO.DIAGQUAD:  # the exact minimiser is known here (x* = b / d): how far is either side from it?
            xstar = bs[:SB] / ds[:SB]
            nrm = np.maximum(1.0, np.linalg.norm(xstar, axis=1))
            res["parity"]["tight"]["gpu_to_exact_minimiser_err_max"] = float((np.linalg.norm(gxt - xstar, axis=1) / nrm).max())
            res["parity"]["tight"]["cpu_reference_order_to_exact_minimiser_err_max"] = float(
                (np.linalg.norm(rt["x"] - xstar, axis=1) / nrm).max())
            res["parity"]["tight"]["note"] = (
                "no problem of this family (kappa up to 1e3, |f*| ~ 10..100) reaches ||g|| < 1e-9 on either side: an "
                "objective-value line search in fp64 stalls near ||g|| ~ 1e-7 and both stop on MinStepLength / MaxIteration "
                "(SURVEY.md section 6).  The reference's own answer is therefore defined to ~1e-7 in x -- see the two "
                "distances to the exact minimiser b/d -- and GPU and CPU agree to that level; where the gradient test IS "
                "attainable (kappa <= 100) the 1e-8 bar holds: tests/test_gpu_parity.py::"
                "test_north_star_tolerance_vs_reference_summation")
        res["parity"]["ok"] = bool(ferr.max() <= 1e-10 and bit["x"] and bit["f"] and bit["iterations"])
        res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]

    print(json.dumps(res))
    sys.stdout.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if "parity" in res and not res["parity"]["ok"]:
        sys.stderr.write("bench.py: PARITY VIOLATION (objective tolerance or bit-exactness against the oracle)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
