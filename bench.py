#!/usr/bin/env python3
"""bench.py -- batched L-BFGS on MI355X: iterations/sec, HBM roofline, CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
by torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

A "step" is one complete pass of the hot path over one batch: every problem of the batch is
solved from its initial guess to convergence by ONE launch of the fused solver kernel
(fl_lbfgs_batched), then -- when N > 1 -- the converged results (x*, f*, iterations, status)
are gathered to rank 0 with one RCCL gather per array.  The workload is the north-star
configuration: L-BFGS, Memory = 10, batch 65536 per GPU (weak scaling), n = 1024 convex
diagonal quadratics of BASELINE.json's config-3 family (kappa log-uniform in [10, 1000],
b ~ U(-1,1), x0 = 0, Precision = 1e-6: the tightest
gradient tolerance every problem of this family can meet with an fp64 objective-value line search), inputs generated on the device (Philox) and resident
in HBM before the timed region.  value = L-BFGS iterations (line searches) of all ranks / sec.

roofline: the fused solver kernel is the only kernel in the timed region.  Its ALGORITHMIC
bytes per launch are the two-loop recursion's streaming bytes, (4*cnt + 2) * 8n per
iteration and problem with cnt = min(history, Memory) (line-search trials and
direction updates run from registers and are credited with nothing), divided by the kernel's
launch duration measured with HIP events on the launch stream.  "two_loop" reports the
stand-alone two-loop kernel on the same history for the north-star "fraction of HBM roofline
on the two-loop recursion".  cpu_baseline: the CPU oracle (reference summation order),
OpenMP over problems on all host cores, on a bounded sample of rank 0's batch.
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
    ap.add_argument("--batch", type=int, default=65536, help="problems per GPU")
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
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("FL_BENCH_ONE_DEVICE"):  # rehearsal: every rank on GPU 0 (needs --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
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
    B, n, m = args.batch, args.n, args.memory

    # ---- synthetic inputs, generated on the device, resident in HBM
    x0 = torch.zeros(B, n, dtype=torch.float64, device=dev)
    d = b = None
    if objective == NLO.DIAGQUAD:
        d = torch.empty(B, n, dtype=torch.float64, device=dev)
        b = torch.empty(B, n, dtype=torch.float64, device=dev)
        NLO.synth_diag_spectrum(SEED + rank, d, 10.0, 1000.0)
        NLO.synth_uniform(SEED + rank, b, -1.0, 1.0)
    else:
        NLO.synth_uniform(SEED + rank, x0, 0.9, 1.1)  # x0 = 1 + 0.1 u
    x = torch.empty_like(x0)
    ws = NLO.workspace(B, n, m, dev)
    opts = NLO.default_options(NLO.LBFGS_, **opt_kw)
    from FortranLibrary import distributed as D

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
        if world > 1:  # the single exchange of the path: converged results to rank 0 over xGMI
            D.gather_results({"x": x, "f": out["f"], "iters": out["iters"], "status": out["status"]}, dst=0)
        return out

    def sync():
        if world > 1:
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
    iters_step = out["iters"].to(torch.int64).sum().reshape(1).to(cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(iters_step, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_iters_per_step = int(iters_step.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- rank 0: roofline of the solver kernel (HIP events on the launch stream)
    it = out["iters"].to(torch.int64)
    k = torch.clamp(it - 1, min=0)  # two-loops performed per problem
    full = torch.clamp(k - m, min=0)
    part = torch.minimum(k, torch.tensor(m, device=dev))
    cnt_sum = part * (part + 1) // 2 + full * m  # sum of min(j, m), j = 1..k
    algo_bytes = float((8 * n * (4 * cnt_sum + 2 * k)).sum().item())
    kern_ms = sum(a.elapsed_time(bb) for a, bb in ev) / len(ev)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    status = out["status"].cpu().numpy()
    traffic = None  # PMC-measured HBM-side bytes per launch, recorded from a separate rocprofv3 pass
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(args.workload)
        if rec and (rec["batch_per_gpu"], rec["n"], rec["memory"]) == (B, n, m) and rec["precision"] == opts.precision:
            traffic = rec["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    res = {
        "metric": "lbfgs_iterations_per_sec",
        "value": total_iters_per_step * args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"L-BFGS m={m}, batch {B} per GPU, n={n}, "
                               + ("convex diagonal quadratics kappa in [10,1000], Precision 1e-6"
                                  if objective == NLO.DIAGQUAD else "chained Rosenbrock x0=1+0.1u, Precision 1e-10"),
                   "batch_per_gpu": B, "n": n, "memory": m, "solver": "LBFGS", "line_search": "StrongWolfe",
                   "exchange": "gather x*,f*,iters,status to rank 0" if world > 1 else "none"},
        "params_per_sec": total_iters_per_step * args.steps / dt * n,
        "iterations_per_step": total_iters_per_step,
        "converged_fraction": float((status == 0).mean()),
        "roofline": {"bound": "hbm", "kernel": "fl_solve_kernel<NW,EPT,OBJ,LBFGS> (fused solver)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "algorithmic bytes = two-loop recursion only, (4*cnt+2)*8n per iteration and problem"},
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
        tl_bytes = float(B) * (4 * m + 2) * 8 * n
        res["two_loop"] = {"kernel": "two_loop_kernel<NW,EPT>", "achieved": tl_bytes / (tl_ms * 1e-3) / 1e9,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": tl_bytes / (tl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": tl_ms,
                           "algorithmic_bytes_per_launch": tl_bytes}

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
        res["parity"] = {"final_f_rel_err_max": float(np.max(np.abs(gf - ref["f"]) / den)),
                         "minimiser_err_max": float(np.max(np.linalg.norm(gx - ref["x"], axis=1)
                                                           / np.maximum(1.0, np.linalg.norm(ref["x"], axis=1)))),
                         "sample": S, "tolerance": {"f_rel": 1e-10, "x": 1e-8},
                         "minimiser_note": "reference-order CPU run and GPU run both stop at ||g|| < Precision = 1e-6; with "
                                           "kappa up to 1e3 that defines the minimiser only to ~1e-6, the distance between "
                                           "two valid stopping points.  The 1e-8 tolerance is checked at Precision 1e-9 in "
                                           "tests/test_gpu_parity.py::test_north_star_tolerance_vs_reference_summation"}
        # the same problems against the oracle in the kernels' summation order: every bit must agree
        SB = min(S, 256)
        T_, E_ = NLO.reduction_geometry(n)
        tre = O.solve_batch(O.LBFGS, kind, xs[:SB], d=ds[:SB] if ds is not None else None,
                            b=bs[:SB] if bs is not None else None, opts=oo, sum_mode=O.TREE, threads=T_, ept=E_,
                            nthreads=cores)
        res["parity"]["bit_exact_vs_oracle_kernel_order"] = {
            "problems": SB,
            "x": bool(np.array_equal(gx[:SB].view(np.uint64), tre["x"].view(np.uint64))),
            "f": bool(np.array_equal(gf[:SB].view(np.uint64), tre["f"].view(np.uint64))),
            "iterations": bool(np.array_equal(out["iters"][:SB].cpu().numpy(), tre["iters"]))}
        res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]

    print(json.dumps(res))
    sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
