#!/usr/bin/env python3
"""Wall time of the fused kernels by batch size and waves per problem (MI355X, one GPU): the table behind the break-even
batches of the helper-wave kernels (csrc/fl_solver_kernels.hip: select_replicas) -> profiles/r04/geometry_by_batch.txt.

For every workload and batch the same problems (the first `batch` of the benched family) are solved by the plain kernel and
with 1 / 3 helper waves ("r2" / "r4": FL_FORCE_REPLICAS, read by the library per call), three launches each, the minimum kept
(HIP events on the launch stream).  (At commit ca4f3a1 this tool also forced the latency geometries 2x2 .. 8x4 that round 4
measured and dropped: FL_FORCE_GEOMETRY; those columns of the committed table come from there.)
usage: python tools/geometry_by_batch.py [c5 headline c3 c2 lbfgs512 cg512 lbfgs2048] [--batches 256,1024,2048,4096,8192]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
sys.path.insert(0, ROOT)
import torch
import FortranLibrary.NonlinearOptimization as NLO
from bench import SEED

dev = torch.device("cuda:0")


def quad(B, n, klo, khi):
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty(B, n, dtype=torch.float64, device=dev)
    NLO.synth_diag_spectrum(SEED, d, klo, khi)
    NLO.synth_uniform(SEED, b, -1.0, 1.0)
    return d, b


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best, out = None, None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return out, best


def workload(name, B):
    """-> (run(), candidates) for the first B problems of the family"""
    if name == "c5":
        n, M, m = 512, 8, 10
        d, b = quad(B, n, 2.0, 10.0)
        x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
        NLO.synth_uniform(SEED + 7, x0, 0.05, 0.15)
        x = torch.empty_like(x0)
        ws = NLO.workspace(B, n, m, dev)

        def run():
            x.copy_(x0)
            return NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, d, b, UnconstrainedSolver="LBFGS", workspace_=ws, Precision=1e-10, Memory=m)
        return run, ["1x8", "1x8r2", "1x8r4", "auto"]
    if name in ("headline", "lbfgs512", "lbfgs256", "lbfgs2048"):
        n = {"headline": 1024, "lbfgs512": 512, "lbfgs256": 256, "lbfgs2048": 2048}[name]
        d, b = quad(B, n, 10.0, 1000.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        ws = NLO.workspace(B, n, 10, dev)

        def run():
            x.zero_()
            return NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000)
        return run, {1024: ["2x8"], 512: ["1x8"], 256: ["1x4"], 2048: ["4x8"]}[n]
    if name in ("c3", "cg512"):
        n = 1024 if name == "c3" else 512
        d, b = quad(B, n, 10.0, 1000.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)

        def run():
            x.zero_()
            return NLO.ConjugateGradient(NLO.DIAGQUAD, x, d, b, Precision=1e-6, MaxIteration=3000)
        return run, (["1x16"] if n == 1024 else ["1x8"])
    if name == "c2":
        n, m = 256, 10
        x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
        NLO.synth_uniform(SEED, x0, 0.9, 1.1)
        ws = NLO.workspace(B, n, m, dev)
        x = torch.empty_like(x0)

        def run():
            x.copy_(x0)
            return NLO.LBFGS(NLO.ROSENBROCK, x, workspace_=ws, Precision=1e-10, MaxIteration=3000, Memory=m)
        return run, ["1x4"]
    if name == "c4":  # dense BFGS n = 4096, a fixed 20 iterations (bench.py config_c4)
        n = 4096
        d, b = quad(B, n, 10.0, 100.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        opt = NLO.default_options(NLO.BFGS_, Precision=1e-12, MaxIteration=19, ExactStep=0)
        import ctypes as C
        nbytes = NLO.FL.fl_workspace_bytes_for(NLO.BFGS_, B, n, C.byref(opt))
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)

        def run():
            x.zero_()
            return NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-12, MaxIteration=19, ExactStep=0)
        return run, ["8x8"]
    raise SystemExit("unknown workload " + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["c5", "headline"])
    ap.add_argument("--batches", default="256,1024,2048,4096,8192")
    args = ap.parse_args()
    for name in args.workloads:
        for B in [int(v) for v in args.batches.split(",")]:
            run, cands = workload(name, B)
            row = {"workload": name, "batch": B}
            for i, c in enumerate(cands):
                # (the first one is the throughput geometry: not a latency candidate -> ignored; "r2" / "r4": replicated groups)
                if c == "auto":  # what the library does by itself (helper waves by batch, staged launches)
                    os.environ.pop("FL_FORCE_REPLICAS", None)
                else:
                    os.environ["FL_FORCE_REPLICAS"] = c.split("r")[1] if "r" in c else "1"
                out, ms = timed(run)
                row[c] = round(ms, 3)
                row[c + "_iters"] = int(out["iters"].to(torch.int64).sum())
            os.environ.pop("FL_FORCE_GEOMETRY", None)
            os.environ.pop("FL_FORCE_REPLICAS", None)
            print(json.dumps(row), flush=True)
            del run
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
