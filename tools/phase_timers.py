#!/usr/bin/env python3
"""Where a problem's wall time goes inside the fused kernel, by phase (tuning build libFL_timers.so, tools/build_timers_variant.sh;
run with FL_LIBRARY=<that file>): BASELINE config 5's family at --batch problems under each geometry; prints, for the slowest
problem and the mean, time (ms) and calls in: fast_forward (tight objective-only loop), fast_forward_grow, objective-only
evaluation, full evaluation, advance() inside a search, advance() ending a search (convergence tests + new direction = the
two-loop recursion).  usage: FL_LIBRARY=.../libFL_timers.so python tools/phase_timers.py [--batch 1024] """
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd")); sys.path.insert(0, ROOT)
import torch
import FortranLibrary.NonlinearOptimization as NLO
from bench import SEED
sys.path.insert(0, os.path.join(ROOT, "tools"))
from geometry_by_batch import workload

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--workload", default="c5")
ap.add_argument("--geos", default="1x8")
a = ap.parse_args()
names = ["fast_forward", "fast_forward_grow", "eval_f", "eval_fg", "advance_in_search", "advance_direction"]
run, _ = workload(a.workload, a.batch)
buf = torch.zeros(a.batch, 12, dtype=torch.int64, device="cuda:0")
os.environ["FL_PHASE_BUFFER"] = hex(buf.data_ptr())
for geo in a.geos.split(","):
    os.environ["FL_FORCE_GEOMETRY"] = geo
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf.zero_(); e0.record(); out = run(); e1.record(); torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(float)
    tot = t[:, :6].sum(1)
    k = int(tot.argmax())
    row = {"geometry": geo, "kernel_ms": round(e0.elapsed_time(e1), 2), "slowest_problem": k,
           "slowest": {n: [round(t[k, i] * 1e-5, 2), int(t[k, 6 + i])] for i, n in enumerate(names)},
           "slowest_total_ms": round(tot[k] * 1e-5, 2), "slowest_nf_ng_iters_outer": [int(out[q][k]) for q in ("nf", "ng", "iters", "outer") if q in out],
           "mean": {n: [round(t[:, i].mean() * 1e-5, 2), int(t[:, 6 + i].mean())] for i, n in enumerate(names)}}
    print(json.dumps(row), flush=True)
