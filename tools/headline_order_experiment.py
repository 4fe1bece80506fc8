"""Headline workload at one GPU's share of a strong-scaling run (8192 of the 65 536 problems): how much of the kernel's time is
its tail, and does an order known from the problem data (condition number of the diagonal quadratic) remove it?
usage: python3 tools/headline_order_experiment.py [batch ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import torch  # noqa: E402
import FortranLibrary.NonlinearOptimization as NLO  # noqa: E402

dev = torch.device("cuda:0")
n, m, SEED = 1024, 10, 20240607
for B in [int(a) for a in sys.argv[1:]] or [8192, 16384, 65536]:
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty(B, n, dtype=torch.float64, device=dev)
    NLO.synth_diag_spectrum(SEED, d, 10.0, 1000.0)
    NLO.synth_uniform(SEED, b, -1.0, 1.0)
    ws = NLO.workspace(B, n, m, dev)

    def run(dd, bb, reps=3):
        best, out = 1e9, None
        x = torch.empty(B, n, dtype=torch.float64, device=dev)
        for _ in range(reps):
            x.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = NLO.LBFGS(NLO.DIAGQUAD, x, dd, bb, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best, out

    ms, out = run(d, b)
    it = out["iters"].to(torch.int64)
    kappa = d.max(1).values / d.min(1).values
    corr = float(torch.corrcoef(torch.stack([kappa.log(), it.double()]))[0, 1])
    print(json.dumps({"batch": B, "order": "as generated", "ms": ms, "iterations_min_mean_max": [int(it.min()), float(it.double().mean()), int(it.max())],
                      "corr_log_kappa_iterations": corr}))
    for name, perm in (("most iterations first (oracle)", torch.argsort(it, descending=True)), ("largest kappa first", torch.argsort(kappa, descending=True)),
                       ("fewest iterations first", torch.argsort(it))):
        ms2, _ = run(d[perm].contiguous(), b[perm].contiguous())
        print(json.dumps({"batch": B, "order": name, "ms": ms2}))
