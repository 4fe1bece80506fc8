"""C5 (aug-Lagrangian + L-BFGS, 8192 x n=512): how much of the kernel's time is its tail?  The problems are independent
and a workgroup owns one from start to end, dispatched in batch order; the evaluation counts differ 16-fold.  Upper
bound for any scheduling: the same batch with the problems sorted by decreasing cost (longest first).
usage: python3 tools/c5_order_experiment.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import torch  # noqa: E402
import FortranLibrary.NonlinearOptimization as NLO  # noqa: E402

dev = torch.device("cuda:0")
B, n, M, m = 8192, 512, 8, 10
SEED = 20240607
d = torch.empty(B, n, dtype=torch.float64, device=dev)
b = torch.empty(B, n, dtype=torch.float64, device=dev)
NLO.synth_diag_spectrum(SEED, d, 2.0, 10.0)
NLO.synth_uniform(SEED, b, -1.0, 1.0)
x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
NLO.synth_uniform(SEED + 7, x0, 0.05, 0.15)
ws = NLO.workspace(B, n, m, dev)


def run(dd, bb, xx0, reps=3):
    x = torch.empty_like(xx0)
    best = 1e9
    out = None
    for _ in range(reps):
        x.copy_(xx0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, dd, bb, UnconstrainedSolver="LBFGS", workspace_=ws, Precision=1e-10, Memory=m)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, out


ms, out = run(d, b, x0)
nf = out["nf"].to(torch.int64)
print(json.dumps({"order": "as generated", "ms": ms, "nf_min_mean_max": [int(nf.min()), float(nf.double().mean()), int(nf.max())]}))
np.save(os.path.join(ROOT, "gpurun_out", "c5_nf.npy"), nf.cpu().numpy())
for name, perm in (("longest first", torch.argsort(nf, descending=True)), ("shortest first", torch.argsort(nf)),
                   ("random permutation", torch.randperm(B, device=dev))):
    ms2, out2 = run(d[perm].contiguous(), b[perm].contiguous(), x0[perm].contiguous())
    same = bool(torch.equal(out2["nf"].to(torch.int64), nf[perm]))
    print(json.dumps({"order": name, "ms": ms2, "same_counts": same}))

# How early is a problem's cost known?  Pilot runs stopped after k outer iterations (MaxIteration = k), their
# evaluation counts as the predictor of the order.
for k in (2, 4, 8, 12, 16, 24):
    xk = x0.clone()
    outk = NLO.AugmentedLagrangian(NLO.DIAGQUAD, xk, M, d, b, UnconstrainedSolver="LBFGS", workspace_=ws, Precision=1e-10, Memory=m,
                                   MaxIteration=k)
    torch.cuda.synchronize()
    nfk = outk["nf"].to(torch.int64)
    corr = float(torch.corrcoef(torch.stack([nfk.double(), nf.double()]))[0, 1])
    cn = outk["cnorm2"].double()
    corr_c = float(torch.corrcoef(torch.stack([cn.clamp_min(1e-300).log(), nf.double()]))[0, 1])
    permc = torch.argsort(cn, descending=True)
    msc, _ = run(d[permc].contiguous(), b[permc].contiguous(), x0[permc].contiguous())
    print(json.dumps({"pilot_outer_iterations": k, "predictor": "constraint norm after the pilot", "correlation_of_log_cnorm2_with_total": corr_c,
                      "ms_in_that_order": msc, "done_in_pilot": float((outk["status"] == 0).double().mean())}))
    perm = torch.argsort(nfk, descending=True)
    ms2, out2 = run(d[perm].contiguous(), b[perm].contiguous(), x0[perm].contiguous())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xk.copy_(x0)
    torch.cuda.synchronize()
    e0.record()
    NLO.AugmentedLagrangian(NLO.DIAGQUAD, xk, M, d, b, UnconstrainedSolver="LBFGS", workspace_=ws, Precision=1e-10, Memory=m, MaxIteration=k)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"pilot_outer_iterations": k, "pilot_ms": e0.elapsed_time(e1), "pilot_share_of_evaluations": float(nfk.sum()) / float(nf.sum()),
                      "correlation_with_total": corr, "ms_in_pilot_order": ms2}))
