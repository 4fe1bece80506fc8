import sys, os, time
sys.path.insert(0, "fortran-library_amd"); sys.path.insert(0, "tests")
import numpy as np, torch
import FortranLibrary.NonlinearOptimization as NLO
from FortranLibrary.basic import FL
dev = torch.device("cuda:0")
def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); t.append(e0.elapsed_time(e1))
    return min(t)
for n, B in ((256, 4096), (512, 1024), (1024, 256), (1024, 16), (2048, 64), (4096, 16), (4096, 1), (8192, 4)):
    T, E = NLO.reduction_geometry(n); ld = T * E
    G = torch.randn(B, n, n, dtype=torch.float64, device=dev)
    A0 = torch.zeros(B, n, ld, dtype=torch.float64, device=dev)
    A0[:, :, :n] = G @ G.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64, device=dev)
    del G
    b0 = torch.randn(B, n, dtype=torch.float64, device=dev)
    A, b = A0.clone(), b0.clone()
    res = {}
    for mode, thr in (("seq", 1 << 30), ("blocked", 32)):
        if mode == "seq" and n > 4096: continue
        FL.fl_set_chol_blocked_min_n(thr)
        def sv():
            A.copy_(A0); b.copy_(b0); NLO.dposv(A, b)
        def tri():
            A.copy_(A0); NLO.dpotri(A)
        def cp():
            A.copy_(A0); b.copy_(b0)
        c = timed(cp)
        res[mode] = (timed(sv) - c, timed(tri) - c)
    print(n, B, {k: (round(v[0], 2), round(v[1], 2)) for k, v in res.items()}, "flops posv %.2e" % (B * n**3 / 3), flush=True)
