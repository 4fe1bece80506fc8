"""Scratch exploration on the GPU box (not part of the product or the tests)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import FortranLibrary.NonlinearOptimization as NLO

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup", e)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; nproc")
dev = torch.device("cuda:0")
B, n, m = 16384, 1024, 10
d = torch.empty(B, n, dtype=torch.float64, device=dev); b = torch.empty_like(d)
NLO.synth_diag_spectrum(20261003, d, 10.0, 1000.0); NLO.synth_uniform(20261003, b, -1.0, 1.0)
ws = NLO.workspace(B, n, m, dev)
for prec in (1e-9, 1e-8, 1e-7, 1e-6):
    for f_fd in (False, True):
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=prec, MaxIteration=3000, f_fd=f_fd)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy(); nf = out["nf"].cpu().numpy(); ng = out["ng"].cpu().numpy()
        xs = (b / d); err = ((x - xs).norm(dim=1) / xs.norm(dim=1)).cpu().numpy()
        print(f"prec {prec:g} f_fd {f_fd}: {dt*1e3:.1f} ms conv {np.mean(st==0):.3f} step {np.mean(st==1):.3f} maxit {np.mean(st==2):.3f} "
              f"iters mean {it.mean():.1f} max {it.max()} nf/iter {nf.sum()/it.sum():.2f} ng/iter {ng.sum()/it.sum():.2f} "
              f"relerr_x med {np.median(err):.2e} max {err.max():.2e} Mit/s {it.sum()/dt/1e6:.2f}")
