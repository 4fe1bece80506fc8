#!/usr/bin/env python3
"""step-by-step trace of the run-time compiled objective path (each step appended to gpurun_out/rtc_steps.log before it runs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
LOG = os.path.join(ROOT, "gpurun_out", "rtc_steps.log")
def step(msg):
    with open(LOG, "a") as fh:
        fh.write(f"{time.time():.3f} {msg}\n"); fh.flush(); os.fsync(fh.fileno())
step("start")
import torch
step("torch imported")
import FortranLibrary.NonlinearOptimization as NLO
import user_sources as US
step("library imported")
rc, log = NLO.compile_check(US.DIAGQUAD, "MyQuadratic", 256, 2, 2)
step(f"compile_check rc={rc}")
dev = torch.device("cuda:0")
x = torch.zeros(4, 256, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
step("device initialised")
obj = NLO.compile_objective(US.DIAGQUAD, "MyQuadratic", 256, solver=NLO.LBFGS_, tune_like=NLO.DIAGQUAD)
step(f"compile_objective done geometry={obj.geometry}")
d = torch.full((4, 256), 2.0, dtype=torch.float64, device=dev); b = torch.ones(4, 256, dtype=torch.float64, device=dev)
out = obj.solve(x, d, b, None, Precision=1e-6, MaxIteration=5)
step("solve launched")
torch.cuda.synchronize()
step(f"solve finished iters={out['iters'].tolist()} f={out['f'].tolist()}")
