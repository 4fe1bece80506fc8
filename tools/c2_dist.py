import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fortran-library_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import FortranLibrary.NonlinearOptimization as NLO
from geometry_by_batch import workload
for B in (4096, 16384, 65536):
    run, _ = workload("c2", B)
    out = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = run(); e1.record(); torch.cuda.synchronize()
    it = out["iters"].double(); nf = out["nf"].double()
    print(B, "ms", round(e0.elapsed_time(e1), 3), "iters min/mean/max", int(it.min()), round(float(it.mean()), 1), int(it.max()), "nf min/mean/max", int(nf.min()), round(float(nf.mean()), 1), int(nf.max()),
          "M it/s", round(float(it.sum()) / e0.elapsed_time(e1) / 1e3, 1), flush=True)
