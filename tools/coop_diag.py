"""diagnosis / measurement: one problem of n unknowns through the fused L-BFGS with 1 ... 256 workgroups per problem"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fortran-library_amd"))
import FortranLibrary.NonlinearOptimization as NLO

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
kappa = np.exp(rng.uniform(np.log(10), np.log(200), 1))  # (the family of tests/test_gpu_cooperative.py)
d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / max(n - 1, 1))[None, :]
b = rng.uniform(-1, 1, (1, n))
dd, bb = torch.tensor(d, device=dev), torch.tensor(b, device=dev)
for G in (1, 4, 16, 32, 64, 128, 256):
    os.environ["FL_COOP_GROUPS"] = str(G)
    ws = None
    for rep in range(2):
        x = torch.zeros(1, n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = NLO.LBFGS(NLO.DIAGQUAD, x, dd, bb, Precision=1e-6, MaxIteration=60, workspace_=ws)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) * 1e3
        ws = out["workspace"]
    print(f"G asked {G:3d} used {NLO.cooperative_groups(NLO.LBFGS_, NLO.DIAGQUAD, 1, n):3d}: {ms:8.2f} ms  iters {int(out['iters'][0])} nf {int(out['nf'][0])} "
          f"status {int(out['status'][0])} f {float(out['f'][0]):.15e} gg {float(out['gg'][0]):.3e}", flush=True)
