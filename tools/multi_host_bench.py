"""The host-array form of the boundary (fl_multi_solve: numpy arrays in and out, upload / solve / download inside) on the headline
workload: the PCIe-inclusive rate, for 1, 2 and 4 shards on however many GPUs the node has (shards on one GPU overlap one
shard's transfers with another's solve).  usage: python3 tools/multi_host_bench.py [batch]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import torch  # noqa: E402
import FortranLibrary.NonlinearOptimization as NLO  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n, m = 1024, 10
dev = torch.device("cuda:0")
d = torch.empty(B, n, dtype=torch.float64, device=dev)
b = torch.empty(B, n, dtype=torch.float64, device=dev)
NLO.synth_diag_spectrum(20240607, d, 10.0, 1000.0)
NLO.synth_uniform(20240607, b, -1.0, 1.0)
dh, bh = d.cpu().numpy(), b.cpu().numpy()
ws = NLO.workspace(B, n, m, dev)
x = torch.zeros(B, n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
out = NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)
torch.cuda.synchronize()
t = time.perf_counter()
x.zero_()
out = NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t) * 1e3
it = int(out["iters"].to(torch.int64).sum())
xd = x.cpu().numpy()
del d, b, x, ws
torch.cuda.empty_cache()
print(json.dumps({"form": "device pointers (what bench.py times)", "ms": dev_ms, "iterations_per_s": it / dev_ms * 1e3}))
for shards in (0, 1, 2, 4, 8):
    best = 1e9
    for rep in range(2):
        xh = np.zeros((B, n))
        t = time.perf_counter()
        o = NLO.multi_solve(NLO.LBFGS_, NLO.DIAGQUAD, xh, dh, bh, nshards=shards, Precision=1e-6, MaxIteration=3000, Memory=m)
        best = min(best, (time.perf_counter() - t) * 1e3)
    same = bool(np.array_equal(xh, xd) and int(o["iters"].astype(np.int64).sum()) == it)
    print(json.dumps({"form": f"host arrays, fl_multi_solve, {shards or 'default'} shard(s) on {NLO.FL.fl_multi_device_count()} GPU(s)", "ms": best,
                      "iterations_per_s": it / best * 1e3, "host_bytes_moved": int(4 * B * n * 8), "same_bits_as_device_form": same}))
