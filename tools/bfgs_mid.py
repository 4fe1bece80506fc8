"""side measurement: quasi-Newton BFGS in the immediate rank-2 form (n <= 1024), 20 iterations of a batch whose matrices do not fit any cache"""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd")); sys.path.insert(0, ROOT)
import ctypes as C
import FortranLibrary.NonlinearOptimization as NLO
dev = torch.device("cuda:0")
SIZES = [tuple(int(v) for v in s.split("x")) for s in os.environ["BFGS_MID_SIZES"].split(",")] if "BFGS_MID_SIZES" in os.environ else ((1024, 4096), (512, 8192), (256, 16384), (128, 32768), (64, 65536), (10, 65536))
for n, B in SIZES[:int(os.environ.get("BFGS_MID_CASES", "6"))]:
    d = torch.empty(B, n, dtype=torch.float64, device=dev); b = torch.empty_like(d)
    NLO.synth_diag_spectrum(7, d, 10.0, 100.0); NLO.synth_uniform(7, b, -1.0, 1.0)
    x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    opt = NLO.default_options(NLO.BFGS_, Precision=1e-12, MaxIteration=19, ExactStep=0)
    nbytes = NLO.FL.fl_workspace_bytes_for(NLO.BFGS_, B, n, C.byref(opt))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
    ms = []
    for rep in range(4):
        x.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-12, MaxIteration=19, ExactStep=0); e1.record()
        torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    it = int(out["iters"].to(torch.int64).sum())
    print(os.path.basename(os.environ.get("FL_LIBRARY", "libFL.so")), json.dumps({"n": n, "batch": B, "ms": round(min(ms[1:]), 2), "iterations": it,
          "H_GB_per_s": round(it * 2 * n * n * 8 / min(ms[1:]) / 1e6, 1)}), flush=True)
    del ws, d, b, x
    torch.cuda.empty_cache()
