"""side measurement: quasi-Newton BFGS by reverse communication (torch objective), n = 1024, 512 problems, 30 iterations"""
import os, sys, time, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import FortranLibrary.NonlinearOptimization as NLO
dev = torch.device("cuda:0")
for n, B in ((1024, 512), (512, 1024), (2048, 128)):
    d = torch.empty(B, n, dtype=torch.float64, device=dev); b = torch.empty_like(d)
    NLO.synth_diag_spectrum(7, d, 10.0, 100.0); NLO.synth_uniform(7, b, -1.0, 1.0)
    def fun(x, req=None):
        dx = d * x
        return 0.5 * (dx * x).sum(1) - (b * x).sum(1), dx - b
    best = 1e9
    for rep in range(3):
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = NLO.minimize_rci(NLO.BFGS_, x, fun, Precision=1e-12, MaxIteration=29, ExactStep=0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(os.path.basename(os.environ.get("FL_LIBRARY", "libFL.so")), json.dumps({"n": n, "batch": B, "ms": round(best * 1e3, 1), "iterations": int(out["iters"].sum()), "steps": out.get("steps")}), flush=True)
