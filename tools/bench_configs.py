#!/usr/bin/env python3
"""Measure the other BASELINE.json configurations (C2-C5) on one MI355X: one JSON line each.

Not the driver's bench (that is bench.py = the north-star headline); this records the numbers DESIGN.md quotes
for the remaining rows of SURVEY.md section 8.  Same method: inputs generated on the device, one launch of the
fused kernel per solve, HIP events on the launch stream, algorithmic bytes per SURVEY.md 8(d), CPU oracle
(reference summation order, OpenMP one problem per thread on the cgroup's cores) on a bounded sample.
usage: python tools/bench_configs.py [c1 c2 c3 c4 c4gemm c5 newton dense big] [--cpu-seconds 10]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import FortranLibrary.NonlinearOptimization as NLO
import oracle_lib as O
from bench import host_cores, cpu_model, SEED, HBM_PEAK_GBS

dev = torch.device("cuda:0")


REPS = None  # --reps: fixed number of timed launches (1 for the PMC passes of tools/profile.sh; no warm-up launch then)


def timed(fn, reps=2):
    if REPS is not None:
        reps = REPS
    if REPS is None:
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return out, sum(ms) / len(ms)


def cpu_rate(call, S0, seconds, cores):
    """iterations/s of the oracle: pilot on S0 problems, then a sample sized for `seconds` of wall time"""
    if seconds <= 0:  # profiling runs: no CPU leg
        class _Nothing(dict):
            def __getitem__(self, k):
                return np.full(1 << 20, np.nan)
        return float("nan"), 1, 0.0, _Nothing()
    t = time.perf_counter()
    call(S0)
    pilot = max(time.perf_counter() - t, 1e-3)
    S = int(max(S0, S0 * seconds / pilot))
    t = time.perf_counter()
    r = call(S)
    dt = time.perf_counter() - t
    return float(r["iters"].sum()) / dt, S, dt, r


def quad(B, n, klo, khi):
    d = torch.empty(B, n, dtype=torch.float64, device=dev)
    b = torch.empty(B, n, dtype=torch.float64, device=dev)
    NLO.synth_diag_spectrum(SEED, d, klo, khi)
    NLO.synth_uniform(SEED, b, -1.0, 1.0)
    return d, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c2", "c3", "c4", "c5"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--reps", type=int, default=None, help="timed launches per configuration (profiling: 1, no warm-up)")
    ap.add_argument("--c4-batch", type=int, default=1024)
    ap.add_argument("--c4-iterations", type=int, default=20)
    ap.add_argument("--c5-batch", type=int, default=8192)
    ap.add_argument("--rci-batch", type=int, default=16384)
    ap.add_argument("--coop-iterations", type=int, default=60)
    ap.add_argument("--rci-modes", default="legacy,full,compact")
    ap.add_argument("--rci-fixed-shape", action="store_true", help="rci: the torch objective always reduces over the full [batch, n] buffers")
    args = ap.parse_args()
    global REPS
    REPS = args.reps
    cores = host_cores()
    host = f"{cores} cores of {cpu_model()}"

    if "c2" in args.configs:  # Batch 4096 independent Rosenbrock n=256, L-BFGS m=10
        B, n, m = 4096, 256, 10
        x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
        NLO.synth_uniform(SEED, x0, 0.9, 1.1)
        ws = NLO.workspace(B, n, m, dev)
        x = torch.empty_like(x0)

        def run():
            x.copy_(x0)
            return NLO.LBFGS(NLO.ROSENBROCK, x, workspace_=ws, Precision=1e-10, MaxIteration=3000, Memory=m)
        out, ms = timed(run, 5)
        it = out["iters"].to(torch.int64)
        k = torch.clamp(it - 1, min=0)
        part = torch.clamp(k, max=m)
        cnt = part * (part + 1) // 2 + torch.clamp(k - m, min=0) * m
        algo = float((8 * n * (4 * cnt + 2 * k)).sum())
        oo = O.defaults(precision=1e-10, maxit=3000, memory=m)
        xh = x0.cpu().numpy()
        rate, S, dt, ref = cpu_rate(lambda s: O.solve_batch(O.LBFGS, O.ROSENBROCK, xh[:min(s, B)], opts=oo, nthreads=cores),
                                    4 * cores, args.cpu_seconds, cores)
        S = min(S, B)
        print(json.dumps({"config": "C2 L-BFGS m=10, Rosenbrock n=256, batch 4096, Precision 1e-10", "ms": ms,
                          "iterations_per_s": float(it.sum()) / ms * 1e3, "iterations": int(it.sum()),
                          "converged_fraction": float((out["status"] == 0).double().mean()),
                          "algorithmic_GBps_two_loop": algo / ms / 1e6, "frac_of_8TBps": algo / ms / 1e6 / HBM_PEAK_GBS,
                          "cpu_iterations_per_s": rate, "cpu": host, "cpu_sample": S,
                          "max_abs_x_minus_1": float((x - 1).abs().max()),
                          "final_f_abs_err_max_vs_cpu": float(np.max(np.abs(out["f"][:S].cpu().numpy() - ref["f"][:S])))}))

    if "c3" in args.configs:  # Batch 65536 convex quadratics n=1024, Dai-Yuan CG
        B, n = 65536, 1024
        d, b = quad(B, n, 10.0, 1000.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)

        def run():
            x.zero_()
            return NLO.ConjugateGradient(NLO.DIAGQUAD, x, d, b, Precision=1e-6, MaxIteration=3000)
        out, ms = timed(run, 3)
        it = out["iters"].to(torch.int64)
        nfg = (out["nf"].to(torch.int64) + out["ng"].to(torch.int64))
        # streaming model: trial 32n + objective 16n bytes per f+g evaluation pair, DY update 32n per iteration
        algo = float((8 * n * (3 * nfg + 4 * it)).sum())
        oo = O.defaults(precision=1e-6, maxit=3000, c2=0.45)
        dh, bh = d[:4096].cpu().numpy(), b[:4096].cpu().numpy()
        rate, S, dt, ref = cpu_rate(lambda s: O.solve_batch(O.CG, O.DIAGQUAD, np.zeros((min(s, 4096), n)), d=dh[:min(s, 4096)],
                                                            b=bh[:min(s, 4096)], opts=oo, nthreads=cores),
                                    4 * cores, args.cpu_seconds, cores)
        S = min(S, 4096)
        print(json.dumps({"config": "C3 CG Dai-Yuan, diagonal quadratics n=1024 kappa in [10,1000], batch 65536, Precision 1e-6",
                          "ms": ms, "iterations_per_s": float(it.sum()) / ms * 1e3, "iterations": int(it.sum()),
                          "converged_fraction": float((out["status"] == 0).double().mean()),
                          "algorithmic_GBps_streaming_model": algo / ms / 1e6,
                          "note": "every CG vector lives in registers: the kernel moves no HBM bytes per iteration",
                          "cpu_iterations_per_s": rate, "cpu": host, "cpu_sample": S,
                          "final_f_rel_err_max_vs_cpu": float(np.max(np.abs(out["f"][:S].cpu().numpy() - ref["f"][:S])
                                                                     / np.abs(ref["f"][:S])))}))

    if "c4" in args.configs:  # Batch 1024 problems n=4096, full dense BFGS, fixed K=20 iterations
        B, n, K = args.c4_batch, 4096, args.c4_iterations
        d, b = quad(B, n, 10.0, 100.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        ws = NLO.bfgs_workspace(B, n, dev)

        def run():
            x.zero_()
            return NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-12, MaxIteration=K - 1, ExactStep=0)
        out, ms = timed(run, 2)
        it = out["iters"].to(torch.int64)
        # rank-2 form: per update 8n^2 (q = H y) + 16n^2 (read H, write H'); the first update only writes 8n^2
        upd = torch.clamp(it - 1, min=0)
        algo = float((upd * 24 * n * n + (it > 0) * 8 * n * n).sum())
        # what the deferred form moves (n > 1024): one 8 n^2 read pass per update after the first, the first fold writes
        # 8 n^2 (H = a I was implicit), every later fold reads and writes 16 n^2
        # (updates 1..J run on the implicit H = a I: no pass over H at all until the first fold)
        J = 8
        U = torch.clamp(it - 1, min=0)
        folds = torch.div(U, J, rounding_mode="floor")
        moved = float((torch.clamp(U - J, min=0) * 8 * n * n + (folds > 0) * 8 * n * n
                       + torch.clamp(folds - 1, min=0) * 16 * n * n).sum())
        Bc = 2 * cores  # the CPU oracle at n=4096 is O(n^2) per update too (update_form 1); as-written form 0 is O(n^3)
        oo = O.defaults(precision=1e-12, maxit=K - 1, exact_step=0)
        dh, bh = d[:Bc].cpu().numpy(), b[:Bc].cpu().numpy()
        t = time.perf_counter()
        if args.cpu_seconds > 0:
            ref = O.solve_batch(O.BFGS, O.DIAGQUAD, np.zeros((Bc, n)), d=dh, b=bh, opts=oo, bfgs_form=1, nthreads=cores)
        else:  # profiling runs: no CPU leg
            ref = {"iters": np.full(Bc, np.nan), "f": np.full(Bc, np.nan)}
        dt = max(time.perf_counter() - t, 1e-9)
        print(json.dumps({"config": f"C4 dense BFGS (ExactStep=0), diagonal quadratics n=4096 kappa in [10,100], batch {B}, "
                                    f"fixed {K} iterations", "ms": ms, "iterations_per_s": float(it.sum()) / ms * 1e3,
                          "iterations": int(it.sum()),
                          "update_form": "rank-2, deferred: updates kept as vectors, folded into H every 8th iteration",
                          "algorithmic_GBps": algo / ms / 1e6, "frac_of_8TBps": algo / ms / 1e6 / HBM_PEAK_GBS,
                          "algorithmic_model": "SURVEY 8d rank-2 form: 24 n^2 B per update",
                          "moved_GBps_model": moved / ms / 1e6, "moved_frac_of_8TBps": moved / ms / 1e6 / HBM_PEAK_GBS,
                          "inverse_hessian_bytes": B * n * n * 8,
                          "cpu_iterations_per_s_rank2_form": float(ref["iters"].sum()) / dt, "cpu": host, "cpu_sample": Bc,
                          "final_f_rel_err_max_vs_cpu_rank2_reference_order": float(np.max(
                              np.abs(out["f"][:Bc].cpu().numpy() - ref["f"]) / np.maximum(np.abs(ref["f"]), 1e-300)))}))

    if "dgemm" in args.configs:  # My_dgemm / My_dgemm_T on the f64 matrix cores (fl_dgemm), square problems
        for n in (1024, 4096, 8192):
            A = torch.randn(n, n, dtype=torch.float64, device=dev)
            Bm = torch.randn(n, n, dtype=torch.float64, device=dev)
            for tA in (False, True):
                out, ms = timed(lambda: NLO.dgemm(A, Bm, transA=tA), 5)
                ref = (Bm @ (A.T if tA else A)) if n <= 4096 else None  # column-major C = op(A) B  <=>  row-major B^T-view product
                err = float((out - ref).abs().max() / (ref.abs().max())) if ref is not None else None
                flop = 2.0 * n ** 3
                print(json.dumps({"config": f"dgemm {'T' if tA else 'N'}N n={n} (fl_dgemm, v_mfma_f64_16x16x4_f64)", "ms": ms,
                                  "TFLOPs": flop / ms / 1e9, "peak_TFLOPs_datasheet": 78.6, "frac": flop / ms / 1e9 / 78.6,
                                  "max_rel_err_vs_torch_matmul": err}))

    if "dsyev" in args.configs:  # My_dsyev through the legacy symbol: host arrays in and out
        import ctypes as C
        from FortranLibrary.basic import FL
        dp = C.POINTER(C.c_double)
        for n in (64, 200, 1024, 2048, 4096):
            G = np.random.default_rng(n).standard_normal((n, n))
            A0 = np.asfortranarray(0.5 * (G + G.T))
            for job in (b"N", b"V"):
                S, w = A0.copy(order="F"), np.zeros(n)
                FL.__linearalgebra_MOD_my_dsyev(job, S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))  # warm-up
                S, w = A0.copy(order="F"), np.zeros(n)
                t = time.perf_counter()
                FL.__linearalgebra_MOD_my_dsyev(job, S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
                dt = time.perf_counter() - t
                t = time.perf_counter()
                wr = np.linalg.eigvalsh(A0) if job == b"N" else np.linalg.eigh(A0)[0]
                dtn = time.perf_counter() - t
                res = float(np.abs(A0 @ S - S * w[None, :]).max()) if job == b"V" else None
                orth = float(np.abs(S.T @ S - np.eye(n)).max()) if job == b"V" else None
                print(json.dumps({"config": f"My_dsyev '{job.decode()}' n={n} " + ("(Householder tridiagonalisation + multisection)" if job == b"N" else "(tridiagonalisation + inverse iteration + Cholesky-QR / Newton-Schulz + back-transformation)"),
                                  "ms_wall_incl_copies": dt * 1e3, "numpy_lapack_ms_on_this_host": dtn * 1e3,
                                  "eigenvalue_err_max_vs_lapack": float(np.abs(w - wr).max()),
                                  "residual_max": res, "orthogonality_max": orth, "norm1": float(np.abs(A0).sum(axis=1).max())}))

    if "c4gemm" in args.configs:  # the as-written two-matmul update on the f64 matrix cores
        B, n = 16, 4096
        T, E = NLO.reduction_geometry(n)
        ld = T * E
        H = torch.zeros(B, n, ld, dtype=torch.float64, device=dev)
        H[:, torch.arange(n), torch.arange(n)] = 1.0
        H[:, :, :n] += 0.01 * torch.randn(B, n, n, dtype=torch.float64, device=dev)
        s = torch.randn(B, n, dtype=torch.float64, device=dev)
        y = s * (1 + torch.rand(B, n, dtype=torch.float64, device=dev)) 
        ws = NLO.bfgs_update_gemm(H, s, y)
        out, ms = timed(lambda: NLO.bfgs_update_gemm(H, s, y, workspace_=ws), 3)
        flop = 4.0 * n ** 3 * B
        PEAK = 78.6  # TFLOP/s FP64 matrix, AMD datasheet; not in the local guides
        print(json.dumps({"config": f"C4-gemm BFGS update as written (2 x n^3 matmul on v_mfma_f64_16x16x4_f64), n=4096, {B} problems",
                          "ms": ms, "TFLOPs": flop / ms / 1e9, "peak_TFLOPs_datasheet": PEAK, "frac": flop / ms / 1e9 / PEAK,
                          "ms_per_problem_update": ms / B,
                          "rank2_form_ms_per_problem_update_at_5.6TBps": 24.0 * n * n * 8 / 5.6e12 * 1e3 / 8}))

    if "c1" in args.configs:  # BASELINE config 1: one Rosenbrock n=10 problem, BFGS, standard start -- plumbing, not throughput
        n = 10
        x0 = torch.full((1, n), -1.2, dtype=torch.float64, device=dev)
        x0[:, 1::2] = 1.0
        for es, label in ((0, "ExactStep=0"), (20, "ExactStep=20 (analytic Hessian refresh)")):
            for B in (1, 65536):
                xb = x0.expand(B, n).contiguous()
                x = torch.empty_like(xb)
                ws = NLO.bfgs_workspace(B, n, dev, NLO.default_options(NLO.BFGS_, ExactStep=es))

                def run():
                    x.copy_(xb)
                    return NLO.BFGS(NLO.ROSENBROCK, x, workspace_=ws, ExactStep=es)
                out, ms = timed(run, 3)
                ref = O.solve_batch(O.BFGS, O.ROSENBROCK, x0.cpu().numpy(), opts=O.defaults(exact_step=es), bfgs_form=1)
                t = time.perf_counter()
                for _ in range(20):
                    O.solve_batch(O.BFGS, O.ROSENBROCK, x0.cpu().numpy(), opts=O.defaults(exact_step=es), bfgs_form=1)
                cpu_ms = (time.perf_counter() - t) / 20 * 1e3
                print(json.dumps({"config": f"C1 BFGS {label}, Rosenbrock n=10 standard start, batch {B} (replicas)", "ms": ms,
                                  "iterations": int(out["iters"][0]), "nf": int(out["nf"][0]), "ng": int(out["ng"][0]),
                                  "f": float(out["f"][0]), "max_abs_x_minus_1": float((x - 1).abs().max()),
                                  "problems_per_s": B / ms * 1e3, "cpu_ms_one_problem_one_core": cpu_ms,
                                  "cpu_iterations": int(ref["iters"][0]), "cpu_f": float(ref["f"][0])}))

    if "newton" in args.configs:  # NewtonRaphson with the analytic Hessian, dense Cholesky per iteration
        for B, n in ((4096, 256), (256, 1024)):
            x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
            NLO.synth_uniform(SEED, x0, 0.9, 1.1)
            x = torch.empty_like(x0)
            ws = NLO.bfgs_workspace(B, n, dev, NLO.default_options(NLO.NEWTON_), NLO.NEWTON_)

            def run():
                x.copy_(x0)
                return NLO.NewtonRaphson(NLO.ROSENBROCK, x, workspace_=ws, Precision=1e-10, MaxIteration=100)
            out, ms = timed(run, 2)
            it = out["iters"].to(torch.int64)
            flop = float(it.sum()) * (n ** 3 / 3.0 + 2.0 * n * n)  # Cholesky + two triangular solves per iteration
            byts = float(it.sum()) * (n * n * 8 * 2.0)              # Hessian written once, factor read back once (lower bound)
            Bc = min(B, 2 * cores)
            t = time.perf_counter()
            ref = O.solve_batch(4, O.ROSENBROCK, x0[:Bc].cpu().numpy(), opts=O.defaults(precision=1e-10, maxit=100), nthreads=cores)
            dt = time.perf_counter() - t
            print(json.dumps({"config": f"NewtonRaphson (analytic Hessian, Cholesky solve), Rosenbrock n={n}, batch {B}", "ms": ms,
                              "iterations": int(it.sum()), "iterations_per_s": float(it.sum()) / ms * 1e3,
                              "converged_fraction": float((out["status"] == 0).double().mean()),
                              "GFLOPs_cholesky_model": flop / ms / 1e6, "GBps_lower_bound": byts / ms / 1e6,
                              "max_abs_x_minus_1": float((x - 1).abs().max()),
                              "cpu_iterations_per_s": float(ref["iters"].sum()) / dt, "cpu": host, "cpu_sample": Bc}))

    if "dense" in args.configs:  # My_dposv / My_dpotri for batches (LA.f90:719-730, 798-812)
        for B, n in ((4096, 256), (256, 1024), (16, 4096)):
            T, E = NLO.reduction_geometry(n)
            ld = T * E
            g = torch.randn(B, n, n, dtype=torch.float64, device=dev)
            spd = g @ g.transpose(1, 2) / n + torch.eye(n, dtype=torch.float64, device=dev)
            A0 = torch.zeros(B, n, ld, dtype=torch.float64, device=dev)
            A0[:, :, :n] = spd  # symmetric: row / column major agree
            rhs0 = torch.randn(B, n, dtype=torch.float64, device=dev)
            A, rhs = torch.empty_like(A0), torch.empty_like(rhs0)

            def run_sv():
                A.copy_(A0)
                rhs.copy_(rhs0)
                return NLO.dposv(A, rhs)

            def run_copy():
                A.copy_(A0)
                rhs.copy_(rhs0)
                return None
            _, ms_copy = timed(run_copy, 3)
            info, ms = timed(run_sv, 3)
            res = float(((spd @ rhs.unsqueeze(2)).squeeze(2) - rhs0).abs().max())

            def run_tri():
                A.copy_(A0)
                return NLO.dpotri(A)
            info2, ms2 = timed(run_tri, 3)
            err = float((A[:, :, :n] @ spd - torch.eye(n, dtype=torch.float64, device=dev)).abs().max())
            print(json.dumps({"config": f"dense SPD kernels n={n}, batch {B}", "dposv_ms": ms - ms_copy,
                              "dposv_GFLOPs": B * (n ** 3 / 3.0 + 2.0 * n * n) / (ms - ms_copy) / 1e6,
                              "dposv_max_residual": res, "dpotri_ms": ms2 - ms_copy,
                              "dpotri_GFLOPs": B * (n ** 3) / (ms2 - ms_copy) / 1e6, "dpotri_max_abs_AinvA_minus_I": err,
                              "all_info_zero": bool((info == 0).all() and (info2 == 0).all())}))

    if "big" in args.configs:  # beyond the register path (n > 4096): vectors in HBM, one workgroup per problem
        for B, n in ((4096, 8192), (1024, 65536)):
            m = 10
            d, b = quad(B, n, 10.0, 1000.0)
            x = torch.zeros(B, n, dtype=torch.float64, device=dev)
            ws = NLO.workspace(B, n, m, dev)

            def run():
                x.zero_()
                return NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-6, MaxIteration=3000, Memory=m)
            out, ms = timed(run, 2)
            it = out["iters"].to(torch.int64)
            nfg = out["nf"].to(torch.int64) + out["ng"].to(torch.int64)
            k = torch.clamp(it - 1, min=0)
            part = torch.clamp(k, max=m)
            cnt = part * (part + 1) // 2 + torch.clamp(k - m, min=0) * m
            algo = float((8 * n * (4 * cnt + 2 * k)).sum())            # SURVEY 8d two-loop bytes
            moved = float((8 * n * (6 * cnt + 10 * k + 3 * nfg)).sum())  # what the pass-wise form streams (fl_big.hpp)
            Bc = min(B, cores)
            oo = O.defaults(precision=1e-6, maxit=3000, memory=m)
            t = time.perf_counter()
            ref = O.solve_batch(O.LBFGS, O.DIAGQUAD, np.zeros((Bc, n)), d=d[:Bc].cpu().numpy(), b=b[:Bc].cpu().numpy(), opts=oo,
                                nthreads=cores)
            dt = time.perf_counter() - t
            print(json.dumps({"config": f"L-BFGS m=10 beyond the register path: diagonal quadratics n={n}, batch {B}, Precision 1e-6",
                              "ms": ms, "iterations_per_s": float(it.sum()) / ms * 1e3, "iterations": int(it.sum()),
                              "converged_fraction": float((out["status"] == 0).double().mean()),
                              "algorithmic_GBps_two_loop": algo / ms / 1e6, "moved_GBps_model": moved / ms / 1e6,
                              "cpu_iterations_per_s": float(ref["iters"].sum()) / dt, "cpu": host, "cpu_sample": Bc,
                              "final_f_rel_err_max_vs_cpu": float(np.max(np.abs(out["f"][:Bc].cpu().numpy() - ref["f"])
                                                                         / np.abs(ref["f"])))}))

    if "bigbfgs" in args.configs:  # dense BFGS beyond the register path: inverse Hessian in HBM, one workgroup per problem
        for B, n, K in ((256, 8192, 20), (64, 16384, 12)):
            d, b = quad(B, n, 10.0, 100.0)
            x = torch.zeros(B, n, dtype=torch.float64, device=dev)
            ws = NLO.bfgs_workspace(B, n, dev)

            def run():
                x.zero_()
                return NLO.BFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, Precision=1e-12, MaxIteration=K - 1, ExactStep=0)
            out, ms = timed(run, 1)
            it = out["iters"].to(torch.int64)
            U = torch.clamp(it - 1, min=0)
            folds = torch.div(U, 8, rounding_mode="floor")
            moved = float((torch.clamp(U - 8, min=0) * 8 * n * n + (folds > 0) * 8 * n * n
                           + torch.clamp(folds - 1, min=0) * 16 * n * n).sum())
            print(json.dumps({"config": f"dense BFGS (ExactStep=0) beyond the register path: quadratics n={n}, batch {B}, {K} iterations",
                              "ms": ms, "iterations_per_s": float(it.sum()) / ms * 1e3, "iterations": int(it.sum()),
                              "moved_GBps_model": moved / ms / 1e6, "inverse_hessian_bytes": B * n * n * 8}))

    if "rci" in args.configs:  # the generic-objective path: the headline workload through reverse communication, torch objective
        B, n, m = args.rci_batch, 1024, 10
        d, b = quad(B, n, 10.0, 1000.0)
        x = torch.zeros(B, n, dtype=torch.float64, device=dev)
        ws = NLO.workspace(B, n, m, dev)
        kw = dict(Precision=1e-6, MaxIteration=3000, Memory=m)

        def fused():
            x.zero_()
            return NLO.LBFGS(NLO.DIAGQUAD, x, d, b, workspace_=ws, **kw)
        outf, msf = timed(fused, 3)
        xf = x.clone()
        calls = {"n": 0, "rows": 0}
        cache = {}

        def f_of(xx, dd, bb):
            dx = dd * xx
            return 0.5 * (dx * xx).sum(1) - (bb * xx).sum(1), dx - bb

        xfull = torch.zeros(B, n, dtype=torch.float64, device=dev)

        def fun(xx, req=None, ids=None, epoch=None):
            calls["n"] += 1
            calls["rows"] += xx.shape[0]
            if args.rci_fixed_shape:
                # an objective whose row sums do NOT depend on how many rows the caller hands over: always evaluated on the full
                # [B, n] buffers (torch picks its reduction strategy from the shape) -- attributes the 4-iteration difference of
                # `compact` at the headline size to the caller's reduction or to the library
                if ids is None:
                    return f_of(xx, d, b)
                i = ids.long()
                xfull.index_copy_(0, i, xx)
                ff, gg_ = f_of(xfull, d, b)
                return ff[i], gg_[i]
            if ids is None:
                return f_of(xx, d, b)
            if cache.get("epoch") != epoch:  # the list of running problems changed: gather their data once
                i = ids.long()
                cache.update(epoch=epoch, d=d[i], b=b[i])
            return f_of(xx, cache["d"], cache["b"])
        first = None
        for mode in args.rci_modes.split(","):
            x.zero_()
            calls.update(n=0, rows=0)
            cache.clear()
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = NLO.minimize_rci(NLO.LBFGS_, x, fun, mode=mode, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            it = int(out["iters"].to(torch.int64).sum())
            if first is None:
                first = (x.clone(), out["iters"].clone(), out["nf"].clone())
            print(json.dumps({"config": f"RCI ({mode}) L-BFGS m=10, diagonal quadratics n=1024, batch {B}, torch objective"
                                        + (" evaluated on fixed-shape buffers" if args.rci_fixed_shape else ""), "ms": dt * 1e3,
                              "iterations_per_s": it / dt, "iterations": it, "steps": out["steps"], "objective_calls": calls["n"],
                              "objective_rows_evaluated": calls["rows"], "trials": int(out["nf"].to(torch.int64).sum()),
                              "rows_per_trial": calls["rows"] / max(1, int(out["nf"].to(torch.int64).sum())),
                              "fused_kernel_ms": msf, "fused_iterations_per_s": float(outf["iters"].to(torch.int64).sum()) / msf * 1e3,
                              "same_bits_as_the_first_mode": bool(torch.equal(x, first[0]) and torch.equal(out["iters"], first[1]) and torch.equal(out["nf"], first[2])),
                              "max_rel_x_diff_vs_fused": float(((x - xf).norm(dim=1) / xf.norm(dim=1)).max()),
                              "slowdown_vs_fused": dt * 1e3 / msf}))

    if "rcigraph" in args.configs:  # the round-trip budget of reverse communication per step: eager launches vs one HIP graph per round
        n, m = 1024, 10
        for B in (256, 2048, 16384):
            d, b = quad(B, n, 10.0, 1000.0)
            x = torch.zeros(B, n, dtype=torch.float64, device=dev)

            def fun(xx, req=None):
                dx = d * xx
                return 0.5 * (dx * xx).sum(1) - (b * xx).sum(1), dx - b
            first = None
            for mode in ("full", "graph"):
                NLO.minimize_rci(NLO.LBFGS_, x.zero_(), fun, mode=mode, Precision=1e-6, MaxIteration=20, Memory=m, check_every=32)  # warm-up
                x.zero_()
                torch.cuda.synchronize()
                t = time.perf_counter()
                out = NLO.minimize_rci(NLO.LBFGS_, x, fun, mode=mode, Precision=1e-6, MaxIteration=3000, Memory=m, check_every=32)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                it = int(out["iters"].to(torch.int64).sum())
                if first is None:
                    first = x.clone()
                print(json.dumps({"config": f"RCI round trip ({mode}) L-BFGS m=10, diagonal quadratics n=1024, batch {B}, torch objective",
                                  "ms": dt * 1e3, "steps": out["steps"], "us_per_step": dt * 1e6 / max(1, out["steps"]),
                                  "iterations_per_s": it / dt, "same_bits_as_full": bool(torch.equal(x, first))}), flush=True)

    if "coop" in args.configs:  # ONE problem of n = 2^20 (the reference's callers: one problem of any dim) by reverse communication
        n, m = 1 << 20, 10
        i = torch.arange(n, dtype=torch.float64, device=dev)
        d1 = (1.0 + 99.0 * i / (n - 1)).unsqueeze(0)
        b1 = torch.sin(i + 1.0).unsqueeze(0)

        def fun(xx):
            dx = d1 * xx
            return 0.5 * (dx * xx).sum(1) - (b1 * xx).sum(1), dx - b1
        for want in ("1", "auto"):
            if want == "auto":
                os.environ.pop("FL_COOP_GROUPS", None)
            else:
                os.environ["FL_COOP_GROUPS"] = want
            x = torch.zeros(1, n, dtype=torch.float64, device=dev)
            NLO.minimize_rci(NLO.LBFGS_, x, fun, Precision=1e-6, MaxIteration=5, Memory=m)  # warm-up
            x.zero_()
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = NLO.minimize_rci(NLO.LBFGS_, x, fun, Precision=1e-6, MaxIteration=args.coop_iterations, Memory=m)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(json.dumps({"config": f"one L-BFGS problem, n = 2^20, diagonal quadratic kappa 100, reverse communication with a torch objective, "
                                        f"workgroups per problem: {out['cooperative_groups']}", "ms": dt * 1e3, "steps": out["steps"],
                              "iterations": int(out["iters"][0]), "iterations_per_s": float(out["iters"][0]) / dt,
                              "ms_per_step": dt * 1e3 / max(1, out["steps"]), "f": float(out["f"][0]), "gnorm": float(out["gg"][0]) ** 0.5}))
        os.environ.pop("FL_COOP_GROUPS", None)

    if "c5" in args.configs:  # aug-Lagrangian wrapping L-BFGS, batch 8192 n=512, 8 equality constraints
        B, n, M, m = args.c5_batch, 512, 8, 10
        d, b = quad(B, n, 2.0, 10.0)
        x0 = torch.empty(B, n, dtype=torch.float64, device=dev)
        NLO.synth_uniform(SEED + 7, x0, 0.05, 0.15)
        x = torch.empty_like(x0)
        ws = NLO.workspace(B, n, m, dev)

        def run():
            x.copy_(x0)
            return NLO.AugmentedLagrangian(NLO.DIAGQUAD, x, M, d, b, UnconstrainedSolver="LBFGS", workspace_=ws,
                                           Precision=1e-10, Memory=m)
        out, ms = timed(run, 2)
        it = out["iters"].to(torch.int64)
        oo = O.defaults(precision=1e-10, memory=m)
        Bc = 4 * cores
        t = time.perf_counter()
        if args.cpu_seconds > 0:
            ref = O.auglag_batch(O.LBFGS, O.DIAGQUAD, x0[:Bc].cpu().numpy(), M, d=d[:Bc].cpu().numpy(), b=b[:Bc].cpu().numpy(),
                                 opts=oo, nthreads=cores)
        else:  # profiling / A-B runs: no CPU leg
            ref = {"iters": np.full(Bc, np.nan), "f": np.full(Bc, np.nan)}
        dt = time.perf_counter() - t
        fx = (0.5 * (d * x * x).sum(1) - (b * x).sum(1))[:Bc].cpu().numpy()
        nfp = out["nf"].double()
        print(json.dumps({"config": f"C5 augmented Lagrangian + L-BFGS m=10, n=512, 8 block-sphere constraints, batch {B}, "
                                    "Precision 1e-10", "ms": ms, "f_evals_per_problem_min_mean_max": [float(nfp.min()), float(nfp.mean()), float(nfp.max())], "inner_iterations_per_s": float(it.sum()) / ms * 1e3,
                          "inner_iterations": int(it.sum()), "outer_iterations_mean": float(out["outer"].double().mean()),
                          "f_evals": int(out["nf"].to(torch.int64).sum()),
                          "converged_fraction": float((out["status"] == 0).double().mean()),
                          "cnorm_max": float(out["cnorm2"].max().sqrt()),
                          "cpu_inner_iterations_per_s": float(ref["iters"].sum()) / dt, "cpu": host, "cpu_sample": Bc,
                          "final_f_rel_err_max_vs_cpu": float(np.max(np.abs(fx - ref["f"]) / np.abs(ref["f"])))}))


if __name__ == "__main__":
    main()
