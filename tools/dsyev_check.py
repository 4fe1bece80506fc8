"""My_dsyev('V') on the GPU box: residual / orthogonality / which path, and the time, per case and size.
usage: python3 tools/dsyev_check.py [n ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import torch  # noqa: E402  (before libFL.so: one HIP runtime)
from FortranLibrary.basic import FL  # noqa: E402

dp = C.POINTER(C.c_double)
FL.fl_dsyev_vectors_workspace_bytes.restype = C.c_size_t


def cases(n):
    rng = np.random.default_rng(100 + n)
    G = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(G)
    m = n // 2
    rep = np.repeat(np.arange(1, n // 8 + 2), 8)[:n].astype(float)
    out = {
        "random": 0.5 * (G + G.T),
        "identity": np.eye(n),
        "diagonal": np.diag(np.arange(n, 0, -1.0)),
        "tridiagonal": np.diag(np.full(n, 2.0)) + np.diag(np.full(n - 1, -1.0), 1) + np.diag(np.full(n - 1, -1.0), -1),
        "projector": Q[:, : max(1, n // 3)] @ Q[:, : max(1, n // 3)].T,
        "graded": (Q * np.logspace(0, -12, n)[None, :]) @ Q.T,
        "zero": np.zeros((n, n)),
        "wilkinson": np.diag(np.abs(np.arange(n) - m).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1),
        "clusters": (Q * rep[None, :]) @ Q.T,
        "near_clusters": (Q * (rep + 1e-13 * rng.standard_normal(n))[None, :]) @ Q.T,
        "tiny": 1e-200 * 0.5 * (G + G.T),
        "huge": 1e+200 * 0.5 * (G + G.T),
    }
    return {k: 0.5 * (v + v.T) for k, v in out.items()}


def direct(A):
    """fl_dsyev_vectors itself: return code and its own quality figures"""
    n = A.shape[0]
    dev = torch.device("cuda:0")
    Ad = torch.tensor(np.asfortranarray(np.tril(A)).T.copy(), device=dev)  # column-major n x n
    w = torch.zeros(n, dtype=torch.float64, device=dev)
    wsb = FL.fl_dsyev_vectors_workspace_bytes(n)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
    q = (C.c_double * 3)()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(Ad.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()), C.c_void_p(ws.data_ptr()),
                             C.c_size_t(wsb), q, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return rc, list(q), (t1 - t0) * 1e3, Ad.cpu().numpy().T, w.cpu().numpy()


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [5, 64, 200, 1024]
    worst = 0.0
    for n in sizes:
        for name, A in cases(n).items():
            norm = max(np.abs(A).sum(axis=1).max(), 1e-300)
            direct(A)  # warm-up
            rc, q, ms, V, w = direct(A)
            S = np.asfortranarray(np.tril(A))
            wl = np.zeros(n)
            t0 = time.perf_counter()
            FL.__linearalgebra_MOD_my_dsyev(b"V", S.ctypes.data_as(dp), wl.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
            tl = (time.perf_counter() - t0) * 1e3
            res = np.abs(A @ S - S * wl[None, :]).max() / norm
            orth = np.abs(S.T @ S - np.eye(n)).max()
            ref = np.linalg.eigvalsh(A)
            ev = np.abs(wl - ref).max() / norm
            resd = np.abs(A @ V - V * w[None, :]).max() / norm if rc == 0 else float("nan")
            worst = max(worst, res, orth)
            print(f"n={n:5d} {name:14s} rc={rc} passes={int(q[2])} quality(orth {q[0]:.1e}, res {q[1]:.1e}) device {ms:8.2f} ms | "
                  f"legacy: {tl:8.2f} ms resid/norm {res:.2e} (direct {resd:.2e}) orth {orth:.2e} eig {ev:.2e}", flush=True)
    print("worst", worst)


if __name__ == "__main__":
    main()
