"""Stress of fl_dsyev_vectors (My_dsyev 'V' fast path) on structured and random matrices: how often its device-side check
sends a matrix to the Jacobi fallback, and the worst residual / orthogonality of what it returns.
usage: python3 tools/dsyev_stress.py [seed] [cases]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
import torch  # noqa: E402
from FortranLibrary.basic import FL  # noqa: E402

FL.fl_dsyev_vectors_workspace_bytes.restype = C.c_size_t
dev = torch.device("cuda:0")


def family(rng, kind, n):
    G = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(G)
    if kind == 0:
        return 0.5 * (G + G.T)
    if kind == 1:  # prescribed spectrum with random multiplicities
        vals = rng.choice(rng.standard_normal(max(1, n // 4)), n)
        return (Q * vals[None, :]) @ Q.T
    if kind == 2:  # rank deficient
        r = max(1, n // 5)
        return G[:, :r] @ G[:, :r].T
    if kind == 3:  # Hilbert-like, very ill conditioned
        i = np.arange(n)
        return 1.0 / (i[:, None] + i[None, :] + 1.0)
    if kind == 4:  # arrow matrix
        A = np.diag(rng.standard_normal(n))
        A[-1, :] = A[:, -1] = rng.standard_normal(n)
        return A
    if kind == 5:  # block diagonal with tiny coupling
        A = np.zeros((n, n))
        h = n // 2
        A[:h, :h] = 0.5 * (G[:h, :h] + G[:h, :h].T)
        A[h:, h:] = A[:n - h, :n - h] if n - h == h else 0.5 * (G[h:, h:] + G[h:, h:].T)
        A += 1e-15 * 0.5 * (G + G.T)
        return A
    if kind == 6:  # graded diagonal + perturbation
        return np.diag(np.logspace(0, -14, n)) + 1e-10 * 0.5 * (G + G.T)
    if kind == 7:  # glued Wilkinson matrices
        m = max(3, min(21, n // 3 * 2 + 1))
        W = np.diag(np.abs(np.arange(m) - m // 2).astype(float)) + np.diag(np.ones(m - 1), 1) + np.diag(np.ones(m - 1), -1)
        A = np.zeros((n, n))
        p = 0
        while p + m <= n:
            A[p:p + m, p:p + m] = W
            if p > 0:
                A[p, p - 1] = A[p - 1, p] = 1e-8
            p += m
        return A
    if kind == 8:  # Toeplitz
        c = rng.standard_normal(n)
        i = np.arange(n)
        return c[np.abs(i[:, None] - i[None, :])]
    # clusters at several scales
    vals = np.concatenate([1.0 + 1e-12 * rng.standard_normal(n // 3), 2.0 + 1e-8 * rng.standard_normal(n // 3),
                           rng.standard_normal(n - 2 * (n // 3))])
    return (Q * vals[None, :]) @ Q.T


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rng = np.random.default_rng(seed)
    fallback = 0
    worst = {"resid": 0.0, "orth": 0.0, "eig": 0.0}
    by_kind = {}
    for t in range(cases):
        kind = t % 10
        n = int(rng.choice([2, 3, 5, 8, 17, 33, 64, 65, 100, 129, 200, 257, 400, 513, 700]))
        A = family(rng, kind, n)
        A = 0.5 * (A + A.T)
        norm = max(np.abs(A).sum(axis=1).max(), 1e-300)
        Ad = torch.tensor(np.tril(A).T.copy(), device=dev)
        w = torch.zeros(n, dtype=torch.float64, device=dev)
        wsb = FL.fl_dsyev_vectors_workspace_bytes(n)
        ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
        q = (C.c_double * 3)()
        rc = FL.fl_dsyev_vectors(C.c_int(n), C.c_void_p(Ad.data_ptr()), C.c_int(n), C.c_void_p(w.data_ptr()), C.c_void_p(ws.data_ptr()),
                                 C.c_size_t(wsb), q, None)
        torch.cuda.synchronize()
        if rc != 0:
            fallback += 1
            by_kind[kind] = by_kind.get(kind, 0) + 1
            print(f"case {t}: kind {kind} n={n}: rc={rc} quality {list(q)}", flush=True)
            continue
        V, wh = Ad.cpu().numpy().T, w.cpu().numpy()
        resid = np.abs(A @ V - V * wh[None, :]).max() / norm
        orth = np.abs(V.T @ V - np.eye(n)).max()
        eig = np.abs(wh - np.linalg.eigvalsh(A)).max() / norm
        if resid > 1e-13 or orth > 1e-13 or eig > 1e-13:
            print(f"case {t}: kind {kind} n={n}: resid {resid:.2e} orth {orth:.2e} eig {eig:.2e} passes {q[2]}", flush=True)
        worst["resid"] = max(worst["resid"], resid)
        worst["orth"] = max(worst["orth"], orth)
        worst["eig"] = max(worst["eig"], eig)
    print(f"{cases} cases: fallback {fallback} {by_kind}; worst residual/||A||_1 {worst['resid']:.2e}, orthogonality {worst['orth']:.2e}, "
          f"eigenvalue error/||A||_1 {worst['eig']:.2e}")


if __name__ == "__main__":
    main()
