#!/usr/bin/env python3
"""Writes tests/golden/bfgs_gemm_4096.npz: the BFGS update AS THE REFERENCE WRITES IT (NO.f90:958-962: two dense
matmuls) computed by the CPU oracle (update_form 0, sequential sums) for ONE asymmetric n = 4096 problem -- 1.4e11
flops on one core, a minute or two -- condensed to sampled rows / columns and probe products (tests/la_cases.py's
digest).  tests/test_gpu_bfgs_gemm.py holds the f64-MFMA kernel to it.  Inputs are functions of the seed below."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import la_cases as LC
import oracle_lib as O


def inputs(n):
    r = LC.rng(f"bfgs_gemm{n}")
    H = r.standard_normal((n, n))  # asymmetric on purpose; H[col, row] (column-major rows of the array)
    s = r.standard_normal(n)
    y = s * r.uniform(0.5, 2.0, n) + 0.1 * r.standard_normal(n)
    return H, s, y


def main():
    n = 4096
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.flo_bfgs_update.argtypes = [C.c_int, dp, dp, dp, C.c_int]
    lib.flo_set_sum_mode(O.SEQ, 64, 2)
    H, s, y = inputs(n)
    ref = np.ascontiguousarray(H).copy()
    t = time.time()
    lib.flo_bfgs_update(n, ref.ctypes.data_as(dp), s.ctypes.data_as(dp), y.ctypes.data_as(dp), 0)
    print(f"oracle form 0, n = {n}: {time.time() - t:.1f} s")
    out = {}
    LC.store_matrix(lambda k, v: out.__setitem__(k, np.asarray(v)), "Hnew", ref, n)
    out["scale"] = np.abs(ref).max()
    out["input_sums"] = np.array([H.sum(), s.sum(), y.sum()])
    path = os.path.join(ROOT, "tests", "golden", "bfgs_gemm_4096.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
