"""My_dsyev twice per size through the legacy symbol (for rocprofv3 --kernel-trace --stats: time per kernel).
usage: python3 tools/dsyev_once.py [N|V] [n ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd"))
from FortranLibrary.basic import FL  # noqa: E402

dp = C.POINTER(C.c_double)
job = b"N"
args = sys.argv[1:]
if args and args[0] in ("N", "V"):
    job = args.pop(0).encode()
for n in [int(a) for a in args] or [1024]:
    G = np.random.default_rng(n).standard_normal((n, n))
    A0 = np.asfortranarray(0.5 * (G + G.T))
    for rep in range(2):
        S, w = A0.copy(order="F"), np.zeros(n)
        t = time.perf_counter()
        FL.__linearalgebra_MOD_my_dsyev(job, S.ctypes.data_as(dp), w.ctypes.data_as(dp), C.byref(C.c_int(n)), C.c_int(1))
        print(job.decode(), n, rep, (time.perf_counter() - t) * 1e3, "ms", flush=True)
