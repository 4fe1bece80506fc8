#!/usr/bin/env python3
"""Wall time of the Fortran smoke test (19 one-problem solves through the legacy symbols with host callbacks,
tests/fortran/test_nlopt.f90) under two builds of libFL.so: usage: python tools/legacy_latency.py [other_libFL.so]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "fortran-library_amd", "fortran", "test_nlopt")
libs = [("in-tree libFL.so", None)] + [(p, p) for p in sys.argv[1:]]
for rep in range(3):
    for name, lib in libs:
        env = dict(os.environ)
        if lib:
            env["LD_PRELOAD"] = os.path.abspath(lib)
        t = time.perf_counter()
        subprocess.run([exe], env=env, stdout=subprocess.DEVNULL, check=True)
        print(f"{name}: {time.perf_counter() - t:.3f} s", flush=True)
