#!/usr/bin/env python3
"""BASELINE config 5 (8192 x n = 512, M = 8) under the staged launches: kernel time by the two thresholds (problems left when
the one-helper and the three-helper stages take over; FL_STAGE_T2 / FL_STAGE_T4, read per call).  -> profiles/r04/c5_staged.txt"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fortran-library_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from geometry_by_batch import workload, timed
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
run, _ = workload("c5", B)
os.environ["FL_AUG_STAGED"] = "0"
_, ms = timed(run, 5)
print(json.dumps({"batch": B, "single_launch_ms": round(ms, 2)}), flush=True)
os.environ["FL_AUG_STAGED"] = "1"
for t2 in (768, 1024, 1536, 2048):
    for t4 in (192, 256, 384, 512, 768):
        if t4 > t2: continue
        os.environ["FL_STAGE_T2"], os.environ["FL_STAGE_T4"] = str(t2), str(t4)
        _, ms = timed(run, 5)
        print(json.dumps({"batch": B, "t2": t2, "t4": t4, "ms": round(ms, 2)}), flush=True)
