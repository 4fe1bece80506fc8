#!/usr/bin/env python3
"""Condense the rocprofv3 runs of the BASELINE configs (tools/profile.sh with PROG=tools/bench_configs.py) into
profiles/<round>/configs_pmc.txt and copy the per-kernel CSV rows next to it.
usage: python tools/config_profiles.py r02 c2 c3 c4 c5 c4gemm"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary as P

MATCH = {"c4gemm": "bfgs_gemm_kernel", "dgemm": "dgemm_kernel"}


def main():
    rnd, cfgs = sys.argv[1], sys.argv[2:]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out = ["# BASELINE configs under rocprofv3: kernel trace (average duration of the dominant kernel) and separate --pmc passes for",
           "# FETCH_SIZE / WRITE_SIZE (KiB per launch of that kernel); memory-side bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950",
           "# correction, MI355X_MICROARCH.md, HBM).  Made by PROG=tools/bench_configs.py PMC_ONLY_TRAFFIC=1 tools/profile.sh",
           f"# {rnd}_<cfg> <cfg>, condensed by tools/config_profiles.py; the kernel's CSV rows are next to this file.", ""]
    for cfg in cfgs:
        match = MATCH.get(cfg, "solve_kernel")
        root = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{cfg}")
        ks = P.kernel_stats(root)
        cs, info = P.counters(root, match)
        hits = [(k, v) for k, v in ks.items() if match in k]
        if not hits or "FETCH_SIZE" not in cs:
            out.append(f"== {cfg}: no data under {root}")
            continue
        name, st = max(hits, key=lambda kv: kv[1]["total_ms"])
        tb = (2 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024
        lines = [json.loads(l) for l in open(os.path.join(root, "bench.json")) if l.startswith("{")]
        b = lines[-1]
        out.append(f"== {cfg}: {b['config']}")
        out.append(f"   kernel {name}  (vgpr {info['vgpr']} [rocprof units], lds {info['lds']} B, workgroup {info['wg']})")
        out.append(f"   kernel trace: {st['calls']} calls, average {st['avg_ms']:.3f} ms, {st['pct']}% of the GPU time of the run; "
                   f"HIP events in the program: {b['ms']:.3f} ms")
        out.append(f"   FETCH_SIZE {cs['FETCH_SIZE'][0]:.6g} KiB, WRITE_SIZE {cs['WRITE_SIZE'][0]:.6g} KiB per launch -> memory side "
                   f"{tb / 1e9:.3f} GB per launch = {tb / st['avg_ms'] / 1e6:.0f} GB/s = {tb / st['avg_ms'] / 1e6 / 8000:.3f} of 8 TB/s")
        keep = ("iterations", "iterations_per_s", "inner_iterations_per_s", "TFLOPs", "frac", "moved_GBps_model",
                "algorithmic_GBps", "converged_fraction")
        out.append("   program: " + json.dumps({k: v for k, v in b.items() if k in keep}))
        shutil.copy(st["file"], os.path.join(dst, f"{cfg}_kernel_stats.csv"))
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            with open(cs[c][2]) as fi, open(os.path.join(dst, f"{cfg}_pmc_{c}.csv"), "w", newline="") as fo:
                rd = csv.DictReader(fi)
                wr = csv.DictWriter(fo, fieldnames=rd.fieldnames)
                wr.writeheader()
                for r in rd:
                    if match in r["Kernel_Name"]:
                        wr.writerow(r)
        shutil.copy(os.path.join(root, "bench.json"), os.path.join(dst, f"{cfg}_bench.json"))
    txt = "\n".join(out) + "\n"
    open(os.path.join(dst, "configs_pmc.txt"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
