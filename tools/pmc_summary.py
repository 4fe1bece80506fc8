#!/usr/bin/env python3
"""Summarise the rocprofv3 output of tools/profile.sh: per-kernel average duration (kernel trace) and PMC counter
sums per dispatch -- and, with --record, write the record bench.py's `roofline` reads (profiles/traffic.json) and
copy the CSVs it was derived from under profiles/<round>/ so that every number of the bench line can be re-derived
from committed files.

    python tools/pmc_summary.py gpurun_out/prof_<tag>                       # print
    python tools/pmc_summary.py gpurun_out/prof_<tag> --record r02 <tag>    # + profiles/traffic.json, profiles/r02/<tag>_*

Units (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of a
16-byte-per-lane streaming read, so memory-side bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  SQ_WAVE_CYCLES, SQ_WAIT_*,
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_INSTS_* count wave-instructions.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _newest(files):
    """gpurun merges every run's output into the same directory: of several runs' files keep the most recent"""
    files = list(files)
    return [max(files, key=os.path.getmtime)] if files else []


def kernel_stats(root):
    out = {}
    for f in _newest(glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                              "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"]), "file": f}
    return out


KERNELS_PER_STEP = 1  # --kernels-per-step K: a step is K dispatches of matching kernels (BASELINE config 5 staged: three)


def counters(root, match):
    """{counter: (sum per STEP of kernels whose name contains `match`, dispatches, csv path)}; a step = KERNELS_PER_STEP
    dispatches (1: per dispatch)"""
    res, info = {}, None
    passes = sorted(glob.glob(os.path.join(root, "pmc_*")))
    files = []
    for d in passes:
        if os.path.isdir(d):
            files += _newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    for f in files:
        acc, nd = collections.defaultdict(float), set()
        for r in csv.DictReader(open(f)):
            if match not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            nd.add(r["Dispatch_Id"])
            info = {"kernel": r["Kernel_Name"], "vgpr": r.get("VGPR_Count"), "agpr": r.get("Accum_VGPR_Count"),
                    "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"), "wg": r.get("Workgroup_Size"),
                    "grid": r.get("Grid_Size")}
        for name, v in acc.items():
            res[name] = (v / max(1.0, len(nd) / float(KERNELS_PER_STEP)), len(nd), f)
    return res, info


def main():
    root = sys.argv[1]
    match = "solve_kernel"
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
    global KERNELS_PER_STEP
    if "--kernels-per-step" in sys.argv:
        KERNELS_PER_STEP = int(sys.argv[sys.argv.index("--kernels-per-step") + 1])
    ks = kernel_stats(root)
    for name, r in sorted(ks.items(), key=lambda kv: -kv[1]["total_ms"])[:12]:
        print(f"  {name[:90]:90s} calls {r['calls']:>4d} avg {r['avg_ms']:10.3f} ms  total {r['total_ms']:10.3f} ms  {r['pct']}%")
    cs, info = counters(root, match)
    print("kernel:", info)
    for name, (v, n, _) in sorted(cs.items()):
        print(f"    {name:28s} per dispatch {v:.6g}   ({n} dispatches)")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        tb = (2 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024
        print(f"    memory-side bytes per dispatch (2*FETCH+WRITE)*1024 = {tb:.6g}")
    if "--record" not in sys.argv:
        return
    rnd, tag = sys.argv[sys.argv.index("--record") + 1: sys.argv.index("--record") + 3]
    import bench
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    src = []
    for f in {v[2] for v in cs.values()}:
        name = f"{tag}_pmc_" + os.path.basename(os.path.dirname(os.path.dirname(f))).replace("pmc_", "") + ".csv"
        # keep only the rows of the kernel in question (the files hold every dispatch of the process)
        with open(f) as fi, open(os.path.join(dst, name), "w", newline="") as fo:
            rd = csv.DictReader(fi)
            wr = csv.DictWriter(fo, fieldnames=rd.fieldnames)
            wr.writeheader()
            for r in rd:
                if match in r["Kernel_Name"]:
                    wr.writerow(r)
        src.append(f"profiles/{rnd}/{name}")
    kavg = None
    for name, r in ks.items():
        if match in name:
            shutil.copy(r["file"], os.path.join(dst, f"{tag}_kernel_stats.csv"))
            src.append(f"profiles/{rnd}/{tag}_kernel_stats.csv")
            # (several matching kernels = the stages of one step: their average durations add up)
            kavg = r["avg_ms"] if (kavg is None or KERNELS_PER_STEP == 1) else kavg + r["avg_ms"]
    bj = os.path.join(root, "bench.json")  # the bench line of the kernel-trace pass (tools/profile.sh)
    b = json.loads([ln for ln in open(bj).read().splitlines() if ln.startswith("{")][-1])
    counters_kept = ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
                     "SQ_WAIT_INST_ANY", "SQ_WAVES", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
                     "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE")
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        allrec = json.load(open(tj))
    except (OSError, ValueError):
        allrec = {}
    if "only_config" in b:  # one of bench.py's BASELINE configurations (bench.py --only-config cX): keyed by b["pmc_key"]
        key = b["only_config"]
        rec = dict(b["pmc_key"])
        rec.update({"kernel": info["kernel"], "kernel_source_hash": bench.kernel_source_hash(), "vgpr": info["vgpr"],
                    "lds": info["lds"], "workgroup": info["wg"], "kernel_avg_ms_kernel_trace": kavg,
                    "iterations_per_launch": b.get("iterations"), "source": sorted(src)})
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            rec["FETCH_SIZE_KiB_per_launch"], rec["WRITE_SIZE_KiB_per_launch"] = cs["FETCH_SIZE"][0], cs["WRITE_SIZE"][0]
            rec["traffic_bytes_per_launch"] = (2 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024
        if b.get("roofline", {}).get("model_bytes_per_launch"):
            rec["model_bytes_per_launch"] = b["roofline"]["model_bytes_per_launch"]
        for c in counters_kept:
            if c in cs:
                rec[c] = cs[c][0]
        if "GRBM_GUI_ACTIVE" in cs and kavg:  # busy shader-clock cycles summed over the 8 XCDs / the kernel's duration
            rec["shader_clock_hz"] = cs["GRBM_GUI_ACTIVE"][0] / 8.0 / (kavg * 1e-3)
        allrec[key] = rec
        json.dump(allrec, open(tj, "w"), indent=1)
        shutil.copy(bj, os.path.join(dst, f"{tag}_bench.json"))
        print("recorded", key, "->", tj, "hash", rec["kernel_source_hash"])
        return
    cfg = b["config"]
    wl = "lbfgs_rosen256" if "Rosenbrock" in cfg["workload"] else "lbfgs_quad1024"
    rec = {"workload": wl, "batch_per_gpu": cfg["batch_per_gpu"], "n": cfg["n"], "memory": cfg["memory"],
           "precision": 1e-10 if wl == "lbfgs_rosen256" else 1e-6,
           "kernel": info["kernel"], "kernel_source_hash": bench.kernel_source_hash(),
           "vgpr": info["vgpr"], "lds": info["lds"], "workgroup": info["wg"],
           "kernel_avg_ms_kernel_trace": kavg,
           "iterations_per_launch": b["iterations_per_step"], "trials_per_launch": b["roofline"]["trial_phase"]["trials_per_launch"],
           "FETCH_SIZE_KiB_per_launch": cs["FETCH_SIZE"][0], "WRITE_SIZE_KiB_per_launch": cs["WRITE_SIZE"][0],
           "traffic_bytes_per_launch": (2 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024,
           "model_bytes_per_launch": b["roofline"]["model_bytes_per_launch"],
           "source": sorted(src)}
    for c in counters_kept:
        if c in cs:
            rec[c] = cs[c][0]
    allrec["comment"] = ("per-launch PMC figures of the dominant kernel from separate rocprofv3 --pmc passes of the bench.py "
                         "workload (tools/profile.sh, tools/pmc_summary.py --record); traffic_bytes_per_launch = "
                         "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half of a wide read). bench.py uses a "
                         "record only while kernel_source_hash and the workload match.")
    allrec[wl] = rec
    json.dump(allrec, open(tj, "w"), indent=1)
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench.json"))
    print("recorded", wl, "->", tj, "hash", rec["kernel_source_hash"])


if __name__ == "__main__":
    main()
