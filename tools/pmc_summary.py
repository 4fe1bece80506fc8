"""Summarise rocprofv3 output dirs: per-kernel average duration (kernel trace) and PMC counter
sums per dispatch.  usage: python tools/pmc_summary.py gpurun_out/prof_<tag>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:10.3f} ms  total {float(r['TotalDurationNs'])/1e6:10.3f} ms  {r['Percentage']}%")
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set); info = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
        info[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    print("==", os.path.basename(os.path.dirname(os.path.dirname(f))))
    for k, c in acc.items():
        if "solve" not in k and "two_loop" not in k: continue
        n = len(nd[k])
        print(f"  {k} dispatches {n} vgpr/agpr/sgpr/lds/wg/grid {info[k]}")
        for name, v in c.items():
            print(f"    {name:24s} per dispatch {v/n:.6g}")
