"""Sum the counters of tools/dsyev_pmc.sh per kernel (all dispatches of the run)."""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:44]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add((k, r["Dispatch_Id"]))
    if "p1/" in f:
        for k, _ in seen:
            calls[k] += 1
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k, "dispatches", calls[k])
    for name, v in sorted(c.items()):
        print(f"    {name:24s} {v:.4g}")
