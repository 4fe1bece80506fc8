"""CPU checks of what bench.py's `roofline` is derived from: the PMC record (profiles/traffic.json) names committed
CSV files, its traffic figure follows from them by the guide's gfx950 rule, and bench.py would only use it for the
kernel sources it was taken from."""
import csv
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def _per_dispatch(path, counter):
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot / max(1, len(disp))


def test_traffic_record_is_derivable_from_the_committed_counter_files():
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["lbfgs_quad1024"]
    for f in rec["source"]:
        assert os.path.exists(os.path.join(ROOT, f)), f
    fetch = [f for f in rec["source"] if f.endswith("pmc_FETCH_SIZE.csv")][0]
    write = [f for f in rec["source"] if f.endswith("pmc_WRITE_SIZE.csv")][0]
    F = _per_dispatch(os.path.join(ROOT, fetch), "FETCH_SIZE")
    W = _per_dispatch(os.path.join(ROOT, write), "WRITE_SIZE")
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KiB) counts half of a wide read on gfx950; WRITE_SIZE is exact
    assert abs((2 * F + W) * 1024 - rec["traffic_bytes_per_launch"]) <= 1e-9 * rec["traffic_bytes_per_launch"]
    # a bound that binds: traffic / kernel time stays below the 8 TB/s peak
    assert rec["traffic_bytes_per_launch"] / (rec["kernel_avg_ms_kernel_trace"] * 1e-3) / 1e9 <= bench.HBM_PEAK_GBS
    assert rec["trials_per_launch"] > rec["iterations_per_launch"] > 0 and rec["SQ_INSTS_VALU"] > 0


def test_traffic_record_is_keyed_to_the_kernel_sources():
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["lbfgs_quad1024"]
    assert len(rec["kernel_source_hash"]) == 16
    if rec["kernel_source_hash"] != bench.kernel_source_hash():
        warnings.warn("profiles/traffic.json was recorded for other kernel sources: bench.py will fall back to the "
                      "L2-request model until tools/profile.sh + tools/pmc_summary.py --record are re-run on the GPU box")
