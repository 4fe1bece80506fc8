#!/usr/bin/env python3
"""VALU issue utilisation of the dominant kernel of a profiled run (tools/profile.sh with all counter passes):
SQ_ACTIVE_INST_VALU counts quad-cycles summed over the waves, GRBM_GUI_ACTIVE the busy shader-clock cycles summed over
the 8 XCDs; a SIMD issues at most one vector instruction per 4 cycles, so

    utilisation = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)

usage: python tools/valu_util.py r02 h:prof_r02h c3:prof_r02v_c3 ...   ->  profiles/<round>/valu_utilisation.txt
(+ the kernel's rows of the two SQ counter passes as profiles/<round>/<tag>_sq_*.csv)"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary as P


def main():
    rnd, specs = sys.argv[1], sys.argv[2:]
    dst = os.path.join(ROOT, "profiles", rnd)
    out = [__doc__.split("usage:")[0].strip(), ""]
    for spec in specs:
        tag, d = spec.split(":")
        root = os.path.join(ROOT, "gpurun_out", d)
        ks = P.kernel_stats(root)
        cs, info = P.counters(root, "solve_kernel")
        name, st = max(((k, v) for k, v in ks.items() if "solve_kernel" in k), key=lambda kv: kv[1]["total_ms"])
        act, gui = cs["SQ_ACTIVE_INST_VALU"][0], cs["GRBM_GUI_ACTIVE"][0]
        util = 4.0 * act / (gui / 8.0 * 1024.0)
        out.append(f"== {tag}: {name}  avg {st['avg_ms']:.3f} ms")
        out.append(f"   SQ_INSTS_VALU {cs['SQ_INSTS_VALU'][0]:.5g}  SQ_ACTIVE_INST_VALU {act:.5g}  SQ_WAVE_CYCLES {cs['SQ_WAVE_CYCLES'][0]:.5g}"
                   f"  SQ_WAIT_ANY {cs['SQ_WAIT_ANY'][0]:.5g}  GRBM_GUI_ACTIVE {gui:.5g}  waves {cs['SQ_WAVES'][0]:.0f}")
        out.append(f"   VALU issue utilisation {util:.3f}   (waves waiting {cs['SQ_WAIT_ANY'][0] / cs['SQ_WAVE_CYCLES'][0]:.3f} of their cycles)")
        for f in {v[2] for k, v in cs.items() if k.startswith("SQ_") or k.startswith("GRBM")}:
            nm = f"{tag}_sq_" + os.path.basename(os.path.dirname(os.path.dirname(f))).replace("pmc_", "")[:24] + ".csv"
            with open(f) as fi, open(os.path.join(dst, nm), "w", newline="") as fo:
                rd = csv.DictReader(fi)
                wr = csv.DictWriter(fo, fieldnames=rd.fieldnames)
                wr.writeheader()
                for r in rd:
                    if "solve_kernel" in r["Kernel_Name"]:
                        wr.writerow(r)
    open(os.path.join(dst, "valu_utilisation.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
