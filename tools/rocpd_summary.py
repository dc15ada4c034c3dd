#!/usr/bin/env python3
"""rocprofv3 (rocpd / sqlite output) -> profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_mfma_util.json.

    python tools/rocpd_summary.py r01f gpurun_out/prof_r01f gpurun_out/pmc_MFMA_f
"""
import csv
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def db_of(d):
    return glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]


def main():
    tag, stats_dir, mfma_dir = sys.argv[1:4]
    c = sqlite3.connect(db_of(stats_dir))
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        w.writerows(rows)
    for r in rows[:6]:
        print(f"{r[0][:72]:72s} {r[1]:5d} {r[3]:9.2f} us")
    c = sqlite3.connect(db_of(mfma_dir))
    per = {}
    for disp, k, cn, v in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection"):
        per[(disp, k, cn)] = per.get((disp, k, cn), 0.0) + float(v)
    agg = {}
    for (_, k, cn), v in per.items():
        if "encoder" in k:
            agg.setdefault(k, {}).setdefault(cn, []).append(v)
    out = {}
    for k, cs in agg.items():
        e = {cn: sum(v) / len(v) for cn, v in cs.items()}
        e["mfma_busy_fraction_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * e["SQ_BUSY_CU_CYCLES"])
        out[k] = e
        print(k[:60], round(e["mfma_busy_fraction_of_simd_cycles"], 4))
    json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- "
                          "python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline", "mean_per_launch": out},
              open(os.path.join(ROOT, "profiles", f"{tag}_pmc_mfma_util.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
