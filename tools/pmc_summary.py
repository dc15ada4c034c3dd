#!/usr/bin/env python3
"""Condenses rocprofv3 counter-collection CSVs (one directory per --pmc pass) into profiles/<name>.json.

Usage (after the GPU passes, which write under gpurun_out/):
    python tools/pmc_summary.py profiles/r01_pmc_traffic.json gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE

Per kernel: mean FETCH_SIZE / WRITE_SIZE (KiB per launch, as rocprofv3 reports them) and the HBM bytes per launch
with the gfx950 read-side correction of MI355X_MICROARCH.md (FETCH_SIZE counts the 128-B requests of wide streams as
64 B: doubled here, an upper bound for gather-width reads).  bench.py reads the file for `roofline.traffic`.
"""
import csv
import glob
import json
import os
import re
import sys


def short_name(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k.replace("srfrd::", "").replace("srfrd_long::", "long::")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    kernels = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):        # rocpd (sqlite) output, the default
            import sqlite3
            per = {}
            for disp, kname, cname, val in sqlite3.connect(f).execute(
                    "select dispatch_id, kernel_name, counter_name, value from counters_collection"):
                per[(disp, kname, cname)] = per.get((disp, kname, cname), 0.0) + float(val)   # summed over instances
            for (disp, kname, cname), val in per.items():
                name = short_name(kname)
                if "rocclr" in name or "at::" in name or "elementwise" in name:
                    continue
                kernels.setdefault(name, {}).setdefault(cname, []).append(val)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = short_name(row["Kernel_Name"])
                if "rocclr" in name or "at::" in name or "elementwise" in name:
                    continue
                c = kernels.setdefault(name, {}).setdefault(row["Counter_Name"], [])
                c.append(float(row["Counter_Value"]))
    res = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> (separate passes) -- python3 bench.py "
                      "--steps 20 --warmup 5 --no-cpu-baseline",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE "
                    "counts 128-B requests as 64 B for wide streams; gather-width reads are uncalibrated, so the doubled "
                    "figure is an upper bound)",
           "kernels": {}}
    for name, ctrs in kernels.items():
        e = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in ctrs.items()}
        rd = e.get("FETCH_SIZE", {}).get("mean", 0.0) * 1024 * 2
        wr = e.get("WRITE_SIZE", {}).get("mean", 0.0) * 1024
        e["hbm_bytes_per_launch"] = {"read_x2_corrected": rd, "write": wr, "total": rd + wr}
        res["kernels"][name] = e
    json.dump(res, open(out, "w"), indent=1)
    for name, e in res["kernels"].items():
        print(f"{name:50s} {e['hbm_bytes_per_launch']['total'] / 1e6:9.2f} MB / launch")


if __name__ == "__main__":
    main()
