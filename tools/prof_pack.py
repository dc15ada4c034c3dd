#!/usr/bin/env python3
"""Runs ON the GPU box after rocprofv3 passes: condenses each pass directory (rocpd sqlite) into a small file next to it and
deletes the databases (gpurun copies at most 64 MiB back).

    python tools/prof_pack.py <tag> <out_dir> stats=<dir> [pmc:<name>=<dir> ...]

  stats=<dir>       rocprofv3 --kernel-trace --stats pass   -> <out_dir>/<tag>_kernel_stats.csv
  pmc:<name>=<dir>  a counter pass                          -> <out_dir>/<tag>_pmc_<name>.json  (mean per launch per kernel, summed over instances)
"""
import csv
import glob
import json
import os
import re
import shutil
import sqlite3
import sys


def db_of(d):
    return glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]


def short_name(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k.replace("srfrd::", "").replace("srfrd_long::", "long::")


def main():
    tag, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    for spec in sys.argv[3:]:
        key, d = spec.split("=", 1)
        c = sqlite3.connect(db_of(d))
        if key == "stats":
            rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
                w.writerows(rows)
            for r in rows[:8]:
                print(f"{r[0][:80]:80s} {r[1]:6d} {r[3]:9.2f} us")
        else:
            name = key.split(":", 1)[1]
            per = {}
            for disp, k, cn, v in c.execute("select dispatch_id,kernel_name,counter_name,value from counters_collection"):
                per[(disp, k, cn)] = per.get((disp, k, cn), 0.0) + float(v)
            agg = {}
            for (_, k, cn), v in per.items():
                kk = short_name(k)
                if "rocclr" in kk or "at::" in kk or "elementwise" in kk:
                    continue
                agg.setdefault(kk, {}).setdefault(cn, []).append(v)
            res = {k: {cn: {"launches": len(v), "mean": sum(v) / len(v)} for cn, v in cs.items()} for k, cs in agg.items()}
            json.dump(res, open(os.path.join(out, f"{tag}_pmc_{name}.json"), "w"), indent=1)
            print(name, "->", len(res), "kernels")
        c.close()
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
