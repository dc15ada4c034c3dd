#!/usr/bin/env python3
"""Turns the condensed output of tools/profile_c2.sh (gpurun_out/<tag>_*) into the tracked evidence files profiles/<round>_*.

    python tools/prof_publish.py r03c r03

Writes <round>_kernel_stats.csv (copied), <round>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE in KiB per launch, read side doubled
as MI355X_MICROARCH.md prescribes for gfx950; bench.py reads it for `roofline.traffic`), <round>_pmc_mfma_util.json and
<round>_pmc_wave_state.json (chip-wide sums per launch with the two derived ratios DESIGN.md quotes).
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC = ("encoder_fwd", "encoder_bwd")


def load(tag, part):
    return json.load(open(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_{part}.json")))


def means(d):
    return {k: {c: v["mean"] for c, v in ctrs.items()} for k, ctrs in d.items()}


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"{tag}_kernel_stats.csv"), os.path.join(prof, f"{rnd}_kernel_stats.csv"))

    fetch, write = load(tag, "fetch"), load(tag, "write")
    traffic = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> (separate passes) -- python3 bench.py --steps 20 "
                          "--warmup 5 --no-cpu-baseline --no-metric-parity --no-secondary   (tools/profile_c2.sh)",
               "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE "
                        "counts 128-B requests as 64 B for wide streams; gather-width reads are uncalibrated, so the doubled "
                        "figure is an upper bound)",
               "kernels": {}}
    step = 0.0
    for k in fetch:
        if k not in write or fetch[k]["FETCH_SIZE"]["launches"] < 20:          # one-off set-up kernels are not part of a step
            continue
        rd = fetch[k]["FETCH_SIZE"]["mean"] * 1024 * 2
        wr = write[k]["WRITE_SIZE"]["mean"] * 1024
        traffic["kernels"][k] = {"FETCH_SIZE": fetch[k]["FETCH_SIZE"], "WRITE_SIZE": write[k]["WRITE_SIZE"],
                                 "hbm_bytes_per_launch": {"read_x2_corrected": rd, "write": wr, "total": rd + wr}}
        step += rd + wr
    traffic["hbm_bytes_per_step"] = step
    json.dump(traffic, open(os.path.join(prof, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

    mf = {k: v for k, v in means(load(tag, "mfma")).items() if k.startswith(ENC)}
    for v in mf.values():                                                      # 4 SIMDs per CU
        v["mfma_busy_fraction_of_simd_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * v["SQ_BUSY_CU_CYCLES"])
    json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python3 "
                          "bench.py --steps 20 --warmup 5 ... (tools/profile_c2.sh)", "mean_per_launch": mf},
              open(os.path.join(prof, f"{rnd}_pmc_mfma_util.json"), "w"), indent=1)

    a, b = means(load(tag, "ws1")), means(load(tag, "ws2"))
    ws = {}
    for k in a:
        if not k.startswith(ENC):
            continue
        v = dict(a[k]); v.update(b.get(k, {}))
        v["valu_per_mfma"] = v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"]
        v["wait_any_fraction_of_wave_cycles"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
        ws[k] = v
    json.dump({"command": "two passes of 8 SQ counters over the same command (tools/profile_c2.sh)",
               "note": "chip-wide sums per launch", "mean_per_launch": ws},
              open(os.path.join(prof, f"{rnd}_pmc_wave_state.json"), "w"), indent=1)

    print(f"HBM bytes per step: {step / 1e6:.1f} MB")
    for k, e in traffic["kernels"].items():
        print(f"  {k:48s} {e['hbm_bytes_per_launch']['total'] / 1e6:8.2f} MB")
    for k, v in ws.items():
        print(f"  {k:48s} MFMA {v['SQ_INSTS_MFMA'] / 1e6:.2f} M  VALU/MFMA {v['valu_per_mfma']:.1f}  "
              f"MFMA busy {mf[k]['mfma_busy_fraction_of_simd_cycles']:.3f}")


if __name__ == "__main__":
    main()
