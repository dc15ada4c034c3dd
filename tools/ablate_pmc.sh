#!/bin/bash
# GPU box: instruction counts of every ablation library (tools/ablate.py --build) -> gpurun_out/ablate_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in $R/srfrd_amd/lib/libabl_${1:-fwd}_*.so; do
  n=$(basename $lib .so)
  rm -rf /tmp/pp_abl
  SRFRD_LIB_PATH=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d /tmp/pp_abl -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-metric-parity --no-secondary > /dev/null 2>&1
  python3 $R/tools/prof_pack.py $n /tmp/pp_abl_out pmc:c=/tmp/pp_abl > /dev/null 2>&1
  python3 - $n <<'P'
import json,sys
d=json.load(open(f"/tmp/pp_abl_out/{sys.argv[1]}_pmc_c.json"))
for k,v in d.items():
    if "encoder_%s" % sys.argv[1].split("_")[1] in k:
        print(sys.argv[1], " ".join(f"{c.replace('SQ_','')}={x['mean']/1e6:.2f}M" for c,x in sorted(v.items())))
P
done
