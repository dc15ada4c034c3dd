#!/bin/bash
# On the GPU box: LDS counters of the C2 bench line (bank conflicts, LDS-array cycles) -> gpurun_out/<tag>_pmc_lds.json
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S="--steps 20 --warmup 5 --no-cpu-baseline --no-metric-parity --no-secondary $*"
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS} -d /tmp/pp_lds -- python3 $R/bench.py $S > /dev/null 2>&1
python3 $R/tools/prof_pack.py $TAG $O pmc:lds=/tmp/pp_lds
