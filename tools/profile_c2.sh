#!/bin/bash
# On the GPU box (gpurun): rocprofv3 kernel stats + PMC passes of the C2 bench line, condensed into gpurun_out/<tag>_*.
#   bash tools/profile_c2.sh <tag> [extra bench.py args]
# Counter passes run on their own (never with --stats / trace domains other than --kernel-trace), as MI355X_MICROARCH.md prescribes.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-metric-parity --no-secondary $*"
S="--steps 20 --warmup 5 $A"
rocprofv3 --kernel-trace --stats -d /tmp/pp_stats -- python3 $R/bench.py --steps 300 $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/pp_mfma -- python3 $R/bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pp_fetch -- python3 $R/bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pp_write -- python3 $R/bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d /tmp/pp_ws1 -- python3 $R/bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d /tmp/pp_ws2 -- python3 $R/bench.py $S > /dev/null 2>&1
python3 $R/tools/prof_pack.py $TAG $O stats=/tmp/pp_stats pmc:mfma=/tmp/pp_mfma pmc:fetch=/tmp/pp_fetch pmc:write=/tmp/pp_write pmc:ws1=/tmp/pp_ws1 pmc:ws2=/tmp/pp_ws2
