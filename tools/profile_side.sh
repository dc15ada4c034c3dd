#!/bin/bash
# On the GPU box (gpurun): rocprofv3 kernel stats of the side workloads (C4 / C5 training step), condensed into gpurun_out/.
#   bash tools/profile_side.sh <tag>
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in C4 C5; do
  rocprofv3 --kernel-trace --stats -d /tmp/pp_$W -- python3 $R/bench.py --workload $W --steps 40 --warmup 15 > $O/${TAG}_${W}_line.json 2>/dev/null
  python3 $R/tools/prof_pack.py ${TAG}_${W}_train $O stats=/tmp/pp_$W
done
