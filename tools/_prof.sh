set -e
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --steps 300 --no-cpu-baseline --no-metric-parity --no-secondary > $R/gpurun_out/prof_c2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-metric-parity --no-secondary > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-metric-parity --no-secondary > $R/gpurun_out/pmc_w.log 2>&1
echo profiled
