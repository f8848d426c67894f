#!/bin/bash
set -e
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_avg
timeout -k 10 300 python scripts/bench_average.py 128 32 8 | tee gpurun_out/r2_avg/bench_average.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_avg/stats -- python scripts/bench_average.py 128 32 8 > /dev/null 2> gpurun_out/r2_avg/err.txt
python scripts/kernel_stats.py gpurun_out/r2_avg/stats > gpurun_out/r2_avg/kernel_stats.txt; head -14 gpurun_out/r2_avg/kernel_stats.txt
