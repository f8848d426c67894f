#!/bin/bash
# GPU box: parity tests, profiled bench (kernel stats), plain bench at 1 and 2 streams.  usage: gpu_check.sh <tag>
tag=${1:-x}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$tag.log 2>&1; tail -2 gpurun_out/pytest_gpu_$tag.log
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err )
python scripts/kernel_stats.py gpurun_out/prof_$tag > gpurun_out/stats_$tag.txt; cat gpurun_out/stats_$tag.txt
bash scripts/bench_sweep.sh "8 1" "8 2"
