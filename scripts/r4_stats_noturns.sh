#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default schedule with independent engines (--no-turns), for comparison with the turn order
set -e
out=gpurun_out/r4p
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/default_noturns -- python3 bench.py --no-cpu-baseline --repeats 1 --no-turns > $out/bench_default_noturns_rocprof.json 2> $out/bench_default_noturns_rocprof.err
python scripts/kernel_stats.py $out/default_noturns > $out/kernel_stats_default_schedule_noturns.txt
head -12 $out/kernel_stats_default_schedule_noturns.txt
