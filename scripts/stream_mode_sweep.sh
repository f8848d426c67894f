#!/bin/bash
# GPU box: do idle hardware queues of the process (streams that ran one kernel, as a communication library's do) cost the engines anything?
set -e
out=gpurun_out/r2_idle_streams.txt
: > $out
for k in 0 1 2 4 8; do
  for s in 3 2; do
    v=$(BENCH_IDLE_STREAMS=$k timeout -k 10 150 python bench.py --steps 100 --warmup 10 --streams $s --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))")
    echo "idle_streams=$k S=$s  $v" | tee -a $out
  done
done
