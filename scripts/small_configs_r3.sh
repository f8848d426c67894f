#!/bin/bash
# GPU box: BASELINE configs 1 and 2 with the round-3 kernels (columns as profiles/r02_small_configs.txt)
out=gpurun_out/r3_small_configs.txt
: > $out
for args in "--config 2 --streams 1" "--config 2 --streams 3" "--config 1 --streams 1 --restarts-per-gpu 1" "--config 2 --streams 1 --restarts-per-gpu 1"; do
  v=$(timeout -k 10 150 python3 bench.py $args --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), round(d['host_enqueue_ms_per_step'],4), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$args  $v" | tee -a $out
done
