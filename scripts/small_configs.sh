#!/bin/bash
# GPU box: BASELINE configs 1 and 2 (launch-bound? no: see profiles/r02_small_configs.txt), and config 4's 64 restarts on ONE GPU through the worker
set -e
out=gpurun_out/r2_cfg2.txt
: > $out
for args in "--config 2 --streams 1" "--config 2 --streams 3" "--config 1 --streams 1 --restarts-per-gpu 1" "--config 2 --streams 1 --restarts-per-gpu 1"; do
  v=$(timeout -k 10 150 python bench.py $args --steps 200 --warmup 20 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), round(d['host_enqueue_ms_per_step'],4), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$args  $v" | tee -a $out
done
timeout -k 10 400 python scripts/bench_worker.py 64 3 2>&1 | tail -5 | tee gpurun_out/r2_worker_64.txt
