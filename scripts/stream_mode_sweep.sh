#!/bin/bash
set -e
out=gpurun_out/r2_engine_contention.txt
: > $out
run() {
  v=$(timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$* :  $v" | tee -a $out
}
run --restarts-per-gpu 2 --streams 1
run --restarts-per-gpu 4 --streams 2
run --restarts-per-gpu 6 --streams 3
run --restarts-per-gpu 8 --streams 4
run --restarts-per-gpu 3 --streams 1
run --restarts-per-gpu 6 --streams 2
run --restarts-per-gpu 9 --streams 3
run --restarts-per-gpu 12 --streams 4
