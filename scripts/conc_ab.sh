#!/bin/bash
set -e
timeout -k 10 500 python scripts/debug_conc_trajectory.py 2>&1 | tail -6 | tee gpurun_out/r2_conc_debug.txt
out=gpurun_out/r2_jac_conc7.txt
: > $out
for cfg in "1 -1" "0 -1" "1 -1" "0 -1"; do
  set -- $cfg
  for args in "--steps 600 --warmup 20" "--steps 20 --warmup 5"; do
    v=$(MTIP_JAC_CONC=$1 MTIP_JAC_CONC_MIN_K=$2 timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "conc=$1 min_k=$2 $args  $v" | tee -a $out
  done
done
