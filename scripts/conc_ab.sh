#!/bin/bash
# GPU box: concurrent V_r replay (MTIP_JAC_CONC=1) with different split thresholds against the fused kernel, default 3 engines
set -e
out=gpurun_out/r2_jac_conc6.txt
: > $out
for cfg in "1 65" "1 57" "1 49" "0 0" "1 65" "0 0"; do
  set -- $cfg
  for args in "--steps 600 --warmup 20" "--steps 20 --warmup 5"; do
    v=$(MTIP_JAC_CONC=$1 MTIP_JAC_CONC_MIN_K=$2 timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "conc=$1 min_k=$2 $args  $v" | tee -a $out
  done
done
