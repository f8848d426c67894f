#!/bin/bash
# GPU box: concurrent V_r replay: split-threshold sweep (orders with k_l >= min_k are split) on one engine and on three
set -e
out=gpurun_out/r2_jac_conc8.txt
: > $out
for s in 1 3; do
  for mk in 2 49 57 65; do
    v=$(MTIP_JAC_CONC=1 MTIP_JAC_CONC_MIN_K=$mk timeout -k 10 150 python bench.py --steps 200 --warmup 20 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items() if k in ('proj','polar')})")
    echo "S=$s conc=1 min_k=$mk  $v" | tee -a $out
  done
  v=$(MTIP_JAC_CONC=0 timeout -k 10 150 python bench.py --steps 200 --warmup 20 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items() if k in ('proj','polar')})")
  echo "S=$s conc=0  $v" | tee -a $out
done
