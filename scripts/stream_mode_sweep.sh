#!/bin/bash
set -e
out=gpurun_out/r2_jac_conc3.txt
: > $out
for f in 1 0; do
  for args in "--steps 20 --warmup 5 --streams 3" "--steps 20 --warmup 5 --streams 1" "--steps 600 --warmup 20 --streams 3" "--steps 600 --warmup 20 --streams 2"; do
    v=$(MTIP_JAC_CONC=$f timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "conc=$f $args  $v" | tee -a $out
  done
done
