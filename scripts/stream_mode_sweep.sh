#!/bin/bash
set -e
out=gpurun_out/r2_stagger.txt
: > $out
for st in 0 150 250 350 0 250; do
  for args in "--steps 600 --warmup 20" "--steps 20 --warmup 5"; do
    v=$(timeout -k 10 150 python bench.py $args --stagger-us $st --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "stagger=$st $args  $v" | tee -a $out
  done
done
