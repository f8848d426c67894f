#!/bin/bash
set -e
out=gpurun_out/r2_jac_conc2.txt
: > $out
for f in 2 1 0; do
  for s in 1 3; do
    v=$(MTIP_JAC_CONC=$f timeout -k 10 120 python bench.py --steps 100 --warmup 10 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "conc=$f S=$s  $v" | tee -a $out
  done
done
v=$(MTIP_JAC_REPLAY=2 timeout -k 10 120 python bench.py --steps 100 --warmup 10 --streams 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
echo "serial replay S=1  $v" | tee -a $out
