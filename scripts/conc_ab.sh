#!/bin/bash
# GPU box: concurrent V_r replay A/B (default split: the largest order) + parity of the 128 x L32 trajectory
set -e
timeout -k 10 500 python tests/tools/debug_conc_trajectory.py 2>&1 | tail -5 | tee gpurun_out/r2_conc_debug.txt
out=gpurun_out/r2_jac_conc9.txt
: > $out
for f in 1 0 1 0; do
  for args in "--steps 600 --warmup 20" "--steps 20 --warmup 5"; do
    v=$(MTIP_JAC_CONC=$f timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items() if k in ('proj','polar')})")
    echo "conc=$f $args  $v" | tee -a $out
  done
done
