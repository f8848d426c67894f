#!/bin/bash
set -e
out=gpurun_out/r2_inv_pm.txt
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2_inv_pm_tests.log 2>&1 || { tail -30 gpurun_out/r2_inv_pm_tests.log; exit 1; }
tail -2 gpurun_out/r2_inv_pm_tests.log
for args in "--steps 200 --warmup 10 --streams 1" "--steps 200 --warmup 10 --streams 3" "--steps 600 --warmup 20" "--steps 20 --warmup 5"; do
  v=$(timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$args  $v" | tee -a $out
done
