#!/bin/bash
# GPU box: check of the inverse-SHT Legendre loop with double-buffered LDS operands
set -e
out=gpurun_out/r2_inv_pipe.txt
: > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "switch_short or transforms or steps_golden or full_size" > gpurun_out/r2_inv_pipe_tests.log 2>&1 || { tail -30 gpurun_out/r2_inv_pipe_tests.log; exit 1; }
tail -2 gpurun_out/r2_inv_pipe_tests.log
for s in 1 3; do
  v=$(timeout -k 10 120 python bench.py --steps 200 --warmup 10 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "S=$s  $v" | tee -a $out
done
