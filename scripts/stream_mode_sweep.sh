#!/bin/bash
set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_nonblocking_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r2_nonblocking_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r2_nonblocking_gpu_tests.log
timeout -k 10 400 python scripts/bench_worker.py 8 3 2>&1 | tail -5 | tee gpurun_out/r2_worker_end_to_end.txt
for args in "--steps 600 --warmup 20" "--steps 20 --warmup 5" "--steps 600 --warmup 20 --streams 1"; do
  v=$(timeout -k 10 150 python bench.py $args --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$args  $v" | tee -a gpurun_out/r2_nonblocking_bench.txt
done
