#!/bin/bash
set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r2_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r2_final_gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_final_bench_driver_like.json 2> gpurun_out/r2_final_bench_driver_like.err
python -c "
import json
d=json.loads(open('gpurun_out/r2_final_bench_driver_like.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])
"
