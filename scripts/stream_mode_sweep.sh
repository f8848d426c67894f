#!/bin/bash
set -e
timeout -k 10 400 python scripts/bench_worker.py 8 3 2>&1 | tail -5 | tee gpurun_out/r2_worker_end_to_end.txt
timeout -k 10 400 python scripts/bench_average.py 128 32 8 2>&1 | tail -3 | tee gpurun_out/r2_avg2.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_final_gpu_tests.log 2>&1 || { tail -30 gpurun_out/r2_final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r2_final_gpu_tests.log
