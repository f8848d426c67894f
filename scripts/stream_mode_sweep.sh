#!/bin/bash
set -e
out=gpurun_out/r2_final_stream_sweep.txt
: > $out
for s in 1 2 3 4 6; do
  v=$(timeout -k 10 150 python bench.py --steps 200 --warmup 10 --streams $s --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['config']['streams_side_by_side'])")
  echo "B=8 S=$s  $v" | tee -a $out
done
for b in 12 16 24; do
  v=$(timeout -k 10 150 python bench.py --steps 200 --warmup 10 --streams 3 --restarts-per-gpu $b --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['config']['streams_side_by_side'])")
  echo "B=$b S=3  $v" | tee -a $out
done
