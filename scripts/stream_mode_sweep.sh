#!/bin/bash
# GPU box: engines per GPU x stream creation mode (MTIP_STREAM_MODE: 0 default blocking stream, 1 non-blocking, 2 non-blocking
# high priority, 3 non-blocking alternating high / low priority), and the VALU / MFMA issue-rate micro-benchmark.
set -e
out=gpurun_out/r2_stream_modes.txt
: > $out
for mode in 0 1 3; do
  for s in 3 4 6; do
    v=$(MTIP_STREAM_MODE=$mode timeout -k 10 120 python bench.py --steps 100 --warmup 10 --streams $s --no-cpu-baseline --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['host_enqueue_ms_per_step'],3))")
    echo "mode=$mode S=$s  $v" | tee -a $out
  done
done
hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_rates.hip -o /tmp/valu_rates 2>/dev/null
timeout -k 10 60 /tmp/valu_rates | tee gpurun_out/r2_valu_rates2.txt
