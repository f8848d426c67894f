#!/bin/bash
# GPU box: fused projection pairs (k_proj_xw / k_proj_ua) against the four separate products
set -e
out=gpurun_out/r2_proj_fuse.txt
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r2_proj_fuse_tests.log 2>&1 || { tail -30 gpurun_out/r2_proj_fuse_tests.log; exit 1; }
tail -2 gpurun_out/r2_proj_fuse_tests.log
for f in 1 0; do
  for s in 1 3; do
    v=$(MTIP_PROJ_FUSE=$f timeout -k 10 120 python bench.py --steps 200 --warmup 10 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "fuse=$f S=$s  $v" | tee -a $out
  done
done
