#!/bin/bash
# GPU box: inverse-SHT A/B (one vs two workgroups per shell) at one engine x 8 restarts and at the default 3 engines
set -e
out=gpurun_out/r2_inv_split.txt
: > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "switch_short" > gpurun_out/r2_inv_split_tests.log 2>&1 || { tail -30 gpurun_out/r2_inv_split_tests.log; exit 1; }
tail -2 gpurun_out/r2_inv_split_tests.log
for split in 2 1; do
  for s in 1 3; do
    v=$(MTIP_SHT_INV_SPLIT=$split timeout -k 10 120 python bench.py --steps 200 --warmup 10 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "split=$split S=$s  $v" | tee -a $out
  done
done
