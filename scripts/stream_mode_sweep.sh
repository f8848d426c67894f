#!/bin/bash
# GPU box: forward SHT with generated Legendre values (k_sht_fwd_gen) against the table kernel (k_sht_fwd_pair)
set -e
out=gpurun_out/r2_fwd_gen.txt
: > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "switch_short or transforms or steps_golden or full_size" > gpurun_out/r2_fwd_gen_tests.log 2>&1 || { tail -30 gpurun_out/r2_fwd_gen_tests.log; exit 1; }
tail -2 gpurun_out/r2_fwd_gen_tests.log
for f in 1 0; do
  for s in 1 3; do
    v=$(MTIP_SHT_FWD_GEN=$f timeout -k 10 120 python bench.py --steps 200 --warmup 10 --streams $s --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
    echo "gen=$f S=$s  $v" | tee -a $out
  done
done
