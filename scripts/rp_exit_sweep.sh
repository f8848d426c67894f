#!/bin/bash
# throughput against the exit thresholds of the Jacobi loop of k_rproj: MTIP_RP_EARLY (largest rotation of a sweep after which the Gram
# check is tried) x MTIP_RP_CORR2_MAX (largest |E_ij| the second-order closing step accepts)
out=gpurun_out/rp_exit_sweep.txt
: > $out
for early in 3e-2 6e-2 1e-1 2e-1; do
  for c2 in 1.5e-4 5e-4; do
    line=$(MTIP_RP_EARLY=$early MTIP_RP_CORR2_MAX=$c2 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.0f it/s  %.3f ms/step  sweeps %s  closing %s' % (d['value'], d['ms_per_step'], d.get('jacobi_sweeps_last_step_restart0', [])[-8:], d.get('jacobi_closing_step_last_step_restart0', [])[-8:]))")
    echo "MTIP_RP_EARLY=$early MTIP_RP_CORR2_MAX=$c2  $line" | tee -a $out
  done
done
