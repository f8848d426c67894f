#!/bin/bash
# GPU box: config 5 (256 shells x L = 48, 128 x 256 grid, B_l metric on), 2 restarts on one engine: the round-2 kernels against their switches
set -e
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_cfg5
out=gpurun_out/r2_cfg5/families.txt
: > $out
run() {
  v=$(env "$@" timeout -k 10 200 python bench.py --config 5 --restarts-per-gpu 2 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['avg_ms']*1e3,1) for k,v in d['kernel_families_ms'].items()})")
  echo "$* :  $v" | tee -a $out
}
run MTIP_DUMMY=0
run MTIP_SHT_FWD_PAIR=0
run MTIP_PROJ_FUSE=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_cfg5/stats -- python bench.py --config 5 --restarts-per-gpu 2 --streams 1 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r2_cfg5/bench.json 2> gpurun_out/r2_cfg5/bench.err
python scripts/kernel_stats.py gpurun_out/r2_cfg5/stats > gpurun_out/r2_cfg5/kernel_stats.txt; head -20 gpurun_out/r2_cfg5/kernel_stats.txt
