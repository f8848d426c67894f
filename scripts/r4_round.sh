#!/bin/bash
# GPU box: round-4 evidence beyond the bench: the full -m gpu suite, the averaging timing, config 5 at ONE restart per GPU
# (BASELINE's config 5: kernel stats, MFMA counters, step time).  usage: r4_round.sh <tag>
set -e
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1 || { tail -40 $out/pytest_gpu.log; exit 1; }
tail -3 $out/pytest_gpu.log
timeout -k 10 300 python scripts/bench_average.py > $out/average_device_resident.txt 2>&1; tail -12 $out/average_device_resident.txt
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 scripts/config5_steps.py 1 8 > $out/cfg5_B1.txt 2>&1; cat $out/cfg5_B1.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c5 -- python3 scripts/config5_steps.py 1 8 > /dev/null 2>&1
python scripts/kernel_stats.py $out/prof_c5 > $out/cfg5_B1_kernel_stats.txt; cat $out/cfg5_B1_kernel_stats.txt
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $out/pmc_c5_mfma -- python3 scripts/config5_steps.py 1 8 > /dev/null 2>&1
python scripts/pmc_mfma.py $out/pmc_c5_mfma > $out/cfg5_B1_pmc_mfma.txt 2>&1; cat $out/cfg5_B1_pmc_mfma.txt
