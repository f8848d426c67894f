#!/bin/bash
# round-3 profile batch: kernel trace + stats of the driver command, CU-time budget, PMC traffic (separate passes), config 5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_driver -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_r3_driver.json 2> gpurun_out/prof_r3_driver.err
python scripts/kernel_stats.py gpurun_out/prof_r3_driver > gpurun_out/r3_kernel_stats_driver.txt
python scripts/cu_time_budget.py gpurun_out/prof_r3_driver 0.6 > gpurun_out/r3_cu_time_budget.txt 2>&1
echo "--- budget"; cat gpurun_out/r3_cu_time_budget.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_default -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_r3_default.json 2> gpurun_out/prof_r3_default.err
python scripts/kernel_stats.py gpurun_out/prof_r3_default > gpurun_out/r3_kernel_stats_default.txt
cp gpurun_out/prof_r3_default/*/*kernel_stats.csv gpurun_out/r3_rocprofv3_kernel_stats_default.csv
cp gpurun_out/prof_r3_driver/*/*kernel_stats.csv gpurun_out/r3_rocprofv3_kernel_stats_driver.csv
echo "--- stats (default schedule)"; head -14 gpurun_out/r3_kernel_stats_default.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r3_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r3_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_r3_fetch gpurun_out/pmc_r3_write gpurun_out/r3_pmc_traffic.json 3 4 > gpurun_out/r3_pmc_hbm_traffic.txt 2>&1
echo "--- pmc"; head -20 gpurun_out/r3_pmc_hbm_traffic.txt
