#!/bin/bash
# round-4 profile batch (GPU box): kernel trace + stats of the driver command and of the default schedule, CU-time budget, PMC
# traffic (separate passes), issue counters of the chained SHT kernels and k_rproj, in-kernel phase stamps of the chained kernels
set -e
out=gpurun_out/r4p
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/driver -- python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_rocprof.json 2> $out/bench_driver_rocprof.err
python scripts/kernel_stats.py $out/driver > $out/kernel_stats_driver_window.txt
python scripts/cu_time_budget.py $out/driver 0.6 > $out/cu_time_budget.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -- python3 bench.py --no-cpu-baseline --repeats 1 > $out/bench_default_rocprof.json 2> $out/bench_default_rocprof.err
python scripts/kernel_stats.py $out/default > $out/kernel_stats_default_schedule.txt
cp $out/default/*/*kernel_stats.csv $out/rocprofv3_kernel_stats.csv
cp $out/driver/*/*kernel_stats.csv $out/rocprofv3_kernel_stats_driver_window.csv
head -12 $out/kernel_stats_default_schedule.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python scripts/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json 3 4 > $out/pmc_hbm_traffic.txt 2>&1; head -24 $out/pmc_hbm_traffic.txt
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $out/pmc_issue -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python scripts/pmc_issue.py $out/pmc_issue k_ > $out/pmc_issue_all_kernels.txt 2>&1; grep -A8 "k_sht_chain\|k_rproj" $out/pmc_issue_all_kernels.txt | head -60
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python scripts/pmc_mfma.py $out/pmc_mfma > $out/pmc_mfma.txt 2>&1; cat $out/pmc_mfma.txt
python scripts/chain_timing.py 3 3 > $out/chain_phase_timers.txt 2>&1
python scripts/chain_timing.py 3 8 >> $out/chain_phase_timers.txt 2>&1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench_steps20_warmup5.err
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python - <<PY
import json
for f in ("bench_steps20_warmup5", "bench_default"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 4), d["repeats"]["windows_ms"], d["roofline"]["frac"], d["roofline"].get("hbm_family", {}).get("frac"), d["cpu_baseline"] and d["cpu_baseline"]["value"], d["whole_step"])
PY
