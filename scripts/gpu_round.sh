#!/bin/bash
# GPU box: everything the round's profiles/ entries come from.  usage: gpu_round.sh <tag> [skip-tests]
# set -e: a step that fails or is killed at its limit ends the call (no further GPU step after a failed one).
set -e
tag=${1:-r01}
mkdir -p gpurun_out/$tag
if [ -z "$2" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/$tag/pytest_gpu.log; exit 1; }
  tail -2 gpurun_out/$tag/pytest_gpu.log
fi
timeout -k 10 300 python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err || { tail -c 1500 gpurun_out/$tag/bench_default.err; exit 1; }
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_like.json 2> gpurun_out/$tag/bench_driver_like.err
timeout -k 10 200 python bench.py --no-roofline --no-cpu-baseline > gpurun_out/$tag/bench_noprof.json 2>/dev/null
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -- python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_rocprof.json 2> gpurun_out/$tag/bench_rocprof.err
python scripts/kernel_stats.py gpurun_out/$tag/stats > gpurun_out/$tag/kernel_stats.txt; cat gpurun_out/$tag/kernel_stats.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_fetch -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_write -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
BP=$(python -c "import json; r=json.loads(open('gpurun_out/$tag/bench_default.json').read().strip().splitlines()[-1])['roofline']; print(r['restarts_per_launch'])")
python scripts/pmc_summary.py gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write gpurun_out/$tag/pmc_traffic.json $BP 4 > gpurun_out/$tag/pmc_summary.txt 2>&1; cat gpurun_out/$tag/pmc_summary.txt
python - <<PY
import json
for f in ("bench_default", "bench_driver_like", "bench_noprof", "bench_rocprof"):
    d = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"], 3), r and {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "cus_used", "avg_launch_ms")},
          r and r.get("hbm_family") and {k: r["hbm_family"][k] for k in ("kernel", "achieved", "frac", "traffic")}, d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
