#!/bin/bash
# GPU box: everything the round's profiles/ entries come from.  usage: gpu_round.sh <tag>
tag=${1:-r01}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1; tail -2 gpurun_out/$tag/pytest_gpu.log
python bench.py > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err; tail -c 600 gpurun_out/$tag/bench_default.err
python bench.py --no-roofline --no-cpu-baseline > gpurun_out/$tag/bench_noprof.json 2>/dev/null
export TMPDIR=/tmp
( cd /tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -- python bench.py --no-cpu-baseline > gpurun_out/$tag/bench_rocprof.json 2> gpurun_out/$tag/bench_rocprof.err )
python scripts/kernel_stats.py gpurun_out/$tag/stats > gpurun_out/$tag/kernel_stats.txt; cat gpurun_out/$tag/kernel_stats.txt
( cd /tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_fetch -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
( cd /tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_write -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
BP=$(python -c "import json; print(json.loads(open('gpurun_out/$tag/bench_default.json').read().strip().splitlines()[-1])['roofline']['restarts_per_launch'])")
python scripts/pmc_summary.py gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write gpurun_out/$tag/pmc_traffic.json $BP 4 > gpurun_out/$tag/pmc_summary.txt 2>&1; cat gpurun_out/$tag/pmc_summary.txt
python - <<PY
import json
for f in ("bench_default", "bench_noprof", "bench_rocprof"):
    d = json.loads(open("gpurun_out/$tag/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 3), d["roofline"] and {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic", "avg_launch_ms")}, d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
