#!/bin/bash
# GPU box: driver-style bench twice, then rocprofv3 kernel stats of the default bench.  usage: r4_prof.sh <tag> [extra bench args]
set -e
tag=${1:-r4p}
shift || true
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/bench_d1.json 2> $out/bench_d1.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/bench_d2.json 2> $out/bench_d2.err
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --no-cpu-baseline --repeats 1 "$@" > $out/bench_rocprof.json 2> $out/bench_rocprof.err
python scripts/kernel_stats.py $out/stats > $out/kernel_stats.txt; head -30 $out/kernel_stats.txt
python - <<PY
import json
for f in ("bench_d1", "bench_d2", "bench_rocprof"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 4), [round(w, 2) for w in d["repeats"]["windows_ms"]], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()})
PY
