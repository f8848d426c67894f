#!/bin/bash
# GPU box: how the 8 restarts of a GPU are dealt to the engines.  usage: r4_sizes.sh <tag>
set -e
tag=${1:-r4z}
out=gpurun_out/$tag
mkdir -p $out
for sz in 3,3,2 4,2,2 2,4,2 2,2,4 4,4 2,2,2,2 6,2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --engine-sizes $sz > $out/bench_d_$sz.json 2> $out/bench_d_$sz.err
  timeout -k 10 200 python bench.py --no-cpu-baseline --engine-sizes $sz --repeats 1 > $out/bench_600_$sz.json 2> $out/bench_600_$sz.err
done
python - <<PY
import json
for sz in "3,3,2 4,2,2 2,4,2 2,2,4 4,4 2,2,2,2 6,2".split():
  for f in ("bench_d_" + sz, "bench_600_" + sz):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()})
PY
