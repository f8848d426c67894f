#!/bin/bash
# GPU box: engines-per-GPU sweep of the driver-style and the 600-step bench.  usage: r4_streams.sh <tag>
set -e
tag=${1:-r4s}
out=gpurun_out/$tag
mkdir -p $out
for st in 2 3 4; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $st > $out/bench_d_s$st.json 2> $out/bench_d_s$st.err
  timeout -k 10 200 python bench.py --no-cpu-baseline --streams $st --repeats 1 > $out/bench_600_s$st.json 2> $out/bench_600_s$st.err
done
python - <<PY
import json
for st in (2, 3, 4):
  for f in ("bench_d_s%d" % st, "bench_600_s%d" % st):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()})
PY
