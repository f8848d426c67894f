#!/bin/bash
# GPU box: parity of the chained SHT kernels, then A/B of the bench with and without them.  usage: r4_chain_check.sh <tag>
set -e
tag=${1:-r4a}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "transforms or steps or trajectory or config" > $out/pytest_subset.log 2>&1 || { tail -40 $out/pytest_subset.log; exit 1; }
tail -3 $out/pytest_subset.log
for ch in 1 0; do
  MTIP_SHT_CHAIN=$ch timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_d_chain$ch.json 2> $out/bench_d_chain$ch.err
  MTIP_SHT_CHAIN=$ch timeout -k 10 200 python bench.py --no-cpu-baseline > $out/bench_600_chain$ch.json 2> $out/bench_600_chain$ch.err
done
python - <<PY
import json
for f in ("bench_d_chain1", "bench_d_chain0", "bench_600_chain1", "bench_600_chain0"):
    d = json.loads(open("$out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), round(d["ms_per_step"], 4), {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()})
PY
