#!/bin/bash
# GPU box: the four-engine cliff against GPU_MAX_HW_QUEUES and the turn order
set -e
out=gpurun_out/r4q; mkdir -p $out
for q in 4 6 8 16; do for t in "" "--no-turns"; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --engine-sizes 2,2,2,2 $t > $out/q${q}${t}.json 2> $out/q${q}${t}.err
done; done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/q*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]), "ms/step %.4f" % d["ms_per_step"], d["config"]["streams_side_by_side"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()})
PY
