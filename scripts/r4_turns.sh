#!/bin/bash
# GPU box: mtip_run_group_async (engines take turns at the transforms) against independent engines, driver window and schedule
set -e
out=gpurun_out/r4t
mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -k "group_run" > $out/tests.log 2>&1 || (tail -30 $out/tests.log; exit 1)
tail -2 $out/tests.log
for spec in "3,3,2 --no-turns" "3,3,2" "2,2,2,2" "4,4" "3,3,2" "2,3,3" "4,2,2"; do
  set -- $spec
  tag=$(echo "$1$2" | tr ',' '_')
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --engine-sizes $1 $2 > $out/d_$tag.json 2> $out/d_$tag.err
done
for spec in "3,3,2 --no-turns" "3,3,2" "2,2,2,2"; do
  set -- $spec
  tag=$(echo "$1$2" | tr ',' '_')
  timeout -k 10 300 python bench.py --no-cpu-baseline --engine-sizes $1 $2 > $out/s_$tag.json 2> $out/s_$tag.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/[ds]_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fam = {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()}
    print(f.split("/")[-1], round(d["value"]), "ms/step %.4f" % d["ms_per_step"], [round(w, 2) for w in d["repeats"]["windows_ms"]], fam)
PY
