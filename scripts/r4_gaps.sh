#!/bin/bash
# GPU box: step period of ONE engine against the sum of its kernels (launch gaps), for 1 / 2 / 3 / 4 restarts; then the default split
set -e
out=gpurun_out/r4g
mkdir -p $out
for spec in "1 1" "2 2" "3 3" "4 4" "6 3,3" "9 3,3,3" "8 3,3,2"; do
  set -- $spec
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --restarts-per-gpu $1 --engine-sizes $2 > $out/b_$2.json 2> $out/b_$2.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/b_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fam = {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()}
    print(f.split("/")[-1], round(d["value"]), "ms/step %.4f" % d["ms_per_step"], [round(w, 2) for w in d["repeats"]["windows_ms"]], fam)
PY
