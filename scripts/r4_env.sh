#!/bin/bash
# GPU box: runtime environment switches that touch kernel launch latency, on the driver's window and the schedule
set -e
out=gpurun_out/r4e; mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/d_$tag.json 2> $out/d_$tag.err; env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > $out/s_$tag.json 2> $out/s_$tag.err; }
run base X=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run nointr HSA_ENABLE_INTERRUPT=0
run q4 GPU_MAX_HW_QUEUES=4
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]), "ms/step %.4f" % d["ms_per_step"], [round(w, 2) for w in d["repeats"]["windows_ms"]])
PY
