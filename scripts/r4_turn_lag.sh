set -e
out=gpurun_out/r4t2; mkdir -p $out
for lag in 1 2; do
  MTIP_TURN_LAG=$lag timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/d_lag$lag.json 2> $out/d_lag$lag.err
  MTIP_TURN_LAG=$lag timeout -k 10 300 python bench.py --no-cpu-baseline > $out/s_lag$lag.json 2> $out/s_lag$lag.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/[ds]_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    fam = {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernel_families_ms"].items()}
    print(f.split("/")[-1], round(d["value"]), "ms/step %.4f" % d["ms_per_step"], [round(w, 2) for w in d["repeats"]["windows_ms"]], fam)
PY
