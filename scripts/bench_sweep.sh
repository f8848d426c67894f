#!/bin/bash
# sweep restarts-per-gpu x streams (GPU box); usage: bench_sweep.sh "8 1" "8 2 --extra-flag" ...
[ $# -eq 0 ] && set -- "8 1" "8 2" "8 4" "16 2" "16 4" "32 2" "32 4" "64 4"
for cfg in "$@"; do
  set -- $cfg
  python bench.py --steps 100 --warmup 10 --restarts-per-gpu $1 --streams $2 --no-cpu-baseline --no-roofline $3 2>/dev/null | grep '^{' | TAG="B=$1 S=$2 $3" python -c '
import json, os, sys
d = json.loads(sys.stdin.read())
print(os.environ["TAG"], round(d["value"]), round(d["ms_per_step"], 3), "host_enq", round(d.get("host_enqueue_ms_per_step", 0), 3))'
done
