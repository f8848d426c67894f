#!/bin/bash
# sweep restarts-per-gpu x streams (GPU box)
for cfg in "8 1" "8 2" "16 1" "16 2" "16 4" "24 3" "32 2" "32 4" "48 3" "64 4"; do
  set -- $cfg
  python bench.py --steps 100 --warmup 10 --restarts-per-gpu $1 --streams $2 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$1 S=$2', round(d['value']), round(d['ms_per_step'],3))"
done
