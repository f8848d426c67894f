#!/bin/bash
# GPU box (one GPU): the multi-rank path of bench.py with two gloo ranks sharing device 0 (rehearsal of the launcher contract; RCCL needs one GPU per rank)
set -e
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --same-device --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_2rank_gloo.json 2> gpurun_out/r2_bench_2rank_gloo.err || { tail -20 gpurun_out/r2_bench_2rank_gloo.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r2_bench_2rank_gloo.json').read().strip().splitlines()[-1])
print(d['n_gpus'], round(d['value']), d['ms_per_step'], d['config']['parallelism'], d['scaling'])
"
