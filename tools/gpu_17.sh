#!/bin/bash
export TMPDIR=/tmp
# N>1 control flow on one GPU: 2 ranks share cuda:0, gloo for the barrier
GDV_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --rows $((1<<26)) 2>&1 | tail -3 | cut -c1-700
# and the exact single-rank driver form
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 5 --warmup 2 --rows $((1<<26)) --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
