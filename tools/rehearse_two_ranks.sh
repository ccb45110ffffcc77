#!/bin/bash
# bench.py as the driver launches it for N = 2, rehearsed on ONE GPU: both ranks pinned to device 0, gloo
# instead of RCCL (which refuses two ranks on one device).  Usage (GPU box): bash tools/rehearse_two_ranks.sh [bench args]
R=$GRAFT_REPO_ROOT
cd $R
BENCH_DEVICE=0 BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --batch 1024 --split-iters 300 "$@"
