#!/bin/bash
# One split-search worker with an RCCL communicator of one rank (all a one-GPU box can host) at 4096^2 and
# 2048^2: the unsplit time per iteration next to the split one.  Usage (GPU box): bash tools/split_one_rank.sh [iters]
R=$GRAFT_REPO_ROOT
IT=${1:-2000}
for S in 4096 2048; do
  D=/tmp/rdv_$S.$$
  mkdir -p $D && echo go > $D/go
  timeout -k 10 200 python3 $R/image-processing_amd/crim_split_worker.py --rank 0 --world 1 --device 0 --rendezvous $D --size $S --iters $IT || { cat $D/worker*.log; exit 1; }
  python3 - <<PY
import json
r = json.load(open("$D/result.0"))
print({k: r.get(k) for k in ("size", "single_gpu_us_per_iteration", "us_per_iteration", "graph_captured", "rccl_ranks", "split_matches_single_gpu", "error")})
PY
done
