#!/bin/bash
# Config 4 (dynam3d_amd.rollout) as 1, 2, 4 (and 8) PROCESSES ON ONE GPU -- the D3D_SHARE_DEVICE0 hook, gloo for the metric gather
# because RCCL refuses duplicate devices -- full-length episodes.  What it measures: the host-side cost of one rank (wall and CPU
# milliseconds per batch step) under N concurrent ranks on one host, and whether N ranks sharing a GPU still add up to the
# single-rank throughput (they must: the GPU is the shared resource; anything lost is host contention or scheduling).
# usage: tools/experiments/multiproc_one_gpu.sh [steps=30] [list of N = "1 2 4"]
STEPS=${1:-30}
NS=${2:-"1 2 4"}
export D3D_SHARE_DEVICE0=1 D3D_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for N in $NS; do
  echo "== $N rank(s) on one GPU, 8 episodes each, $STEPS steps"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      -m dynam3d_amd.rollout --episodes-per-rank 8 --max-steps $STEPS --stop-mod 1000000007 --per-rank 2>&1 | grep -E "^RANK|^\{" 
done
