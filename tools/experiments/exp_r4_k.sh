#!/bin/bash
# round 4, batch K: configs[3] (closed loop with generation, 8 episodes x 50 steps on one GPU) under the two planners
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for pl in host device host device; do
  echo "planner=$pl" >> gpurun_out/r4_rollout_k.txt
  D3D_FF_PLANNER=$pl timeout 900 python -m dynam3d_amd.rollout --episodes-per-rank 8 --max-steps 50 --grammar-stop-mod 1000000 2>/dev/null | tail -1 >> gpurun_out/r4_rollout_k.txt
done
cat gpurun_out/r4_rollout_k.txt
