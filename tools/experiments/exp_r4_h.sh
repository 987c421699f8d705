#!/bin/bash
# round 4, batch H: the device planner -- GPU tests, update timing A/B, whole-step bench A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ff_plan.py -x -q > gpurun_out/r4_ffplan_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4_ffplan_tests.txt
tail -5 gpurun_out/r4_ffplan_tests.txt
timeout 600 python tools/bench_ff_update.py > gpurun_out/r4_ff_update_ab.txt 2>&1; echo "rc=$?" >> gpurun_out/r4_ff_update_ab.txt
cat gpurun_out/r4_ff_update_ab.txt
for pl in host device host device; do
  D3D_FF_PLANNER=$pl timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off --no-decode 2>gpurun_out/r4_bench_h_$pl.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$pl', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4_bench_h.txt
done
