#!/bin/bash
# round 4, batch B: tests of what changed (attention ABI cut, fused-norm skinny GEMM, closed-loop rollout), decode A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_decode_persistent.py tests/test_gpu_policy.py tests/test_rollout.py -m gpu -x -q -s > gpurun_out/r4_tests_b.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests_b.txt
BENCH_DECODE_QUICK=1 python tools/bench_decode.py > gpurun_out/r4_decode_ab.txt 2>&1
tail -5 gpurun_out/r4_tests_b.txt; cat gpurun_out/r4_decode_ab.txt
