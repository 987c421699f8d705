#!/bin/bash
# round 4, batch C: new backward kernels + render training + full-configuration parity + decode A/B (LDS overlay) + default bench with the parity block
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train_ops.py tests/test_train_render.py tests/test_train_ff.py tests/test_tcnn_module.py tests/test_gpu_dense.py::test_gemm_skinny_fused_rmsnorm tests/test_rollout.py -m gpu -x -q -s > gpurun_out/r4_tests_c.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests_c.txt
python -m pytest tests/test_gpu_full_parity.py -m gpu -x -q -s > gpurun_out/r4_tests_c2.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests_c2.txt
BENCH_DECODE_QUICK=1 python tools/bench_decode.py > gpurun_out/r4_decode_ab2.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_c.json 2> gpurun_out/r4_bench_c.err
tail -4 gpurun_out/r4_tests_c.txt; tail -12 gpurun_out/r4_tests_c2.txt; cat gpurun_out/r4_decode_ab2.txt
