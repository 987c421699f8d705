#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py -m gpu -x -q -k "gemm" > gpurun_out/r4_tests_e.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests_e.txt
SWEEP_M=4616 python tools/vit_gemm_sweep.py > gpurun_out/r4_vit_sweep2.txt 2>&1
python tools/experiments/bench_splitk_plain.py > gpurun_out/r4_splitk_plain.txt 2>&1
python bench.py --steps 10 --warmup 3 --cpu-baseline off > gpurun_out/r4_bench_e.json 2> gpurun_out/r4_bench_e.err
tail -3 gpurun_out/r4_tests_e.txt; cat gpurun_out/r4_vit_sweep2.txt gpurun_out/r4_splitk_plain.txt; tail -c 900 gpurun_out/r4_bench_e.json
