#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_train_render.py tests/test_gpu_decode_persistent.py tests/test_gpu_policy.py -m gpu -x -q -s > gpurun_out/r4_tests_d.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_tests_d.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_d.json 2> gpurun_out/r4_bench_d.err
bash tools/profile_round.sh r04 > gpurun_out/r4_profile.log 2>&1
tail -4 gpurun_out/r4_tests_d.txt; tail -c 1500 gpurun_out/r4_bench_d.json
