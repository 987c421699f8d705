#!/bin/bash
mkdir -p gpurun_out/skinny
python tools/experiments/skinny_deep_ab.py 8 2>&1 | grep -v "^Extension\|amdgpu.ids" | tee gpurun_out/skinny/ab_rows8.txt
python tools/experiments/skinny_deep_ab.py 1 2>&1 | grep -v "^Extension\|amdgpu.ids" | tee gpurun_out/skinny/ab_rows1.txt
for m in 0 1 0 1; do echo "D3D_SKINNY_DEEP=$m"; D3D_SKINNY_DEEP=$m BENCH_DECODE_QUICK=1 python tools/bench_decode.py 2>&1 | grep -v "^Extension\|amdgpu.ids" | tail -4; done | tee gpurun_out/skinny/bench_decode.txt
timeout 900 python -m pytest tests/test_gpu_policy.py tests/test_gpu_dense.py -x -q -k "decode or skinny or generate or rmsnorm" 2>&1 | tail -3
