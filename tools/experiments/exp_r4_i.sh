#!/bin/bash
# round 4, batch I: where the device-planned update's 3.3 ms go (host profile + kernel trace)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
D3D_FF_PLANNER=device timeout 300 python tools/experiments/ff_host_profile.py > gpurun_out/r4_ffdev_host_profile.txt 2>&1
export TMPDIR=/tmp
FF_PLANNERS=device FF_BATCHES=8 FF_STEPS=16 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ffdev -o ffdev -- python tools/bench_ff_update.py > gpurun_out/r4_ffdev_trace.txt 2>&1
TR=$(find gpurun_out/prof_ffdev -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$TR" 16 > gpurun_out/r4_ffdev_kernels.txt 2>&1
find gpurun_out/prof_ffdev -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4_ffdev_kernel_stats.csv
find gpurun_out/prof_ffdev -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/r4_ffdev_trace.txt
