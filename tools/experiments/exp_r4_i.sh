#!/bin/bash
# round 4, batch I: kernel traces of the memory update alone, planned on the device and on the host (launch counts, GPU time per step)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for pl in device host; do
  rm -rf gpurun_out/prof_ff_$pl
  FF_PLANNERS=$pl FF_BATCHES=8 FF_STEPS=16 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ff_$pl -o ff -- python tools/bench_ff_update.py > gpurun_out/r4_ff_${pl}_trace.txt 2>&1
  TR=$(find gpurun_out/prof_ff_$pl -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py "$TR" 16 > gpurun_out/r4_ff_${pl}_kernels.txt 2>&1
  rm -rf gpurun_out/prof_ff_$pl
  grep "planner=" gpurun_out/r4_ff_${pl}_trace.txt
  head -1 gpurun_out/r4_ff_${pl}_kernels.txt
done
