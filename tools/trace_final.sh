#!/bin/bash
# rocprofv3 kernel trace of the default benchmark command (final kernels of the round) -> gpurun_out/trace_final/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_final
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --steps 20 --warmup 5 --cpu-baseline off > $out/bench_prof.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 33 > $out/kernel_summary.txt
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
head -14 $out/kernel_summary.txt
tail -1 $out/bench_prof.log | cut -c1-200
