cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04b
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --cpu-baseline off --no-decode > $out/bench_prof.json 2> $out/bench_prof.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 20 > $out/kernel_summary.txt
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/gate_up_launches.py "$f" $out/bench_prof.json > $out/gate_up_launches.csv
rm -rf $out/trace
tail -1 $out/gate_up_launches.csv
