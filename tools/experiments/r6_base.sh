cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_base; rm -rf $out; mkdir -p $out
python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-decode > $out/bench.json 2> $out/bench.err
python tools/stage_times.py > $out/stage_times.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --cpu-baseline off --no-decode > $out/bench_prof.json 2> $out/bench_prof.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 20 > $out/kernel_summary.txt
rm -rf $out/trace
tail -c 600 $out/bench.json; cat $out/stage_times.txt | tail -20; head -30 $out/kernel_summary.txt
