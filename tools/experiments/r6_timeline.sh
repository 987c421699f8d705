cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_timeline; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o glue -- python tools/step_glue.py > $out/glue.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" 0.5 > $out/timeline.txt
python tools/step_idle_report.py "$f" > $out/idle.txt
rm -rf $out/trace
head -40 $out/timeline.txt; tail -3 $out/timeline.txt; cat $out/idle.txt | head -20
