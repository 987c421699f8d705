cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_glue; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o glue -- python tools/step_glue.py > $out/glue.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/step_glue_report.py "$f" > $out/glue_report.txt 2>&1
rm -rf $out/trace
cat $out/glue_report.txt | head -60
