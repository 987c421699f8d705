cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_bench; rm -rf $out; mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err; echo "rc=$?" >> $out/bench_default.err
tail -c 1500 $out/bench_default.json; tail -3 $out/bench_default.err
