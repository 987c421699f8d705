# decode profile: kernel summary of tools/bench_generate.py (prefill + 20 greedy tokens) and the skinny-GEMM microbench
mkdir -p gpurun_out/r2p
python tools/bench_generate.py 2>&1 | tail -2 | tee gpurun_out/r2p/generate.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/tools/bench_generate.py > $GRAFT_REPO_ROOT/gpurun_out/r2p/prof_dec.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_dec -name "*.csv" | head
f=$(find /tmp/prof_dec -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 1 > gpurun_out/r2p/decode_kernel_summary.txt
grep -E "k_gemm_skinny|k_decode_attn|k_norm|kernels " gpurun_out/r2p/decode_kernel_summary.txt
