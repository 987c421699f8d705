#!/bin/bash
# Counter evidence for the FINAL round-2 gate_up kernel (interleaved K loop, transposed epilogue): three separate --pmc passes, then the
# kernel trace of the benchmark command itself.  Writes gpurun_out/pmc_final/ (copy the summaries into profiles/).
set -u
M=${1:-6912}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_final
rm -rf $out && mkdir -p $out
pass() {
  local tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$tag -- python tools/gemm_pmc.py 6 $M > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "*counter_collection.csv" | head -1); cp "$f" $out/${tag}_counters.csv
  k=$(find $out/$tag -name "*kernel_trace.csv" | head -1); cp "$k" $out/${tag}_kernel_trace.csv; rm -rf $out/$tag
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python tools/pmc_summary.py $out k_gemm_nt_256 > $out/summary.json
cat $out/summary.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --steps 20 --warmup 5 --cpu-baseline off > $out/bench_prof.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 33 > $out/kernel_summary.txt
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/trace
head -30 $out/kernel_summary.txt
tail -1 $out/bench_prof.log | cut -c1-200
