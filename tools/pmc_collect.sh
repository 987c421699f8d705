#!/bin/bash
# Three separate rocprofv3 --pmc passes over tools/gemm_pmc.py (MI355X_MICROARCH.md: SQ 8 slots, FETCH_SIZE and WRITE_SIZE
# do not fit one pass).  Writes CSVs under gpurun_out/pmc_<tag>/ ; summarise with tools/pmc_summary.py.
set -u
tag=${1:-gate_up}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/sq -- python tools/gemm_pmc.py 6 6400 > $out/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/fetch -- python tools/gemm_pmc.py 6 6400 > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python tools/gemm_pmc.py 6 6400 > $out/write.log 2>&1
for p in sq fetch write; do f=$(find $out/$p -name "*counter_collection.csv" | head -1); cp "$f" $out/${p}_counters.csv; k=$(find $out/$p -name "*kernel_trace.csv" | head -1); cp "$k" $out/${p}_kernel_trace.csv; rm -rf $out/$p; done
ls -la $out
