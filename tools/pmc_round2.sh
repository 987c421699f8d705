#!/bin/bash
# Round-2 counter evidence for the dominant kernel (phi3.gate_up_proj GEMM + SwiGLU, M = 6912 = the benchmark's packed rows):
#   * three separate rocprofv3 --pmc passes (SQ block | FETCH_SIZE | WRITE_SIZE) of the production kernel (tile order GM = 4)
#   * the SAME kernel with tile orders that change the per-XCD operand duplication (GM = 1: 33 panels per 32 tiles, GM = 2: 18,
#     GM = 4: 12 [production], GM = 8: 12, GM = 16: 18): FETCH_SIZE pass + un-profiled timing, alternating, same process tree
# Writes summaries under gpurun_out/pmc_r02/ (copy into profiles/).
set -u
M=${1:-6912}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_r02
rm -rf $out && mkdir -p $out
pass() {  # tag, counters...
  local tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$tag -- python tools/gemm_pmc.py 6 $M > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "*counter_collection.csv" | head -1); cp "$f" $out/${tag}_counters.csv
  k=$(find $out/$tag -name "*kernel_trace.csv" | head -1); cp "$k" $out/${tag}_kernel_trace.csv; rm -rf $out/$tag
}
export D3D_GEMM_GM=4
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python tools/pmc_summary.py $out k_gemm_nt_256 > $out/summary_gm4.json
for gm in 1 2 8 16; do
  export D3D_GEMM_GM=$gm
  pass fetch_gm$gm FETCH_SIZE GRBM_GUI_ACTIVE
done
# un-profiled timing, two rounds, alternating tile orders
for r in 1 2; do
  for gm in 4 1 2 8 16; do
    echo "GM=$gm round $r: $(D3D_GEMM_GM=$gm python tools/bench_gemm_ab.py 257 7 $M phi3.gate_up 2>/dev/null | tail -1)" >> $out/timing.txt
  done
done
python - <<PY > $out/fetch_by_order.txt
import csv, glob
for gm in (4, 1, 2, 8, 16):
    f = "$out/fetch_counters.csv" if gm == 4 else f"$out/fetch_gm{gm}_counters.csv"
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_gemm_nt_256" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    v = v[1:] if len(v) > 1 else v
    print(f"GM={gm:2d}  FETCH_SIZE {sum(v) / len(v):12.0f} KB  x2 (gfx950 tallies 128-B requests at 64 B) = {sum(v) / len(v) * 2048 / 1e9:.3f} GB per launch")
PY
cat $out/summary_gm4.json $out/fetch_by_order.txt $out/timing.txt
