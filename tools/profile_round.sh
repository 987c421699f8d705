#!/bin/bash
# Per-round profile set: `bash tools/profile_round.sh r04` -> gpurun_out/r04/ (copy the summaries into profiles/ with the round prefix):
#   1. rocprofv3 kernel trace + stats of the DEFAULT benchmark command (minus the CPU leg and the generation block that follow the timed region);
#      per-kernel summary; one row per timed gate_up launch
#   2. three --pmc passes over the dominant GEMM (gate_up, M = 6656: the mean packed rows of the timed steps)
#   3. --pmc passes over the flash-attention kernel at the Phi-3 packed shape and the ViT shape (VALU vs MFMA busy)
#   4. kernel trace of the Pretrain render path (tools/bench_render.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
round=${1:-r04}
out=gpurun_out/$round
rm -rf $out && mkdir -p $out
# ---- 1. benchmark trace ----------------------------------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --cpu-baseline off --no-decode > $out/bench_prof.json 2> $out/bench_prof.err
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 20 > $out/kernel_summary.txt          # 8 memory-warming + 2 warm-up + 10 timed passes
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python tools/gate_up_launches.py "$f" $out/bench_prof.json > $out/gate_up_launches.csv
rm -rf $out/trace
# ---- 2. gate_up PMC -----------------------------------------------------------------------------------------------------------------
M=6656
pass() {
  local tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$tag -- python tools/gemm_pmc.py 6 $M > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "*counter_collection.csv" | head -1); cp "$f" $out/${tag}_counters.csv; rm -rf $out/$tag
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
python tools/pmc_summary.py $out k_gemm_nt_256 > $out/pmc_gate_up_summary.json
# ---- 3. attention PMC ------------------------------------------------------------------------------------------------------------------
for shape in phi3 vit; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
             "FETCH_SIZE"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/a$i -- python tools/attn_pmc.py $shape > $out/a$i.log 2>&1
    f=$(find $out/a$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $out/attn_${shape}_p${i}_counters.csv
    rm -rf $out/a$i
  done
done
python - <<PY > $out/pmc_attn_summary.txt
import csv, collections, glob
for f in sorted(glob.glob("$out/attn_*_counters.csv")):
    acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for r in csv.DictReader(open(f)):
        if "flash" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f.split("/")[-1], {k: round(v / max(n[k], 1)) for k, v in acc.items()}, "mean us under pmc", round(sum(dur) / max(len(dur), 1), 1))
PY
# ---- 4. render path ------------------------------------------------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats --output-format csv -d $out/rtrace -o render -- python tools/bench_render.py > $out/render_prof.log 2>&1
f=$(find $out/rtrace -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$f" 13 > $out/render_kernel_summary.txt     # 3 warm-up + 10 timed render calls (+ the MLP A/B at the end)
rm -rf $out/rtrace
head -12 $out/kernel_summary.txt; cat $out/pmc_gate_up_summary.json | head -30; cat $out/pmc_attn_summary.txt; head -20 $out/render_kernel_summary.txt
