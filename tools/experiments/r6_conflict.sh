cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_conflict; rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/p -- env GEMM_TILE=260 python tools/experiments/gemm_conflict_probe.py > $out/log.txt 2>&1
f=$(find $out/p -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_gemm_nt_256" in r["Kernel_Name"]:
        acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
byd = collections.defaultdict(dict)
for (d, c), v in acc.items():
    byd[int(d)][c] = sum(v)
for d in sorted(byd):
    print(d, byd[d])
PY
cp "$f" $out/counters.csv; rm -rf $out/p
