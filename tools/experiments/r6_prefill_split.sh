#!/bin/bash
# two-stream prefill (D3D_PREFILL_SPLIT=2) against the single packed batch: the step on one box, alternating
mkdir -p gpurun_out/split
for i in 1 2; do for m in 1 2; do
  D3D_PREFILL_SPLIT=$m python bench.py --steps 20 --warmup 5 --cpu-baseline off --no-decode --parity-golden on 2>gpurun_out/split/err_${m}_$i.txt | tail -1 > gpurun_out/split/bench_${m}_$i.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/split/bench_${m}_$i.json").read())
pg=d.get("parity_golden") or {}
print("split=$m run $i: ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "golden ok", pg.get("ok"), "vs_lowp", pg.get("product_vs_lowp"), "vs_f32", pg.get("product_vs_f32"), "band", pg.get("band"))
PY
done; done
