"""Idle time between consecutive kernels of the marked step (tools/step_glue.py under rocprofv3 --kernel-trace): per predecessor kernel,
how long the GPU sat between its end and the next kernel's start (single-stream sections only: the Phi-3 prefill)."""
import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ar = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"]]
step = rows[ar[-2] + 1:ar[-1]]
name = lambda r: (re.search(r"\bk_[a-z0-9_]+(<[^>]*>)?", r["Kernel_Name"]) or re.search(r"[A-Za-z_]+", r["Kernel_Name"])).group(0)
# the prefill = from the first k_norm after k_assemble_prompt to the end
i0 = next(i for i, r in enumerate(step) if "k_assemble_prompt" in r["Kernel_Name"])
pre = step[i0 + 1:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in pre)
span = int(pre[-1]["End_Timestamp"]) - int(pre[0]["Start_Timestamp"])
gaps = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(pre, pre[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    k = name(a) + " -> " + name(b)
    gaps[k][0] += 1; gaps[k][1] += g
print(f"prefill window: {len(pre)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:110s} x{c:3d}  mean gap {g / c / 1e3:6.2f} us  total {g / 1e6:.3f} ms")
