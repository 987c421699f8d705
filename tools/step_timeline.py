"""Timeline of the marked warm step (tools/step_glue.py under rocprofv3 --kernel-trace): per 0.5 ms bin of the step, the busy time of each kernel
family (any stream) and the device's idle time -- where the step's critical path is NOT a GEMM."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ar = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"]]
step = rows[ar[-2] + 1:ar[-1]]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
BIN = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.5e6
nb = int((t1 - t0) // BIN) + 1


def fam(n):
    if "k_gemm_nt_256" in n or "k_gemm_nt<" in n: return "gemm16"
    if "flash_attn" in n: return "attn"
    if "k_gemm_f32x3" in n or "k_row_exp" in n or "k_gemm_f32" in n: return "gemm32"
    if "k_norm" in n or "k_rope" in n or "fixup" in n or "layer_norm" in n: return "rowops"
    if "ffdev" in n or "k_knn" in n or "k_group" in n or "k_set_att" in n or "k_unproj" in n or "frustum" in n or "agent_frame" in n or "k_scatter" in n or "k_gather" in n: return "builder"
    if n.startswith("void at::") or "at::native" in n or "rocclr" in n: return "torch"
    return "other"


busy = [collections.defaultdict(float) for _ in range(nb)]
iv = []
for r in step:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    iv.append((a, b))
    f = fam(r["Kernel_Name"])
    x = a
    while x < b:
        k = int(x // BIN)
        e = min(b, (k + 1) * BIN)
        busy[k][f] += e - x
        x = e
iv.sort()
merged = []
for a, b in iv:
    if merged and a <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], b)
    else: merged.append([a, b])
idle = [BIN] * nb
for a, b in merged:
    x = a
    while x < b:
        k = int(x // BIN)
        e = min(b, (k + 1) * BIN)
        idle[k] -= e - x
        x = e
idle[-1] -= nb * BIN - (t1 - t0)
fams = ["gemm16", "attn", "rowops", "gemm32", "builder", "torch", "other"]
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels; busy us per {BIN / 1e3:.0f} us bin (overlapping streams add up)")
print("  ms    idle  " + "  ".join(f"{f:>7s}" for f in fams))
for k in range(nb):
    print(f"{k * BIN / 1e6:5.1f}  {idle[k] / 1e3:6.0f}  " + "  ".join(f"{busy[k][f] / 1e3:7.0f}" for f in fams))
tot = collections.defaultdict(float)
for k in range(nb):
    for f in fams: tot[f] += busy[k][f]
print("total  %6.0f  " % (sum(idle) / 1e3) + "  ".join(f"{tot[f] / 1e3:7.0f}" for f in fams))
