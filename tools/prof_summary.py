"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals (short names) and GPU-busy fraction."""
import csv, re, sys, collections
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(k_[a-z0-9_]+)", n)
    if m: return m.group(1) + ("<" + re.search(r"<(.*?)>", n).group(1) + ">" if "<" in n else "")
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
        mt = re.search(r"MT(\d+x\d+x\d+)", n); ty = re.search(r"Bljk_([A-Z]+)_", n)
        return "hipBLASLt " + (ty.group(1) if ty else "") + " MT" + (mt.group(1) if mt else "")
    m = re.search(r"at::native::([a-zA-Z0-9_]+)", n)
    if m:
        inner = re.findall(r"([a-zA-Z0-9_]+(?:Functor|_kernel_cuda|_kernel_impl|kernel)[a-zA-Z0-9_]*)", n)
        return "torch " + (inner[-1] if inner else m.group(1))[:60]
    return n[:60]
tot = collections.defaultdict(lambda: [0, 0.0])
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = short(r["Kernel_Name"]); tot[k][0] += 1; tot[k][1] += d; busy += d
print(f"kernels {len(rows)}  span {(t1 - t0) / 1e6:.1f} ms  sum(kernel) {busy / 1e6:.1f} ms  per-step (/{steps:g}) {busy / 1e6 / steps:.2f} ms")
print(f"{'kernel':70s} {'calls':>7s} {'total ms':>9s} {'ms/step':>8s} {'avg us':>8s} {'%':>6s}")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k:70s} {c:7d} {d / 1e6:9.2f} {d / 1e6 / steps:8.3f} {d / 1e3 / c:8.1f} {100 * d / busy:6.2f}")
