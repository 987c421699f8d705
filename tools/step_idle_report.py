"""GPU idle time inside the marked step (tools/step_glue.py under rocprofv3 --kernel-trace): union of all kernel intervals (any stream) against the
span, per millisecond of the step -- where is the device waiting for the host?"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ar = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"]]
step = rows[ar[-2] + 1:ar[-1]]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step)
merged = []
for a, b in iv:
    if merged and a <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], b)
    else: merged.append([a, b])
busy = sum(b - a for a, b in merged)
print(f"step span {(t1 - t0) / 1e6:.3f} ms, device busy (union over streams) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
bins = {}
prev = 0
for a, b in merged:
    if a > prev:
        ms = int(prev // 1e6)
        bins[ms] = bins.get(ms, 0) + (a - prev)
    prev = b
for ms in sorted(bins):
    if bins[ms] > 20e3: print(f"  idle in [{ms}, {ms + 1}) ms of the step: {bins[ms] / 1e3:.0f} us")
