import csv, re, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ar = [i for i, r in enumerate(rows) if "erfinv" in r["Kernel_Name"]]
lo, hi = ar[-2], ar[-1]
step = rows[lo + 1:hi]
ours = [r for r in step if re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])]
other = [r for r in step if r not in ours]
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"step: {len(step)} kernels, ours {len(ours)} ({sum(map(d, ours)) / 1e3:.2f} ms), other {len(other)} ({sum(map(d, other)) / 1e3:.3f} ms)")
t0 = int(step[0]["Start_Timestamp"])
prev = None
for i, r in enumerate(step):
    n = r["Kernel_Name"]
    if r in ours:
        prev = re.search(r"\bk_[a-z0-9_]+", n).group(0); continue
    m = re.findall(r"([a-zA-Z0-9_]+(?:Functor|_kernel_cuda|_kernel_impl|kernel)[a-zA-Z0-9_]*)", n)
    nxt = next((re.search(r"\bk_[a-z0-9_]+", q["Kernel_Name"]).group(0) for q in step[i + 1:] if q in ours), None)
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {d(r):6.1f} us  {(m[-1] if m else n)[:70]:70s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}  after {prev}  before {nxt}")
