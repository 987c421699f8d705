"""Decode-row GEMMs (M = 8) of one Phi-3 layer + lm_head: time and effective weight bandwidth of the skinny kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
hd = HipDense()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M = 8
tot = 0
for name, N, K, epi in (("qkv", 9216, 3072, "none"), ("o", 3072, 3072, "res"), ("gate_up", 16384, 3072, "swiglu"), ("down", 3072, 8192, "res"), ("lm_head", 32064, 3072, "none")):
    # rotate over 8 weight copies so that the weights come from HBM, not the 256 MB Infinity Cache
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(8)]
    if epi == "swiglu": ws = [interleave_gate_up(w) for w in ws]
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    i = [0]
    def f():
        w = ws[i[0] % 8]; i[0] += 1
        if epi == "swiglu": return hd.linear_swiglu(x, w)
        return hd.linear(x, w, None, None, r if epi == "res" else None)
    t = timeit(f, 48)
    tot += t if name != "lm_head" else 0
    extra = ""
    print(f"{name:8s} N={N} K={K}: {t * 1e3:.1f} us  {N * K * 2 / t / 1e9:.2f} TB/s{extra}", flush=True)
print(f"layer sum {tot * 1e3:.1f} us -> x32 = {tot * 32:.2f} ms/token")
