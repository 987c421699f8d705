"""Interleaved A/B of d3d_gemm_nt tile variants on the step's GEMM shapes (one process, variants alternated per round;
median and min over the rounds, error against torch).  usage: bench_gemm_ab.py [tiles=257,259] [rounds=7] [M=6912]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from dynam3d_amd.hip_dense import HipDense, interleave_gate_up

tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "257,259").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
M = int(sys.argv[3]) if len(sys.argv) > 3 else 6912
which = sys.argv[4] if len(sys.argv) > 4 else "all"
hd = HipDense()
torch.manual_seed(0)
shapes = [("phi3.qkv", M, 9216, 3072, "none", torch.bfloat16), ("phi3.o", M, 3072, 3072, "res", torch.bfloat16),
          ("phi3.gate_up", M, 16384, 3072, "swiglu", torch.bfloat16), ("phi3.down", M, 3072, 8192, "res", torch.bfloat16),
          ("vit.qkv", 4616, 3072, 1024, "bias", torch.float16), ("vit.out", 4616, 1024, 1024, "bias_res", torch.float16),
          ("vit.fc1", 4616, 4096, 1024, "bias_quick_gelu", torch.float16), ("vit.fc2", 4616, 1024, 4096, "bias_res", torch.float16)]
if which != "all":
    shapes = [s for s in shapes if s[0].startswith(which)]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, m, n, k, epi, dt in shapes:
    x = (torch.randn(m, k, device="cuda") * 0.5).to(dt)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
    b = (torch.randn(n, device="cuda") * 0.1).to(dt)
    r = (torch.randn(m, n, device="cuda")).to(dt)
    if epi == "swiglu":
        wi = interleave_gate_up(w)
        ref = lambda: (lambda gu: (gu[:, n // 2:].float() * F.silu(gu[:, :n // 2].float())).to(dt))(F.linear(x, w))
        own = lambda: hd.linear_swiglu(x, wi)
    elif epi == "none":
        ref, own = (lambda: F.linear(x, w)), (lambda: hd.linear(x, w, None, None))
    elif epi == "res":
        ref, own = (lambda: F.linear(x, w) + r), (lambda: hd.linear(x, w, None, None, r))
    elif epi == "bias":
        ref, own = (lambda: F.linear(x, w, b)), (lambda: hd.linear(x, w, b, None))
    elif epi == "bias_res":
        ref, own = (lambda: F.linear(x, w, b) + r), (lambda: hd.linear(x, w, b, None, r))
    elif epi == "bias_quick_gelu":
        ref = lambda: (lambda y: y * torch.sigmoid(1.702 * y))(F.linear(x, w, b))
        own = lambda: hd.linear(x, w, b, "quick_gelu")
    yr = ref().float()
    times = {t: [] for t in tiles}
    errs = {}
    for t in tiles:
        if t >= 256 and n % 256:
            continue
        HipDense.TILE = t
        y = own().float()
        errs[t] = float((y - yr).norm() / yr.norm())
        y2 = own().float()
        errs[t] = (errs[t], bool(torch.equal(y, y2)))
    for _ in range(rounds):
        for t in tiles:
            if t in errs:
                HipDense.TILE = t
                times[t].append(timeit(own))
    HipDense.TILE = 0
    fl = 2.0 * m * n * k / 1e9
    out = []
    for t in tiles:
        if t in errs:
            ts = sorted(times[t])
            med = ts[len(ts) // 2]
            out.append(f"tile{t}: med {med * 1e3:7.1f} us ({fl / med:5.0f} TF/s) min {ts[0] * 1e3:7.1f} err {errs[t][0]:.1e} det {errs[t][1]}")
    print(f"{name:14s} M={m} N={n} K={k}  " + " | ".join(out), flush=True)
