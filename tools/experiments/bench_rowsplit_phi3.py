"""Phi-3 o_proj / down_proj / qkv at M = 6400 .. 6912: the library's choice (one round of 256-tiles + a K-split tail + fix-up launch) against a ROW split:
the 256 x 256 kernel on the row tiles that fill whole rounds, the 128 x 128 kernel on the remaining rows (no partial sums, no fix-up)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dt = torch.bfloat16
for M in (6400, 6656, 6912):
    for name, N, K, res in (("o_proj", 3072, 3072, True), ("down", 3072, 8192, True), ("qkv", 9216, 3072, False)):
        x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        r = torch.randn(M, N, device="cuda").to(dt) if res else None
        tn, tm = N // 256, M // 256
        HipDense.TILE = 0
        out = {"auto": timeit(lambda: hd.linear(x, w, None, None, r))}
        for rounds in (1, 2, 3):
            rt = rounds * 256 // tn                       # row tiles that fit `rounds` rounds
            if rt >= tm or rt == 0: continue
            m1 = rt * 256
            def split():
                HipDense.TILE = 257
                hd.linear(x[:m1], w, None, None, None if r is None else r[:m1])
                HipDense.TILE = 130
                hd.linear(x[m1:], w, None, None, None if r is None else r[m1:])
            out[f"{rt} row tiles ({rt * tn} tiles) on 256 + {M - m1} rows on 128"] = timeit(split)
        HipDense.TILE = 0
        fl = 2.0 * M * N * K
        print(f"M={M} {name:6s}: " + "   ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF/s)" for k, v in out.items()), flush=True)
