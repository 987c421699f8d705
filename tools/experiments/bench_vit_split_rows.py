"""ViT fc1 / fc2 / out-proj at M = 4616: the 256 x 256 kernel on the rows that make exactly one round of 256 tiles + the 128-tile kernel on the
rest, against the library's own choice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 4616
for dt in (torch.float16, torch.bfloat16):
    for name, N, K, act in (("qkv", 3072, 1024, None), ("out", 1024, 1024, None), ("fc1", 4096, 1024, "quick_gelu"), ("fc2", 1024, 4096, None)):
        x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        b = torch.randn(N, device="cuda").to(dt); r = torch.randn(M, N, device="cuda").to(dt) if act is None else None
        tn = N // 256
        res = {}
        HipDense.TILE = 0
        res["auto"] = timeit(lambda: hd.linear(x, w, b, act, r))
        for rows_tiles in sorted({min(256 // tn, M // 256), M // 256}):
            m1 = rows_tiles * 256
            def split():
                HipDense.TILE = 257
                y1 = hd.linear(x[:m1], w, b, act, None if r is None else r[:m1])
                HipDense.TILE = 0
                y2 = hd.linear(x[m1:], w, b, act, None if r is None else r[m1:])
                return y1, y2
            res[f"256-kernel on {m1} rows ({rows_tiles * tn} tiles) + auto on {M - m1}"] = timeit(split)
            HipDense.TILE = 257
            res[f"   (its 256 part alone"] = timeit(lambda: hd.linear(x[:m1], w, b, act, None if r is None else r[:m1]))
            HipDense.TILE = 0
            res[f"   its remainder alone)"] = timeit(lambda: hd.linear(x[m1:], w, b, act, None if r is None else r[m1:]))
        HipDense.TILE = 0
        print(f"{str(dt)[6:]:8s} {name} N={N} K={K}: " + "   ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
