"""ViT projection shapes (CLIP ViT-L/14@224: M = 8 x 257; llava tower @336: M = 8 x 577): 2- vs 4-deep LDS ring."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return a.elapsed_time(b) / n
for dt in (torch.float16, torch.bfloat16):
    for M in (2056, 4616):
        for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
            x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
            b = torch.randn(N, device="cuda").to(dt); r = torch.randn(M, N, device="cuda").to(dt)
            out = []
            for rep in range(2):
                for tile in (130, 132, 258, 0):
                    HipDense.TILE = tile
                    out.append((tile, timeit(lambda: hd.linear(x, w, b, None, r))))
            HipDense.TILE = 0
            t_ref = timeit(lambda: torch.nn.functional.linear(x, w, b) + r)
            fl = 2.0 * M * N * K / 1e9
            print(f"{str(dt)[6:]:8s} M={M} {name} N={N} K={K} tiles128={((M + 127) // 128) * (N // 128)}  " + "  ".join(f"t{t} {ms * 1e3:.1f}us" for t, ms in out) + f"  torch {t_ref * 1e3:.1f}us", flush=True)
