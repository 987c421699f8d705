"""Split-K tail (tile 258) vs whole-tile (257) vs 128x128 (128) on shapes with a partial last round."""
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
    return a.elapsed_time(b) / n
for (M, N, K) in [(5632, 3072, 3072), (5888, 3072, 3072), (6144, 3072, 3072), (6400, 3072, 3072), (6656, 3072, 3072), (5632, 3072, 8192), (5888, 3072, 8192), (6144, 3072, 8192), (6400, 3072, 8192), (6656, 3072, 8192)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    out = []
    for rep in range(2):
        for tile in (128, 258, 0):
            HipDense.TILE = tile
            out.append((tile, timeit(lambda: hd.linear(x, w, None, None, r))))
    HipDense.TILE = 0
    fl = 2.0 * M * N * K / 1e9
    print(f"M={M} N={N} K={K} tiles256={((M + 255) // 256) * (N // 256)}  " + "  ".join(f"t{t} {ms:.3f} ({fl / ms:.0f})" for t, ms in out), flush=True)
