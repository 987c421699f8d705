import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
def t(M, N, K, tile, n=20):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.float16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.float16)
    HipDense.TILE = tile
    for _ in range(3): hd.linear(x, w, None, None)
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for r in range(5):
        a.record()
        for _ in range(n): hd.linear(x, w, None, None)
        b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b) / n * 1e3)
    HipDense.TILE = 0
    return best
for K in (1024, 4096):
    for wgs, (M, N) in ((128, (2048, 1024)), (256, (2048, 2048)), (512, (4096, 2048)), (768, (6144, 2048)), (1024, (8192, 2048))):
        us = t(M, N, K, 128)
        print(f"K={K} 128-tile workgroups {wgs:5d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s", flush=True)
