"""In-kernel cycle stamps of the 256 x 256 GEMM loop (tile codes 301 = production loop 257, 302 = interleaved loop 260, 303 = 260 without
any load inside the K loop): shader clock under load, cycles per K tile, prologue / epilogue cycles.  Each stamped launch follows a burst
of ordinary launches so that the clock is the sustained one.  The library prints one line per stamped launch on stderr."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynam3d_amd.hip_dense import HipDense

hd = HipDense()
torch.manual_seed(0)
for M, N, K in ((6912, 9216, 3072), (6912, 3072, 8192), (4616, 3072, 1024)):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for tile in (301, 302, 303):
        for rep in range(2):
            HipDense.TILE = {301: 257, 302: 260, 303: 260}[tile]
            for _ in range(200):
                hd.linear(x, w, None, None)
            HipDense.TILE = tile
            hd.linear(x, w, None, None)
            torch.cuda.synchronize()
HipDense.TILE = 0
