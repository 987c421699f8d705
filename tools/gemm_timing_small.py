import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
for M, N, K in ((256, 256, 3072), (1024, 2048, 3072), (2048, 4096, 3072), (4096, 4096, 3072)):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for rep in range(2):
        HipDense.TILE = 260
        for _ in range(50):
            hd.linear(x, w, None, None)
        HipDense.TILE = 302
        hd.linear(x, w, None, None)
        torch.cuda.synchronize()
