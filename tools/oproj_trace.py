import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
M, N, K = 6656, 3072, int(sys.argv[1]) if len(sys.argv) > 1 else 3072
x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16()
for _ in range(30): hd.linear(x, w, None, None, r)
torch.cuda.synchronize()
