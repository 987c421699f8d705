import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
B,H,S,d = 8,32,900,96
qkv = torch.randn(B,S,3*H,d,device="cuda").to(torch.bfloat16)
for _ in range(4): o = hd.attention_qkv(qkv, H, True)
torch.cuda.synchronize(); print("ok")
