import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
shape = sys.argv[1] if len(sys.argv) > 1 else "vit"
B, H, S, d, causal, dt = (8, 16, 577, 64, False, torch.float16) if shape == "vit" else (8, 32, 900, 96, True, torch.bfloat16)
qkv = torch.randn(B, S, 3 * H, d, device="cuda").to(dt)
for _ in range(4):
    o = hd.attention_qkv(qkv, H, causal)
torch.cuda.synchronize()
print("ok")
