"""Target of the --pmc passes over the flash-attention kernel (tools/profile_round.sh): the step's own packed causal Phi-3 shape / the ViT towers' shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
if (sys.argv[1] if len(sys.argv) > 1 else "phi3") == "phi3":
    lens = [828, 826, 1072, 800, 1012, 753, 766, 769]; H, d = 32, 96
    T = sum(lens); Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    for _ in range(5): o = hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T)
else:
    qkv = (torch.randn(8, 577, 48, 64, device="cuda") * 0.5).to(torch.float16)
    for _ in range(5): o = hd.attention_qkv(qkv, 16, False)
torch.cuda.synchronize(); print("ok")
