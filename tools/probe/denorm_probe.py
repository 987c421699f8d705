import sys; sys.path.insert(0, "/root/repo")
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
torch.manual_seed(0)
M, N, K = 256, 256, 256
for scale_a, scale_w in ((1e-6, 1.0), (1.0, 1e-6), (3e-5, 3e-5), (1e-5, 1000.0)):
    a = (torch.randn(M, K, device="cuda") * scale_a).half()
    w = (torch.randn(N, K, device="cuda") * scale_w).half()
    ref = a.double() @ w.double().t()
    out = hd.linear(a, w, None, None).double()
    # outputs are rounded to fp16; compare relative to ref magnitude
    print(f"a~{scale_a:g} w~{scale_w:g}: |a| subnormal frac {(a.abs() < 6.1e-5).float().mean():.2f}  ref rms {ref.pow(2).mean().sqrt():.3e}  out rms {out.pow(2).mean().sqrt():.3e}  rel err {((out - ref).norm() / ref.norm()):.3e}")
