import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynam3d_amd.f32_ops import F32Ops
f = F32Ops()
for M in (4608, 5200, 400):
    x = torch.randn(M, 7, device="cuda"); w = torch.randn(768, 7, device="cuda"); b = torch.randn(768, device="cuda")
    y = f.linear(x, w, b)
    ref = x @ w.t() + b
    print(M, "err", float((y - ref).abs().max()))
    torch.cuda.synchronize(); a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f.linear(x, w, b)
    e.record(); torch.cuda.synchronize(); print("  us per call", a.elapsed_time(e) / 50 * 1e3)
