"""Runs only the dominant GEMM (phi3.gate_up_proj, fused SwiGLU) a few times -- target of the rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
hd = HipDense()
torch.manual_seed(0)
M, N, K = (int(sys.argv[2]) if len(sys.argv) > 2 else 6400), 16384, 3072
x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
w = interleave_gate_up((torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    y = hd.linear_swiglu(x, w)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
