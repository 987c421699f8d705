"""gate_up (N = 16384, K = 3072, fused SwiGLU) at row counts around the step's: default dispatcher vs D3D_GEMM_TAIL128=0 (set by the caller)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
hd = HipDense()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dt = torch.bfloat16
w = interleave_gate_up((torch.randn(16384, 3072, device="cuda") * 3072 ** -0.5).to(dt))
out = []
for M in (6144, 6400, 6656, 6912, 7168):
    x = (torch.randn(M, 3072, device="cuda") * 0.5).to(dt)
    timeit(lambda: hd.linear_swiglu(x, w))
    out.append(f"M={M}: {timeit(lambda: hd.linear_swiglu(x, w)):.1f} us")
print(os.environ.get("D3D_GEMM_TAIL128", "1"), "  ".join(out))
