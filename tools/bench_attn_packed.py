"""Packed (cu_seqlens) causal flash attention at the step's own sequence lengths vs the same work as a dense batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
H, d = 32, 96
for lens in ([755, 835, 946, 744, 938, 766, 761, 729], [900] * 8, [768] * 8, [1024] * 8, [640] * 8):
    T = sum(lens); Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    t = timeit(lambda: hd.attention_packed(qkv, H, True, cu, len(lens), max(lens)))
    fl = sum(4.0 * H * s * s * d / 2 for s in lens)
    tiles = sum(sum(min(2 * q + 2, -(-s // 64)) for q in range(-(-s // 128))) for s in lens) * H
    print(f"lens {lens[:3]}.. T={T}: {t * 1e3:.1f} us  {fl / t / 1e9:.0f} TF/s   workgroup-tiles {tiles}  -> {t * 1e3 / (tiles / 512):.2f} us per tile-round", flush=True)
