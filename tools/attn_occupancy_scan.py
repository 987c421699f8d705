"""v2 flash attention, equal-work workgroups: B sequences of 1024 tokens x 32 heads = 128 B workgroups (4 per (sequence, head), all 9 block-tiles).
T(B) against B tells whether a second resident workgroup per CU adds throughput: 256 CUs -> B = 2 is one workgroup per CU, B = 4 two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
H, d = 32, 96
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for S in (1024, 512):
    for B in (1, 2, 3, 4, 6, 8, 12, 16):
        lens = [S] * B
        T = sum(lens)
        qkv = (torch.randn(T, 3 * H, d, device="cuda") * 0.5).bfloat16()
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        us = timeit(lambda: hd.attention_packed(qkv, H, True, cu, B, S, n_valid=T))
        wgs = B * H * ((S // 128 + 1) // 2)
        print(f"S {S} B {B:2d}: workgroups {wgs:5d} ({wgs / 256:5.2f} per CU)  {us:7.1f} us   {us / (wgs / 256):6.1f} us per (workgroup per CU)")
