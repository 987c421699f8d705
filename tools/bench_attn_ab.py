"""A/B of the two flash-attention kernels inside one process: v1 (16x16x32, attn_kernels.hip) / v2 (32x32x16, attn2_kernels.hip), alternating,
at the step's own shapes: Phi-3 packed causal (hd 96, bf16) and the ViT towers (8 x 577, 16 heads, hd 64, fp16 / bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def ab(fn, label, flops):
    res = {}
    for rnd in range(2):
        for v2 in (False, True):
            HipDense.ATTN_V2 = v2
            res.setdefault(v2, []).append(timeit(fn))
    t1, t2 = min(res[False]), min(res[True])
    print(f"{label}: v1 {t1:.1f} us ({flops / t1 / 1e6:.0f} TF/s)   v2 {t2:.1f} us ({flops / t2 / 1e6:.0f} TF/s)   x{t1 / t2:.2f}   rounds v1 {np.round(res[False], 1)} v2 {np.round(res[True], 1)}", flush=True)
H, d = 32, 96
for lens in ([828, 826, 1072, 800, 1012, 753, 766, 769], [755, 835, 946, 744, 938, 766, 761, 729], [900] * 8, [1024] * 8):
    T = sum(lens); Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    ab(lambda: hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T), f"phi3 packed causal {lens[:3]}.. T={T}", sum(4.0 * H * s * s * d / 2 for s in lens))
for dt in (torch.float16, torch.bfloat16):
    qkv = (torch.randn(8, 577, 48, 64, device="cuda") * 0.5).to(dt)
    ab(lambda: hd.attention_qkv(qkv, 16, False), f"vit 8x577 16 heads hd64 {dt}", 4.0 * 8 * 16 * 577 * 577 * 64)
