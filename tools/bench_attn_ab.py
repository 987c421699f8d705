"""A/B of the flash-attention kernels inside one process: v1 (16x16x32, attn_kernels.hip) / v2 (32x32x16, attn2_kernels.hip) / v3 (v2 + LDS-DMA +
bulk fragment prefetch, attn3_kernels.hip), alternating,
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
        for name, v2, v3 in (("v1", False, False), ("v2", True, False), ("v3", True, True)):
            HipDense.ATTN_V2, HipDense.ATTN_V3 = v2, v3
            res.setdefault(name, []).append(timeit(fn))
    HipDense.ATTN_V2 = HipDense.ATTN_V3 = True
    t = {k: min(v) for k, v in res.items()}
    print(f"{label}: " + "   ".join(f"{k} {t[k]:.1f} us ({flops / t[k] / 1e6:.0f} TF/s)" for k in t) + f"   v3/v2 x{t['v2'] / t['v3']:.2f}   rounds " +
          " ".join(f"{k} {np.round(res[k], 1)}" for k in res), flush=True)
H, d = 32, 96
for lens in ([828, 826, 1072, 800, 1012, 753, 766, 769], [755, 835, 946, 744, 938, 766, 761, 729], [900] * 8, [1024] * 8):
    T = sum(lens); Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    ab(lambda: hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T), f"phi3 packed causal {lens[:3]}.. T={T}", sum(4.0 * H * s * s * d / 2 for s in lens))
for dt in (torch.float16, torch.bfloat16):
    qkv = (torch.randn(8, 577, 48, 64, device="cuda") * 0.5).to(dt)
    ab(lambda: hd.attention_qkv(qkv, 16, False), f"vit 8x577 16 heads hd64 {dt}", 4.0 * 8 * 16 * 577 * 577 * 64)
