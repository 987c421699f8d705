"""A/B of the two flash-attention kernels in one process (D3D_ATTN_V4=1 selects csrc/attn4_kernels.hip; read per call): the step's packed causal Phi-3 shape and the ViT
towers' shape, interleaved rounds, median / min in us, and each kernel's distance from float32 attention on the same 16-bit inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
torch.manual_seed(0)


def bench(fn, n=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ab(name, fn, ref):
    res = {"0": [], "1": []}
    err = {}
    for v in ("0", "1"):
        os.environ["D3D_ATTN_V4"] = v
        o = fn().float()
        err[v] = float((o - ref).norm() / ref.norm())
        fn(); fn()
    for rnd in range(7):
        for v in ("0", "1"):
            os.environ["D3D_ATTN_V4"] = v
            res[v].append(bench(fn))
    print(f"{name}: attn3 median {np.median(res['0']):.1f} us (min {min(res['0']):.1f}), attn4 median {np.median(res['1']):.1f} us (min {min(res['1']):.1f}); "
          f"rel L2 vs float32 attention: attn3 {err['0']:.2e}, attn4 {err['1']:.2e}", flush=True)


# Phi-3 packed causal, with the fused query RoPE as the step runs it
lens = [828, 826, 1072, 800, 1012, 753, 766, 769]; H, d = 32, 96
T = sum(lens); Tp = (T + 255) // 256 * 256
qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
refs = []
o = 0
for n in lens:
    x = qkv[o:o + n].float()
    q, k, v = (x[:, i * H:(i + 1) * H].transpose(0, 1) for i in range(3))
    refs.append(F.scaled_dot_product_attention(q[None], k[None], v[None], is_causal=True)[0].transpose(0, 1))
    o += n
ref = torch.cat(refs + [torch.zeros(Tp - T, H, d, device="cuda")])
ab("phi3 packed causal S=753..1072 H=32 hd=96 bf16", lambda: hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T), ref)
lens2 = [734, 781, 762, 830, 715, 720, 798, 811]
T2 = sum(lens2); Tp2 = (T2 + 255) // 256 * 256
qkv2 = (torch.randn(Tp2, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
cu2 = torch.tensor([0] + list(np.cumsum(lens2)), dtype=torch.int32, device="cuda")
inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32, device="cuda") / d))
ang = torch.arange(2048, dtype=torch.float32, device="cuda")[:, None] * inv[None]
cos, sin = ang.cos().to(torch.bfloat16).float().contiguous(), ang.sin().to(torch.bfloat16).float().contiguous()
os.environ["D3D_ATTN_V4"] = "0"
ref2 = hd.attention_packed(qkv2, H, True, cu2, len(lens2), max(lens2), n_valid=T2, rope_q=(cos, sin)).float()
ab("phi3 bench-point lengths + fused q RoPE (ref = attn3)", lambda: hd.attention_packed(qkv2, H, True, cu2, len(lens2), max(lens2), n_valid=T2, rope_q=(cos, sin)), ref2)
# ViT: dense non-causal 577 tokens, 16 heads of 64, 8 images, fp16 (CLIP) and bf16 (llava)
for dt in (torch.float16, torch.bfloat16):
    qv = (torch.randn(8, 577, 48, 64, device="cuda") * 0.5).to(dt)
    q, k, v = (qv[:, :, i * 16:(i + 1) * 16].float().transpose(1, 2) for i in range(3))
    rv = F.scaled_dot_product_attention(q, k, v).transpose(1, 2)
    ab(f"vit dense 8 x 577 x 16 heads hd=64 {dt}", lambda: hd.attention_qkv(qv, 16, False), rv)
