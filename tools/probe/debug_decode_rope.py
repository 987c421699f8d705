"""Debug probe: d3d_decode_attention's fused RoPE against d3d_rope_inplace + the plain form, element by element."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from dynam3d_amd.hip_dense import HipDense

hd = HipDense()
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(6)
    H, d, Tmax = 4, 96, 5
    lens = [1, 63, 300, 129]
    B, T = len(lens), sum(lens)
    prompt = (torch.randn(T + 7, 3 * H, d, device="cuda") * 0.8).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32, device="cuda") / d))
    ang = torch.arange(max(lens) + Tmax + 1, dtype=torch.float32, device="cuda")[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
    kn_a, vn_a = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    kn_b, vn_b = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    for t in range(Tmax):
        raw = (torch.randn(B, 3 * H, d, device="cuda") * 0.8).to(dt)
        pos = (lens_d + t).contiguous()
        rot = raw.clone().view(B, 3 * H * d)
        hd.rope_inplace(rot, cos, sin, 1, 2 * H, d, pos)
        rot = rot.view(B, 3 * H, d)
        hd.decode_attention(rot.view(B, -1), prompt.view(T + 7, -1), cu, kn_a, vn_a, H, t, max(lens))
        hd.decode_attention(raw.view(B, -1), prompt.view(T + 7, -1), cu, kn_b, vn_b, H, t, max(lens), rope=(cos, sin, pos))
        a, b = kn_a[:, t].float(), kn_b[:, t].float()
        diff = (a != b)
        print(dt, "t", t, "v equal", bool(torch.equal(vn_a, vn_b)), "k mismatches", int(diff.sum()), "of", diff.numel(), "max abs", float((a - b).abs().max()))
        half = d // 2
        for i in diff.nonzero()[:4].tolist():
            bb, hh, dd = i
            x = raw[bb, H + hh].float()
            p = int(pos[bb])
            j = dd % half
            print("   at", i, "rope_inplace", float(a[bb, hh, dd]), "fused", float(b[bb, hh, dd]), "x1", float(x[j]), "x2", float(x[j + half]),
                  "c", float(cos[p, j]), "s", float(sin[p, j]))
