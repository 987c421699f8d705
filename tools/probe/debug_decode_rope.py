import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(6)
    H, d, Tmax = 4, 96, 2
    lens = [1, 63]
    B, T = len(lens), sum(lens)
    prompt = (torch.randn(T + 7, 3 * H, d, device="cuda") * 0.8).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32, device="cuda") / d))
    ang = torch.arange(max(lens) + Tmax + 1, dtype=torch.float32, device="cuda")[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
    kn_a, vn_a = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    kn_b, vn_b = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    raw = (torch.randn(B, 3 * H, d, device="cuda") * 0.8).to(dt)
    pos = lens_d.contiguous()
    rot = raw.clone().view(B, 3 * H * d)
    hd.rope_inplace(rot, cos, sin, 1, 2 * H, d, pos)
    rot = rot.view(B, 3 * H, d)
    hd.decode_attention(rot.view(B, -1), prompt.view(T + 7, -1), cu, kn_a, vn_a, H, 0, max(lens))
    hd.decode_attention(raw.view(B, -1), prompt.view(T + 7, -1), cu, kn_b, vn_b, H, 0, max(lens), rope=(cos, sin, pos))
    a, b = kn_a[:, 0].float(), kn_b[:, 0].float()
    diff = (a != b)
    print(dt, "mismatches", int(diff.sum()), "of", diff.numel(), "max abs", float((a - b).abs().max()))
    idx = diff.nonzero()[:5]
    for i in idx.tolist():
        bb, hh, dd = i
        x = raw[bb, H + hh].float(); p = int(pos[bb])
        half = d // 2
        j = dd % half
        c, s = float(cos[p, j]), float(sin[p, j])
        x1, x2 = float(x[j]), float(x[j + half])
        print("  at", i, "a", float(a[bb, hh, dd]), "b", float(b[bb, hh, dd]), "x1", x1, "x2", x2, "c", c, "s", s)
