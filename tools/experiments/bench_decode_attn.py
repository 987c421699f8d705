"""d3d_decode_attention alone: 8 sequences x 32 heads x 96, prompt K/V read in place from per-layer prefill QKV buffers (32 of them, cycled, so
that no launch finds its keys in a cache), us per launch and TB/s for 1 / 2 / 4 / 8 key ranges.  usage: bench_decode_attn.py [prompt_len=864]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dynam3d_amd.hip_dense import HipDense

S = int(sys.argv[1]) if len(sys.argv) > 1 else 864
B, H, hd, NL = 8, 32, 96, 32
hdn = HipDense()
T = B * S
bufs = [(torch.randn(T, 3 * H * hd, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(NL)]
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
knew = torch.zeros((NL, B, 20, H, hd), dtype=torch.bfloat16, device="cuda")
vnew = torch.zeros_like(knew)
qn = (torch.randn(B, 3 * H * hd, device="cuda") * 0.5).to(torch.bfloat16)
cos = torch.ones((S + 32, hd // 2), device="cuda")
sin = torch.zeros((S + 32, hd // 2), device="cuda")
pos = torch.full((B,), S, dtype=torch.int32, device="cuda")
bytes_per = B * S * 2 * H * hd * 2
for sp in (1, 2, 4, 8):
    os.environ["D3D_DECODE_SPLIT"] = str(sp)
    outs = []
    for rep in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for l in range(NL):
            hdn.decode_attention(qn, bufs[l], cu, knew[l], vnew[l], H, 0, S, rope=(cos, sin, pos))
        b.record()
        torch.cuda.synchronize()
        outs.append(a.elapsed_time(b) / NL * 1e3)
    us = min(outs)
    print(f"key ranges {sp}: {us:6.1f} us per launch  {bytes_per / us / 1e6:5.2f} TB/s", flush=True)
