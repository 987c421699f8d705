"""Micro-benchmark + correctness of d3d_gemm_nt against torch (hipBLASLt) on the step's GEMM shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
import torch.nn.functional as F

hd = HipDense()
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7200
shapes = [("phi3.qkv", M, 9216, 3072, "none"), ("phi3.o", M, 3072, 3072, "res"), ("phi3.gate_up", M, 16384, 3072, "swiglu"),
          ("phi3.gate_up_plain", M, 16384, 3072, "none"), ("phi3.down", M, 3072, 8192, "res"),
          ("vit.qkv", 4616, 3072, 1024, "bias"), ("vit.out", 4616, 1024, 1024, "bias_res"), ("vit.fc1", 4616, 4096, 1024, "bias_quick_gelu"),
          ("vit.fc2", 4616, 1024, 4096, "bias_res")]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for dt in (torch.bfloat16, torch.float16):
    for name, m, n, k, epi in shapes:
        x = (torch.randn(m, k, device="cuda") * 0.5).to(dt)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(dt)
        b = (torch.randn(n, device="cuda") * 0.1).to(dt)
        r = (torch.randn(m, n, device="cuda")).to(dt)
        if epi == "swiglu":
            wi = interleave_gate_up(w)
            ref = lambda: (lambda gu: (gu[:, n // 2:].float() * F.silu(gu[:, :n // 2].float())).to(dt))(F.linear(x, w))
            own = lambda: hd.linear_swiglu(x, wi)
        elif epi == "none":
            ref, own = (lambda: F.linear(x, w)), (lambda: hd.linear(x, w, None, None))
        elif epi == "res":
            ref, own = (lambda: F.linear(x, w) + r), (lambda: hd.linear(x, w, None, None, r))
        elif epi == "bias":
            ref, own = (lambda: F.linear(x, w, b)), (lambda: hd.linear(x, w, b, None))
        elif epi == "bias_res":
            ref, own = (lambda: F.linear(x, w, b) + r), (lambda: hd.linear(x, w, b, None, r))
        elif epi == "bias_quick_gelu":
            ref = lambda: (lambda y: y * torch.sigmoid(1.702 * y))(F.linear(x, w, b))
            own = lambda: hd.linear(x, w, b, "quick_gelu")
        yr = ref().float()
        res = {}
        for tile in (128, 257, 258, 0):
            if tile >= 256 and n % 256:
                continue
            HipDense.TILE = tile
            err = (own().float() - yr).norm() / yr.norm()
            res[tile] = (timeit(own), float(err))
        HipDense.TILE = 0
        t_ref, t_plain = timeit(ref), timeit(lambda: F.linear(x, w))
        fl = 2.0 * m * n * k / 1e9
        own_s = "  ".join(f"own{t} {v[0]:.3f} ms ({fl / v[0]:.0f} TF/s, err {v[1]:.1e})" for t, v in res.items())
        print(f"{str(dt)[6:]:9s} {name:20s} M={m} N={n} K={k} {own_s}  torch+epi {t_ref:.3f} ms ({fl / t_ref:.0f})  torch gemm only {t_plain:.3f} ms ({fl / t_plain:.0f})", flush=True)
    if len(sys.argv) > 2:
        break
