"""A/B of the split-K partial-sum traffic: non-temporal stores / loads (production) against ordinary cached accesses (D3D_SPLITK_PLAIN=1),
on the Phi-3 o_proj / down_proj shapes at the step's row counts and on the ViT fc2 shape forced onto the split kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dynam3d_amd import _lib
from dynam3d_amd.hip_dense import EPI, HipDense, _p

hd = HipDense()


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def gemm(x, w, out, bias, res, epi, tile):
    M, K = x.shape
    N = w.shape[0]
    dt = 0 if x.dtype == torch.bfloat16 else 1
    fn = hd.lib.d3d_gemm_nt if tile == 0 else None
    if tile == 0:
        _lib.check(hd.lib.d3d_gemm_nt(_p(x), _p(w), _p(out), _p(bias), _p(res), M, N, K, x.stride(0), w.stride(0), N, dt, EPI[epi], hd._stream()))
    else:
        _lib.check(hd.lib.d3d_gemm_nt_tile(_p(x), _p(w), _p(out), _p(bias), _p(res), M, N, K, x.stride(0), w.stride(0), N, dt, EPI[epi], tile, hd._stream()))


cases = [("o_proj", torch.bfloat16, 6656, 3072, 3072, "res", 0), ("o_proj", torch.bfloat16, 6912, 3072, 3072, "res", 0),
         ("down_proj", torch.bfloat16, 6656, 3072, 8192, "res", 0), ("down_proj", torch.bfloat16, 6912, 3072, 8192, "res", 0),
         ("vit fc2 (forced split)", torch.float16, 4616, 1024, 4096, "bias_res", 264), ("vit fc2 (auto)", torch.float16, 4616, 1024, 4096, "bias_res", 0),
         ("o_proj, 256 x 128 tile", torch.bfloat16, 6912, 3072, 3072, "res", 266), ("down_proj, 256 x 128 tile", torch.bfloat16, 6912, 3072, 8192, "res", 266),
         ("o_proj, 256 x 128 tile", torch.bfloat16, 6656, 3072, 3072, "res", 266), ("qkv, 256 x 128 tile", torch.bfloat16, 6656, 9216, 3072, "none", 266),
         ("qkv, auto", torch.bfloat16, 6656, 9216, 3072, "none", 0)]
for name, dt, M, N, K, epi, tile in cases:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    b = torch.randn(N, device="cuda").to(dt) if "bias" in epi else None
    r = torch.randn(M, N, device="cuda").to(dt)
    out = torch.empty((M, N), dtype=dt, device="cuda")
    res = {"0": [], "1": []}
    outs = {}
    for rep in range(3):
        for plain in ("0", "1"):
            os.environ["D3D_SPLITK_PLAIN"] = plain
            res[plain].append(timeit(lambda: gemm(x, w, out, b, r, epi, tile)))
            outs[plain] = out.clone()
    os.environ["D3D_SPLITK_PLAIN"] = "0"
    same = torch.equal(outs["0"], outs["1"])
    print(f"{name:24s} M={M} N={N} K={K}: non-temporal {'/'.join(f'{u:.1f}' for u in res['0'])} us | plain {'/'.join(f'{u:.1f}' for u in res['1'])} us | bit-identical {same}", flush=True)
