"""Every tile variant of d3d_gemm_nt on the four ViT-L/14@336 projection shapes at the step's M = 8 x 577 = 4616 rows, with the
epilogue each one carries in the tower (qkv: bias; out-proj / fc2: bias + residual; fc1: bias + QuickGELU), alternating the variants
inside one process (two rounds), plus hipBLASLt (torch) on the bare product for scale.  Prints microseconds and TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynam3d_amd import _lib
from dynam3d_amd.hip_dense import EPI, HipDense, _p

hd = HipDense()


def timeit(fn, n=40):
    for _ in range(4):
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
    if tile == 0:
        _lib.check(hd.lib.d3d_gemm_nt(_p(x), _p(w), _p(out), _p(bias), _p(res), M, N, K, x.stride(0), w.stride(0), N, dt, EPI[epi], hd._stream()))
    else:
        _lib.check(hd.lib.d3d_gemm_nt_tile(_p(x), _p(w), _p(out), _p(bias), _p(res), M, N, K, x.stride(0), w.stride(0), N, dt, EPI[epi], tile, hd._stream()))


TILES = (0, 130, 132, 164, 260, 264)
Ms = [int(m) for m in os.environ.get("SWEEP_M", "4616,4608").split(",")]
for dtype in (torch.float16,):
    for M in Ms:
        for name, N, K, epi in (("qkv", 3072, 1024, "bias"), ("out", 1024, 1024, "bias_res"), ("fc1", 4096, 1024, "bias_quick_gelu"), ("fc2", 1024, 4096, "bias_res")):
            x = (torch.randn(M, K, device="cuda") * 0.5).to(dtype)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
            b = torch.randn(N, device="cuda").to(dtype)
            r = torch.randn(M, N, device="cuda").to(dtype) if epi == "bias_res" else None
            out = torch.empty((M, N), dtype=dtype, device="cuda")
            res = {}
            for rep in range(2):
                for t in TILES:
                    if t in (260, 264) and N % 256:
                        continue
                    try:
                        us = timeit(lambda: gemm(x, w, out, b, r, epi, t))
                    except RuntimeError as e:
                        us = float("nan")
                    res.setdefault(t, []).append(us)
            ref = timeit(lambda: torch.nn.functional.linear(x, w))
            fl = 2.0 * M * N * K
            best = min((min(v), t) for t, v in res.items() if v == v)
            print(f"{str(dtype)[6:]:8s} M={M} {name:4s} N={N} K={K} | " + "  ".join(f"t{t}: {'/'.join(f'{u:.1f}' for u in v)}" for t, v in res.items())
                  + f" | hipBLASLt(bare) {ref:.1f} us | best t{best[1]} {best[0]:.1f} us = {fl / best[0] / 1e6:.0f} TF/s; auto {fl / min(res[0]) / 1e6:.0f} TF/s", flush=True)
