"""d3d_gemm_nt_f32 (fp32 MFMA) against torch's float32 F.linear (hipBLASLt sgemm) on the token builder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from dynam3d_amd.f32_ops import F32Ops

f = F32Ops()
torch.manual_seed(0)


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


for M, N, K, act in ((4736, 2304, 768, None), (4736, 768, 768, None), (4736, 3072, 768, "gelu"), (4736, 768, 3072, None), (4608, 3072, 1536, None), (128, 768, 768, None),
                     (9000, 2304, 768, None), (128, 2304, 768, None), (128, 3072, 768, "gelu"), (128, 768, 3072, None), (300, 3072, 1552, None), (600, 768, 768, None)):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(N, device="cuda")
    own = lambda: f.linear(x, w, b, act=act)
    def own32():
        f.SPLIT = False
        try:
            return f.linear(x, w, b, act=act)
        finally:
            f.SPLIT = True
    f.SPLIT = True
    ref = (lambda: F.gelu(F.linear(x, w, b))) if act else (lambda: F.linear(x, w, b))
    t_own, t_ref, t_32 = timeit(own), timeit(ref), timeit(own32)
    fl = 2.0 * M * N * K / 1e9
    err = float((own().double() - ref().double()).norm() / ref().double().norm())
    print(f"M={M:5d} N={N:5d} K={K:5d} {act or '':5s} split {t_own * 1e3:7.1f} us ({fl / t_own:6.1f} TF/s)   f32-mfma {t_32 * 1e3:7.1f} us ({fl / t_32:6.1f} TF/s)   torch {t_ref * 1e3:7.1f} us ({fl / t_ref:6.1f} TF/s)   rel diff {err:.1e}")
