"""(Needs tools/experiments/gemm_persistent_walk/gemm_kernels.hip built into the library.)  A/B of the 256 x 256 GEMM with workgroup turnover (default) against the persistent tile walk (D3D_GEMM_PERSIST=1, read per call): the
step's multi-round shapes, interleaved rounds in one process, bit-identity of the results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
hd = HipDense()
torch.manual_seed(0)


def bench(fn, n=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, M, N, K, swiglu in (("gate_up", 6656, 16384, 3072, True), ("gate_up", 6144, 16384, 3072, True), ("qkv", 6656, 9216, 3072, False), ("qkv", 6912, 9216, 3072, False),
                              ("vit fc1 x2", 9232, 4096, 1024, False)):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    if swiglu:
        w = interleave_gate_up(w)
    hd.TILE = 0 if swiglu else 260
    fn = (lambda: hd.linear_swiglu(x, w)) if swiglu else (lambda: hd.linear(x, w, None, None))
    outs, res = {}, {"0": [], "1": []}
    for v in ("0", "1"):
        os.environ["D3D_GEMM_PERSIST"] = v
        outs[v] = fn().clone()
        fn(); fn()
    for rnd in range(7):
        for v in ("0", "1"):
            os.environ["D3D_GEMM_PERSIST"] = v
            res[v].append(bench(fn))
    same = torch.equal(outs["0"], outs["1"])
    fl = 2.0 * M * N * K
    print(f"{name:10s} M {M} N {N} K {K}: turnover median {np.median(res['0']):7.1f} us ({fl / np.median(res['0']) / 1e6:6.0f} TF/s)  persistent {np.median(res['1']):7.1f} us "
          f"({fl / np.median(res['1']) / 1e6:6.0f} TF/s)  bit-identical {same}", flush=True)
