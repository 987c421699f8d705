"""Does the spilling GEMM build fail on its own?  The step's GEMM shapes against a float32 reference, per 256 x 256 output tile, in one fresh
process (run from the root of the tree under test).  Prints one line per shape: worst relative tile error and the tiles above tolerance."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dynam3d_amd import hip_dense as H
from dynam3d_amd.hip_dense import interleave_gate_up
hd = H.HipDense()
torch.manual_seed(7)
dt = torch.bfloat16
shapes = [(768, 3072, 3072, "res"), (768, 3072, 8192, "res"), (768, 9216, 3072, "none"), (768, 16384, 3072, "swiglu"),
          (5632, 3072, 3072, "res"), (5632, 3072, 8192, "res"), (5632, 9216, 3072, "none"), (5632, 16384, 3072, "swiglu"),
          (4616, 3072, 1024, "bias"), (4616, 1024, 1024, "bias_res"), (4616, 4096, 1024, "gelu"), (4616, 1024, 4096, "bias_res")]
for rep in range(2):
    for M, N, K, kind in shapes:
        x = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        b = torch.randn(N, device="cuda").to(dt)
        r = torch.randn(M, N, device="cuda").to(dt)
        ref = x.float() @ w.float().t()
        if kind == "res":
            out, ref = hd.linear(x, w, None, None, r), ref + r.float()
        elif kind == "bias":
            out, ref = hd.linear(x, w, b, None), ref + b.float()
        elif kind == "bias_res":
            out, ref = hd.linear(x, w, b, None, r), ref + b.float() + r.float()
        elif kind == "gelu":
            out, ref = hd.linear(x, w, b, "gelu"), torch.nn.functional.gelu(ref + b.float())
        elif kind == "swiglu":
            g, u = ref[:, :N // 2], ref[:, N // 2:]
            out, ref = hd.linear_swiglu(x, interleave_gate_up(w)), torch.nn.functional.silu(g) * u
        else:
            out = hd.linear(x, w, None, None)
        torch.cuda.synchronize()
        err = (out.float() - ref).abs()
        No = ref.shape[1]
        tm, tn = (M + 255) // 256, (No + 255) // 256
        scale = float(ref.abs().mean())
        bad = []
        for i in range(tm):
            for j in range(tn):
                e = float(err[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].max())
                if not (e < 0.1 * max(scale, 1e-3) * 8):
                    bad.append((i, j, round(e, 3)))
        print(f"rep {rep} M {M} N {N} K {K} {kind:8s}: mean |ref| {scale:.3f}, max err {float(err.max()):.4f}, bad tiles {len(bad)} of {tm * tn} {bad[:6]}", flush=True)
