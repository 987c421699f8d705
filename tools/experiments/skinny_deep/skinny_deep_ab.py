"""A/B of the decode token's weight-streaming GEMMs: k_gemm_skinny (round 2-5) against k_gemm_skinny_deep (round 6; D3D_SKINNY_DEEP=0 / 1,
read per call): bit-identity of the results and us / TB/s per launch on the five projections of a Phi-3 layer at 8 rows.  Weights rotate over
`COPIES` distinct tensors per shape so that no launch finds its weights in the Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from dynam3d_amd import hip_dense as H
from dynam3d_amd.hip_dense import interleave_gate_up

hd = H.HipDense()
torch.manual_seed(3)
dt = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
COPIES = int(os.environ.get("COPIES", "6"))
shapes = [("qkv  (norm)", 9216, 3072, "norm"), ("o_proj (res)", 3072, 3072, "res"), ("gate_up (norm, SwiGLU)", 16384, 3072, "norm_swiglu"),
          ("down (res)", 3072, 8192, "res"), ("lm_head (norm)", 32064, 3072, "norm")]
for name, N, K, kind in shapes:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(COPIES)]
    if kind == "norm_swiglu":
        ws = [interleave_gate_up(w) for w in ws]
    g = (1 + 0.1 * torch.randn(K, device="cuda")).float()
    r = torch.randn(M, N, device="cuda").to(dt)

    def run(w):
        if kind == "norm":
            return hd.linear_rmsnorm(x, g, 1e-5, w)
        if kind == "norm_swiglu":
            return hd.linear_rmsnorm(x, g, 1e-5, w, swiglu=True)
        return hd.linear(x, w, None, None, r)

    res = {}
    for mode in ("0", "1"):
        os.environ["D3D_SKINNY_DEEP"] = mode
        out = run(ws[0]).clone()
        for _ in range(3):
            for w in ws:
                run(w)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                for w in ws:
                    run(w)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / (10 * COPIES))
        res[mode] = (out, best)
    same = torch.equal(res["0"][0], res["1"][0])
    mb = N * K * 2 / 1e6
    print(f"{name:26s} N {N:6d} K {K:5d} rows {M}: old {res['0'][1]:6.2f} us ({mb / res['0'][1]:5.2f} TB/s)   deep {res['1'][1]:6.2f} us ({mb / res['1'][1]:5.2f} TB/s)   bit-identical {same}", flush=True)
