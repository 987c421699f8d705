"""Stress: thousands of back-to-back split-K-tail and skinny GEMM launches of alternating shapes must reproduce their first result bit
for bit (arrival counters re-armed, no stale partials, no lost slices)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
torch.manual_seed(0)
cases = []
for (M, N, K) in ((6400, 3072, 3072), (5888, 3072, 8192), (6144, 3072, 1024), (8, 3072, 8192), (8, 9216, 3072), (16, 32064, 768)):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    cases.append((x, w, r, hd.linear(x, w, None, None, r).clone()))
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    for x, w, r, ref in cases:
        y = hd.linear(x, w, None, None, r)
        if it % 20 == 0 and not torch.equal(y, ref):
            bad += 1
torch.cuda.synchronize()
for x, w, r, ref in cases:
    bad += 0 if torch.equal(hd.linear(x, w, None, None, r), ref) else 1
print("mismatches:", bad)
