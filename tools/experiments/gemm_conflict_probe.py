"""Where do the 256-tile GEMM's SQ_LDS_BANK_CONFLICT cycles come from -- the K loop or the transposed epilogue?  The same grid at K = 1024,
3072 and 8192 (plain epilogue): a K-loop source scales with K, an epilogue source does not.  Run under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT
SQ_WAVE_CYCLES --kernel-trace`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
hd.TILE = int(os.environ.get("GEMM_TILE", "0"))      # 0 = the library heuristic; 260 = the 256 x 256 interleaved kernel
torch.manual_seed(0)
M, N = 6656, 4096
for K in (1024, 3072, 8192):
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for _ in range(3):
        y = hd.linear(x, w, None, None)
torch.cuda.synchronize()
print("ok")
