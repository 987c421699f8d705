"""How long do the waves of the v2 flash-attention kernel sit at the per-tile barrier?  Needs a diagnostics build of the library:
    hipcc ... -DD3D_FA_STAMP -c dynam3d_amd/csrc/attn2_kernels.hip   (see tools/attn_barrier_stamps.sh)
Prints, per shape: cycles per wave-tile, and the share of them spent between arriving at the barrier and leaving it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd import _lib
from dynam3d_amd.hip_dense import HipDense
hd = HipDense()
HipDense.ATTN_V3 = False                              # the stamps live in the v2 kernel
lib = _lib.load()
lib.d3d_fa_stamp_read.argtypes = [C.c_void_p, C.c_int32]
buf = (C.c_ulonglong * 4)()


def read(reset=True):
    assert lib.d3d_fa_stamp_read(C.cast(buf, C.c_void_p), 1 if reset else 0) == 0
    return [int(v) for v in buf]


for name, H, d, lens, causal, dt in (("phi3 packed causal", 32, 96, [828, 826, 1072, 800, 1012, 753, 766, 769], True, torch.bfloat16),
                                     ("phi3 8 x 1024", 32, 96, [1024] * 8, True, torch.bfloat16),
                                     ("vit 8 x 577", 16, 64, [577] * 8, False, torch.float16),
                                     ("ONE workgroup per CU: phi3 1 x 1024, 32 heads", 32, 96, [1024], True, torch.bfloat16)):
    T = sum(lens); Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    for _ in range(3):
        hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens), n_valid=T)
    torch.cuda.synchronize(); read()
    n = 10
    for _ in range(n):
        hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens), n_valid=T)
    torch.cuda.synchronize()
    tot, bar, tiles, waves = read()
    print(f"{name:50s} waves/launch {waves // n:6d}  wave-tiles/launch {tiles // n:7d}  cycles per wave-tile {tot / tiles:7.0f}  at the barrier {bar / tiles:6.0f} ({100 * bar / tot:4.1f} %)")
