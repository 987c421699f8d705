"""Timing ablations of the round-6 attention kernel (csrc/attn4_kernels.hip built with -DD3D_ATTN_ABL=n into tools/experiments/build/):
which part of a key tile the kernel's time follows.  Results of n != 0 are wrong by construction; only the times mean anything."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = {0: "baseline", 1: "no softmax arithmetic", 2: "no MFMAs", 3: "no tile requests after the prologue (barriers stay)", 4: "no requests, no barriers",
         5: "fragment reads only in the prologue", 6: "no key tiles: prologue + epilogue only"}
lens = [734, 781, 762, 830, 715, 720, 798, 811]; H, d = 32, 96
T = sum(lens); Tp = (T + 255) // 256 * 256
qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.zeros(Tp, H, d, device="cuda", dtype=torch.bfloat16)
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
vit = (torch.randn(8, 577, 48, 64, device="cuda") * 0.5).to(torch.float16)
vout = torch.zeros(8, 577, 16, 64, device="cuda", dtype=torch.float16)
p = lambda t: C.c_void_p(t.data_ptr())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(lib, shape):
    if shape == "phi3":
        return lib.d3d_flash_attention_v4(p(qkv), p(out), len(lens), max(lens), H, d, C.c_int64(3 * H * d), C.c_int64(0), 0, H, 2 * H, 1, max(lens), p(cu), None, None, 0, stream)
    return lib.d3d_flash_attention_v4(p(vit), p(vout), 8, 577, 16, 64, C.c_int64(48 * 64), C.c_int64(577 * 48 * 64), 0, 16, 32, 0, 577, None, None, None, 1, stream)


def t_us(lib, shape, n=30):
    for _ in range(3):
        assert run(lib, shape) == 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        run(lib, shape)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


libs = {}
for a in NAMES:
    f = os.path.join(HERE, "build", f"libattn4_abl{a}.so")
    if os.path.isfile(f):
        libs[a] = C.CDLL(f)
        libs[a].d3d_flash_attention_v4.restype = C.c_int32
for shape in ("phi3", "vit"):
    res = {a: [] for a in libs}
    for rnd in range(5):
        for a in libs:
            res[a].append(t_us(libs[a], shape))
    for a in libs:
        print(f"{shape:5s} ABL {a} {NAMES[a]:55s} median {np.median(res[a]):7.1f} us  min {min(res[a]):7.1f}", flush=True)
