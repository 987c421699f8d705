"""Phase timeline of the persistent decode kernel (D3D_DECODE_DEBUG bit 32): layer 5, workgroups 0 and G-1, microseconds per phase."""
import ctypes as C
import os
import sys

os.environ["D3D_DECODE_PERSISTENT"] = "1"
os.environ["D3D_DECODE_DEBUG"] = str(32 | int(sys.argv[1]) if len(sys.argv) > 1 else 32)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dynam3d_amd import _lib, dense_ops as D
from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
from dynam3d_amd.weights import synth_state_dict

D.enable_hip_kernels(["all"])
cfg = Phi3Config(vocab=32064, hidden=3072, layers=8, heads=32, kv_heads=32, mlp=8192)
sd = synth_state_dict(phi3_param_spec(cfg), seed=3)
lens = [864] * 8
dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
T = sum(lens)
x = (torch.randn(T, cfg.hidden, device="cuda") * 0.5).to(torch.bfloat16)
for rep in range(3):
    dec.generate_packed(x, lens, max_new_tokens=8)
lib = _lib.load()
out = (C.c_uint64 * 32)()
lib.d3d_phi3_decode_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.d3d_phi3_decode_stamps.restype = C.c_int32
assert lib.d3d_phi3_decode_stamps(C.c_void_p(torch.cuda.current_stream().cuda_stream), out) == 0
names = ["norm A", "gemm A (qkv)", "barrier A", "attention", "barrier B", "stage C", "gemm C (o)", "barrier C", "norm D", "gemm D (gate_up)", "barrier D",
         "stage E", "gemm E (down)", "barrier E"]
for w, base in (("workgroup 0", 0), ("workgroup G-1", 16)):
    t = [out[base + i] for i in range(16)]
    print(w, " layer total %.1f us" % ((t[15] - t[0]) / 100.0))
    # stamps: 0 start, 1 norm A, 2 gemm A, 3 barrier A, 4 attention, 5 barrier B, 6 stage C, ...
    for i, n in enumerate(names):
        print("   %-18s %6.2f us" % (n, (t[i + 1] - t[i]) / 100.0))
