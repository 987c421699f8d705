"""Decode-only latency of the Phi-3 stack (8 sequences, KV cache): ms per generated token = (T tokens - 2 tokens) / (T - 2) on the same prefill,
min over repetitions, for the launch-per-op path with 1 / 2 / 4 / 8 key ranges in the decode attention and for the persistent kernel.
usage: bench_decode.py [prompt_len=864] [layers=32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynam3d_amd import dense_ops as D
from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
from dynam3d_amd.weights import synth_state_dict

S = int(sys.argv[1]) if len(sys.argv) > 1 else 864
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
D.enable_hip_kernels(["all"])
cfg = Phi3Config(vocab=32064, hidden=3072, layers=layers, heads=32, kv_heads=32, mlp=8192)
sd = synth_state_dict(phi3_param_spec(cfg), seed=3, device="cuda", dtype_for=lambda n: torch.float32 if "norm" in n else torch.bfloat16)
dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
del sd
lens = [S] * 8
x = (torch.randn(sum(lens), cfg.hidden, device="cuda") * 0.5).to(torch.bfloat16)
T = 34


def timed(n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dec.generate_packed(x, lens, max_new_tokens=n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


weights_gb = layers * (9216 * 3072 + 3072 * 3072 + 16384 * 3072 + 3072 * 8192) * 2 / 1e9 + 32064 * 3072 * 2 / 1e9
kv_gb = layers * 8 * S * 2 * 3072 * 2 / 1e9
modes = [("launch-per-op, 1 key range", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "1"}),
         ("launch-per-op, 2 key ranges", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "2"}),
         ("launch-per-op, 4 key ranges", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "4"}),
         ("launch-per-op, 8 key ranges", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "8"}),
         ("launch-per-op, default", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_FUSE_NORM": "0"}),
         ("launch-per-op, RMSNorm fused into the GEMMs", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_FUSE_NORM": "1"}),
         ("launch-per-op, default (again)", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_FUSE_NORM": "0"}),
         ("launch-per-op, fused norm (again)", {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_FUSE_NORM": "1"}),
         ("persistent kernel", {"D3D_DECODE_PERSISTENT": "1"})]
if os.environ.get("BENCH_DECODE_QUICK") == "1":
    modes = modes[4:8]
if os.environ.get("BENCH_DECODE_R05") == "1":        # round 5: one-pass decode attention (D3D_DECODE_ATTN=2) and non-temporal weight loads (D3D_SKINNY_NT=1)
    base = {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "1"}
    modes = [(f"attention kernel {a}, weight loads {'nt' if n == '1' else 'default'}", dict(base, D3D_DECODE_ATTN=a, D3D_SKINNY_NT=n))
             for _ in range(2) for a, n in (("1", "0"), ("2", "0"), ("1", "1"), ("2", "1"))]
if os.environ.get("BENCH_DECODE_R05") == "2":        # one-pass attention: waves per workgroup x keys in flight
    base = {"D3D_DECODE_PERSISTENT": "0", "D3D_DECODE_SPLIT": "1", "D3D_DECODE_ATTN": "2", "D3D_SKINNY_NT": "0"}
    modes = [(f"one-pass attention, cfg {c}", dict(base, D3D_DECODE_ATTN2_CFG=c)) for _ in range(2) for c in ("804", "404", "1604", "808", "1608")]
for name, env in modes:
    os.environ.pop("D3D_DECODE_SPLIT", None)
    os.environ.update(env)
    timed(3)
    short = min(timed(2) for _ in range(4))
    long_ = min(timed(T) for _ in range(4))
    ms = (long_ - short) / (T - 2)
    print(f"{name:48s} {ms:6.3f} ms per token  ({(weights_gb + kv_gb) / ms:5.2f} TB/s over {weights_gb:.2f} GB weights + {kv_gb:.2f} GB keys/values)", flush=True)
