"""Experiment: the packed Phi-3 prefill of 8 prompts as ONE launch sequence against TWO half-batches (4 prompts each) on two HIP streams, so that
the partial last rounds / split-K tails / small kernels of one half run beside the other half's GEMMs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
from dynam3d_amd.weights import synth_state_dict

D.enable_hip_kernels(["all"])
cfg = Phi3Config()
sd = synth_state_dict(phi3_param_spec(cfg), seed=0, device="cuda")
dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
lens = [828, 826, 1072, 800, 1012, 753, 766, 769]
def pack(ls):
    T = sum(ls); Tp = (T + 255) // 256 * 256
    x = torch.zeros((Tp, cfg.hidden), dtype=torch.bfloat16, device="cuda")
    x[:T] = (torch.randn(T, cfg.hidden, device="cuda") * 0.5).bfloat16()
    return x
x_all = pack(lens)
# balanced halves by token count
order = np.argsort(lens)[::-1]
ha, hb = [], []
for i in order:
    (ha if sum(lens[j] for j in ha) <= sum(lens[j] for j in hb) else hb).append(int(i))
la, lb = [lens[i] for i in ha], [lens[i] for i in hb]
xa, xb = pack(la), pack(lb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    return dec.prefill_logits_packed(x_all, lens)
def two():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    ctxs = []
    with torch.cuda.stream(s1):
        ca = dec.packed_context(la, xa.shape[0])
    with torch.cuda.stream(s2):
        cb = dec.packed_context(lb, xb.shape[0])
    ya, yb = xa, xb
    n = len(dec.layers)
    for li in range(n):
        last = li == n - 1 and dec.PRUNE_LAST_LAYER
        with torch.cuda.stream(s1):
            ya = dec.layer_packed(li, ya, ca, None, prune=last)
        with torch.cuda.stream(s2):
            yb = dec.layer_packed(li, yb, cb, None, prune=last)
    with torch.cuda.stream(s1):
        oa = dec.final_logits(ya if ya.shape[0] == len(la) else ya[ca["last_rows"]])
    with torch.cuda.stream(s2):
        ob = dec.final_logits(yb if yb.shape[0] == len(lb) else yb[cb["last_rows"]])
    main.wait_stream(s1); main.wait_stream(s2)
    return oa, ob
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print(f"one stream, 8 prompts ({x_all.shape[0]} rows): {timeit(one):.2f} ms    two streams, 4 + 4 prompts ({xa.shape[0]} + {xb.shape[0]} rows): {timeit(two):.2f} ms", flush=True)
