"""Both vision towers concurrently on two HIP streams (as build_inputs runs them), per forced GEMM tile variant: what matters there is
CU-time per GEMM, not the latency of an isolated launch (a partial last round is filled by the other tower's kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.hip_dense import HipDense
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.towers import preprocess_rgb

D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=4)
rgb = torch.randint(0, 255, (B, 224, 224, 3), dtype=torch.uint8, device=dev)
px = preprocess_rgb(rgb)
side = torch.cuda.Stream()


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        a = net.llava_vision.forward(px)
    _, g = net.rgb_encoder.forward(px)
    main.wait_stream(side)
    return a, g


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


variants = [int(v) for v in sys.argv[1:]] or [0, 128, 257, 258]
for rep in range(2):
    for tile in variants:
        HipDense.TILE = tile
        print(f"TILE {tile:4d}: both towers concurrently {timeit(both):6.2f} ms   clip alone {timeit(lambda: net.rgb_encoder.forward(px)):6.2f}   llava alone {timeit(lambda: net.llava_vision.forward(px)):6.2f}", flush=True)
HipDense.TILE = 0
