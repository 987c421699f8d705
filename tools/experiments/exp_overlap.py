"""Upper-bound experiment for a chunked prefill: how much of build_inputs (CLIP tower, 3D-token update with its host round trips, llava tower,
prefix) hides under a Phi-3 prefill running on another stream?  Steady state of: [prefill of frame i-1's prompt on stream A] || [build_inputs of
frame i on the main stream], against the serial step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes

D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=64)
net.feature_fields.initialize_camera_setting(90., 90.)
ep = SyntheticEpisodes(B, seed=0)
instr = [INSTRUCTION_64] * B
frames = []
for _ in range(40):
    fr = ep.next()
    frames.append((dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev)), [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm))
def build(i):
    o, p, h, s = frames[i]
    return net.build_inputs(o, instr, p, h, patch_segm=s, return_rows="packed")
for i in range(10):
    x, L = build(i); net.llm.prefill_logits_packed(x, L)
torch.cuda.synchronize()
# serial
t0 = time.perf_counter()
for i in range(10, 20):
    x, L = build(i); lo = net.llm.prefill_logits_packed(x, L)
torch.cuda.synchronize(); ser = (time.perf_counter() - t0) / 10 * 1e3
# pipelined across frames (NOT a legal schedule for the real loop: the next observation depends on this step's action)
A = torch.cuda.Stream()
main = torch.cuda.current_stream()
x, L = build(20); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(21, 31):
    A.wait_stream(main)
    with torch.cuda.stream(A):
        lo = net.llm.prefill_logits_packed(x, L)
    x.record_stream(A)
    x, L = build(i)
    main.wait_stream(A)
torch.cuda.synchronize(); pip = (time.perf_counter() - t0) / 10 * 1e3
print(f"serial step {ser:.2f} ms; prefill(i-1) || build_inputs(i): {pip:.2f} ms per step")
