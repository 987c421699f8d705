"""Latency of the reference's real per-step call `policy.net(...) -> List[str]` (prefill + greedy generation with the KV cache,
max_new_tokens=20 like VLN-POL:463) at batch 8, next to the first-token path `forward_logits`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device="cuda"), device="cuda", batch_size=B, max_steps=24)
net.feature_fields.initialize_camera_setting(90.0, 90.0)
ep = SyntheticEpisodes(B, seed=0)
frames = []
for _ in range(16):
    fr = ep.next()
    frames.append((dict(rgb=torch.from_numpy(fr.rgb).cuda(), depth=torch.from_numpy(fr.depth).cuda()), [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm))
def step(i, n_new):
    o, p, h, s = frames[i]
    return net(o, [INSTRUCTION_64] * B, p, h, patch_segm=s, max_new_tokens=n_new)
for i in range(8):
    o, p, h, s = frames[i]; net.forward_logits(o, [INSTRUCTION_64] * B, p, h, patch_segm=s)
res = {}
step(8, 20)                                             # untimed: first 20-token call allocates the KV buffers
for n_new, idx in ((1, (8, 9, 10, 11)), (20, (12, 13, 14, 15))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in idx:
        step(i, n_new)
    torch.cuda.synchronize(); res[n_new] = (time.perf_counter() - t0) / len(idx) * 1e3
print(f"forward(max_new_tokens=1)  {res[1]:.1f} ms/step")
print(f"forward(max_new_tokens=20) {res[20]:.1f} ms/step  -> {(res[20] - res[1]) / 19:.2f} ms per further token (8 sequences), {8 * 20 / res[20] * 1e3:.0f} generated tokens/s")
