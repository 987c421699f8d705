"""Run by tests/test_gpu_fresh_process.py in a FRESH interpreter: the full-configuration step (B = 8, both ViT-L/14@336 towers, 3D-token builder,
Phi-3-mini x 32 packed prefill) on memory steps 0 and 1 of seeded synthetic episodes with seeded device-generated weights; prints one line per
step with the SHA-256 of the float32 logits.  No oracle, no tolerance: two processes must print the same lines."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes

D.enable_hip_kernels(["all"])
D.strict(True)
cfg, B = PolicyConfig(), 8
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device="cuda"), device="cuda", batch_size=B, max_steps=8)
net.feature_fields.initialize_camera_setting(90.0, 90.0)
ep = SyntheticEpisodes(B, seed=11)
for step in range(2):
    fr = ep.next()
    obs = dict(rgb=torch.from_numpy(fr.rgb).cuda(), depth=torch.from_numpy(fr.depth).cuda())
    x, lens = net.build_inputs(obs, [INSTRUCTION_64] * B, [p.tolist() for p in fr.positions], list(fr.headings), patch_segm=fr.patch_segm, return_rows="packed")
    lo = net.llm.prefill_logits_packed(x, lens).float()            # (no synchronisation between the builder's streams and the prefill: as the step runs)
    b, r0 = B - 1, sum(lens[:-1])                                  # the LAST prompt alone (its rows sit in other tiles, other K splits than inside the batch)
    xs = x.new_zeros(((lens[b] + 255) // 256 * 256, x.shape[1]))
    xs[:lens[b]] = x[r0:r0 + lens[b]]
    single = net.llm.prefill_logits_packed(xs, [lens[b]])[0].float()
    rel = float((single - lo[b]).norm() / lo[b].norm())
    assert torch.isfinite(lo).all() and rel < 3e-2, rel            # == the same prompt inside the packed batch, up to GEMM tile-shape effects (16-bit band)
    print(f"step {step} lens {lens} sha256 {hashlib.sha256(lo.cpu().numpy().tobytes()).hexdigest()} alone-vs-packed {rel:.3e}", flush=True)
assert not D.counts()["fallback"], D.counts()
