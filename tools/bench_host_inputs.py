"""Step rate when the observations arrive as HOST tensors (PCIe-inclusive) next to the HBM-resident rate bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device="cuda"), device="cuda", batch_size=B, max_steps=32)
net.feature_fields.initialize_camera_setting(90.0, 90.0)
ep = SyntheticEpisodes(B, seed=0)
frames = [ep.next() for _ in range(26)]
def run(i, host):
    fr = frames[i]
    rgb, depth = torch.from_numpy(fr.rgb), torch.from_numpy(fr.depth)
    if not host:
        rgb, depth = rgb.cuda(), depth.cuda()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    lo = net.forward_logits(dict(rgb=rgb, depth=depth), [INSTRUCTION_64] * B, [p.tolist() for p in fr.positions], list(fr.headings), patch_segm=fr.patch_segm)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for i in range(10): run(i, False)
res = {}
for host, idx in ((False, range(10, 18)), (True, range(18, 26))):
    res[host] = sum(run(i, host) for i in idx) / 8 * 1e3
print(f"inputs resident in HBM: {res[False]:.2f} ms/step ({8e3 / res[False]:.1f} env-steps/s)   host tensors (H2D inside the step): {res[True]:.2f} ms/step ({8e3 / res[True]:.1f} env-steps/s)")
