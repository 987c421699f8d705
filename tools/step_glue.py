"""Which kernels of one warm step are NOT ours?  Run under `rocprofv3 --kernel-trace`: the script brackets the last step with two marker
launches (torch.erfinv of 777 elements) so that tools/step_glue_report.py can cut the step out of the trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes

D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=32)
net.feature_fields.initialize_camera_setting(90., 90.)
ep = SyntheticEpisodes(B, seed=0)
instr = [INSTRUCTION_64] * B
for step in range(13):
    fr = ep.next()
    obs = dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev))
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    if step == 12:
        torch.cuda.synchronize(); torch.erfinv(torch.zeros(777, device=dev)); torch.cuda.synchronize()
    net.forward_logits(obs, instr, pos, hd, patch_segm=fr.patch_segm)
    if step == 12:
        torch.cuda.synchronize(); torch.erfinv(torch.zeros(777, device=dev)); torch.cuda.synchronize()
