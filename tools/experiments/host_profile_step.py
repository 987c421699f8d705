"""cProfile of the host side of warm steps (un-synchronised loop): which Python functions pace the launch stream."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=40)
net.feature_fields.initialize_camera_setting(90., 90.)
ep = SyntheticEpisodes(B, seed=0)
instr = [INSTRUCTION_64] * B
frames = []
for _ in range(26):
    fr = ep.next()
    frames.append((dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev)), [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm))
for i in range(13):
    net.forward_logits(frames[i][0], instr, frames[i][1], frames[i][2], patch_segm=frames[i][3])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(13, 26):
    net.forward_logits(frames[i][0], instr, frames[i][1], frames[i][2], patch_segm=frames[i][3])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
