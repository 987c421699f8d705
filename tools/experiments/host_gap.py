"""Host time between the 3D-token update's last device->host read and the first Phi-3 launches (the GPU idles through it: ~1 ms in
profiles/r06_step_timeline.txt): perf_counter stamps around the host sections of a warm step, mean over the timed steps, un-profiled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
from dynam3d_amd import towers, policy

D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=40)
net.feature_fields.initialize_camera_setting(90., 90.)
ep = SyntheticEpisodes(B, seed=0)
instr = [INSTRUCTION_64] * B
frames = []
for _ in range(28):
    fr = ep.next()
    frames.append((dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev)), [p.tolist() for p in fr.positions], list(fr.headings), fr.patch_segm))
stamps = {}
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        stamps.setdefault(tag, []).append((t0, time.perf_counter()))
        return r
    setattr(obj, name, g)
wrap(net.feature_fields, "update_feature_fields", "update")
wrap(net.feature_fields, "get_environment_features", "query")
wrap(net, "_assemble_packed", "assemble")
wrap(net.llm, "packed_context", "context")
wrap(net.llm, "layer_packed", "layer")
wrap(net, "build_inputs", "build_inputs")
wrap(net.llm, "prefill_logits_packed", "prefill")
for i in range(28):
    if i == 13:
        torch.cuda.synchronize(); t_start = time.perf_counter()
    obs, pos, hd, segm = frames[i]
    net.forward_logits(obs, instr, pos, hd, patch_segm=segm)
torch.cuda.synchronize()
n = 15
print(f"{(time.perf_counter() - t_start) / n * 1e3:.2f} ms per step over {n} un-synchronised steps")
def mean(tag, k=1, which=None):
    v = stamps[tag][-n * k:]
    return v
upd, qry, asm_, ctx, bi, pf = (stamps[t][-n:] for t in ("update", "query", "assemble", "context", "build_inputs", "prefill"))
lay = stamps["layer"][-n * 32:]
d = lambda xs: 1e3 * float(np.mean(xs))
print("host ms: update call %.2f | query call %.2f | query end -> assemble start (prefix MLPs, cats) %.2f | assemble %.2f | build_inputs end -> context start %.2f | "
      "context %.2f | context end -> layer 0 launched %.2f | 32 layers enqueue %.2f | prefill call total %.2f"
      % (d([b - a for a, b in upd]), d([b - a for a, b in qry]), d([asm_[i][0] - qry[i][1] for i in range(n)]), d([b - a for a, b in asm_]),
         d([ctx[i][0] - bi[i][1] for i in range(n)]), d([b - a for a, b in ctx]), d([lay[32 * i][1] - ctx[i][1] for i in range(n)]),
         d([lay[32 * i + 31][1] - lay[32 * i][0] for i in range(n)]), d([b - a for a, b in pf])))
print("host ms from query return to layer 0 launched: %.2f" % d([lay[32 * i][1] - qry[i][1] for i in range(n)]))
