"""Per-stage wall time of one warm step (synchronising between stages, so no overlap): where does the step go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynam3d_amd import dense_ops as D
from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
from dynam3d_amd.towers import preprocess_rgb

D.enable_hip_kernels(["all"])
cfg = PolicyConfig(); B = 8; dev = "cuda"
net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device=dev), device=dev, batch_size=B, max_steps=32)
net.feature_fields.initialize_camera_setting(90., 90.)
ep = SyntheticEpisodes(B, seed=0)
instr = [INSTRUCTION_64] * B
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for step in range(14):
    fr = ep.next()
    obs = dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev))
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    if step < 9:
        net.forward_logits(obs, instr, pos, hd, patch_segm=fr.patch_segm); continue
    ff = net.feature_fields
    t = [sync()]
    depth = obs["depth"]; d24 = net._depth24(depth, 1, (0., 10.)); px = preprocess_rgb(obs["rgb"]); t.append(sync())
    _, grid = net.rgb_encoder.forward(px); t.append(sync())
    dfull = net.ops.preprocess_depth(depth[..., 0], 0., 10.).view(B, 1, 224, 224)
    ff.delete_old_features_from_camera_frustum(dfull, pos, hd); t.append(sync())
    ff.update_feature_fields(d24, grid.view(B, 1, 576, -1), None, pos, hd, patch_segm=fr.patch_segm); t.append(sync())
    env = ff.get_environment_features(pos, hd); info = ff.get_patch_3d_info(d24.reshape(B, -1)); t.append(sync())
    pf = net.llava_vision.forward(px); t.append(sync())
    names = ["prep", "clip_vit", "ff_delete", "ff_update", "ff_query", "llava_vit"]
    for n, a, b in zip(names, t[:-1], t[1:]):
        acc.setdefault(n, []).append((b - a) * 1e3)
    # LM on a fresh full build (state already advanced -> use the logits path of the NEXT frame separately)
from dynam3d_amd.profiling import TIMER
TIMER.enabled = True
TIMER.reset()
for step in range(3):
    fr = ep.next()
    obs = dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev))
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    t0 = sync(); x, L = net.build_inputs(obs, instr, pos, hd, patch_segm=fr.patch_segm, return_rows="packed"); t1 = sync()
    lo = net.llm.prefill_logits_packed(x, L); t2 = sync()
    acc.setdefault("build_inputs_total", []).append((t1 - t0) * 1e3); acc.setdefault("phi3_prefill", []).append((t2 - t1) * 1e3)
for k, v in acc.items():
    print(f"{k:20s} {sum(v) / len(v):8.2f} ms   {['%.1f' % x for x in v]}")
torch.cuda.synchronize()
for k, (n, ms) in sorted(TIMER.summary().items()):
    print(f"  event-timed {k:22s} {ms:8.3f} ms  x{n // 3} per step")
