"""The 3D-memory update alone (frustum deletion + update_feature_fields + get_environment_features) with the bookkeeping planned on the
host (csrc/ff_state.cpp) and on the device (csrc/ff_plan_kernels.hip): wall ms per step at a warm memory, device-to-host reads per
step, and that both leave bit-identical stores.  B = 8 (the bench's batch) and B = 1 (configs[1])."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dynam3d_amd.feature_fields import Feature_Fields
from dynam3d_amd.synthetic import SyntheticEpisodes
from dynam3d_amd.weights import ff_param_spec, synth_state_dict

STEPS, WARM = int(os.environ.get("FF_STEPS", "16")), int(os.environ.get("FF_WARM", "8"))
sd = synth_state_dict(ff_param_spec(), 0)
reads = [0]
_cpu = torch.Tensor.cpu


def counting_cpu(self, *a, **k):
    if self.is_cuda:
        reads[0] += 1
    return _cpu(self, *a, **k)


torch.Tensor.cpu = counting_cpu


def frames(B):
    ep = SyntheticEpisodes(B, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    out = []
    for t in range(STEPS):
        fr = ep.next()
        depth = torch.from_numpy(fr.depth).cuda()
        out.append((fr, depth, torch.randn(B, 1, 576, 768, device="cuda", generator=g).half()))
    return out


PLANNERS = os.environ.get("FF_PLANNERS", "host,device,host,device").split(",")
for B in [int(b) for b in os.environ.get("FF_BATCHES", "8,1").split(",")]:
    fs = frames(B)
    res = {}
    for planner in PLANNERS:
        ff = Feature_Fields(B, "cuda", sd, max_steps=STEPS + 2, planner=planner)
        ff.initialize_camera_setting(90.0, 90.0)
        ops = ff.ops
        tot, rd = 0.0, 0
        for t, (fr, depth, grid) in enumerate(fs):
            dfull = ops.preprocess_depth(depth[..., 0]).view(B, 1, depth.shape[1], depth.shape[2])
            d24 = ops.resize_nearest_preprocess(depth[..., 0], 24, 24).view(B, 1, 576)
            pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
            torch.cuda.synchronize()
            r0, t0 = reads[0], time.perf_counter()
            ff.delete_old_features_from_camera_frustum(dfull, pos, hd)
            ff.update_feature_fields(d24, grid, None, pos, hd, patch_segm=fr.patch_segm)
            ev = ff.get_environment_features(pos, hd)
            torch.cuda.synchronize()
            if t >= WARM:
                tot += time.perf_counter() - t0
                rd += reads[0] - r0
        n = STEPS - WARM
        key = planner
        res.setdefault(key, []).append((tot / n * 1e3, rd / n))
        res[key + "_state"] = (ff.pools.inst_fts.clone(), ff.pools.zone_fts.clone(), ff.pools.inst_pos.clone(), [x.clone() for x in ev["batch_instance_fts"]])
    for k in sorted(set(PLANNERS)):
        print(f"B={B} planner={k:6s}: " + " / ".join(f"{ms:.2f} ms per step, {r:.1f} device-to-host reads" for ms, r in res[k]), flush=True)
    if "host_state" in res and "device_state" in res:
        same = all(torch.equal(a, b) for a, b in zip(res["host_state"][:3], res["device_state"][:3])) and \
            all(torch.equal(a, b) for a, b in zip(res["host_state"][3], res["device_state"][3]))
        print(f"B={B}: stores and get_environment_features outputs bit-identical between the two planners: {same}", flush=True)
