"""Where does Feature_Fields.update_feature_fields spend its time?  Wraps every ops/dense/state call with a synchronised timer."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd.feature_fields import Feature_Fields
from dynam3d_amd.weights import ff_param_spec, synth_state_dict
from dynam3d_amd.synthetic import SyntheticEpisodes

B = 8
ff = Feature_Fields(B, "cuda", synth_state_dict(ff_param_spec(), 0), max_steps=20)
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[tag + name] += time.perf_counter() - t0; cnt[tag + name] += 1
        return r
    setattr(obj, name, w)
for n in ("unproject_append", "append_fts", "group_stats7", "gather_fts", "knn", "merge_input", "scatter_rows", "group_stats4", "gather_rows", "fill_rows", "frustum_cull", "agent_frame_compact"):
    wrap(ff.ops, n, "ops.")
for n in ("encode_patch_sets", "encode_zone_sets", "merge_logits"):
    wrap(ff.dense, n, "dense.")
for n in ("plan_merge", "plan_zones", "begin_view", "end_view", "apply_hits", "live_ids"):
    wrap(ff.state, n, "state.")
ep = SyntheticEpisodes(B, seed=0)
rng = np.random.default_rng(0)
tot = 0
for t in range(12):
    fr = ep.next()
    depth = torch.from_numpy(fr.depth).cuda()[..., 0]
    dfull = ff.ops.preprocess_depth(depth).view(B, 1, 224, 224)
    d24 = ff.ops.resize_nearest_preprocess(depth, 24, 24).view(B, 1, 576)
    grid = torch.randn(B, 1, 576, 768, device="cuda").half()
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    if t == 8:
        acc.clear(); cnt.clear(); tot = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ff.delete_old_features_from_camera_frustum(dfull, pos, hd)
    ff.update_feature_fields(d24, grid, None, pos, hd, patch_segm=fr.patch_segm)
    ff.get_environment_features(pos, hd)
    torch.cuda.synchronize(); tot += time.perf_counter() - t0
n = 4
print(f"total per step {tot / n * 1e3:.2f} ms (with per-call syncs)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / n * 1e3:7.3f} ms/step  ({cnt[k] / n:.1f} calls)")
print(f"  {'(unaccounted python/H2D)':28s} {(tot - sum(acc.values())) / n * 1e3:7.3f} ms/step")
