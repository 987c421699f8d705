"""cProfile of the host side of one warm 3D-token update (B = 8) running on the GPU: where do the 4 ms of wall time go?"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynam3d_amd.feature_fields import Feature_Fields
from dynam3d_amd.weights import ff_param_spec, synth_state_dict
from dynam3d_amd.synthetic import SyntheticEpisodes

B = 8
ff = Feature_Fields(B, "cuda", synth_state_dict(ff_param_spec(), 0), max_steps=20)
ep = SyntheticEpisodes(B, seed=0)
pr = cProfile.Profile()
import time
for t in range(14):
    fr = ep.next()
    depth = torch.from_numpy(fr.depth).cuda()[..., 0]
    dfull = ff.ops.preprocess_depth(depth).view(B, 1, 224, 224)
    d24 = ff.ops.resize_nearest_preprocess(depth, 24, 24).view(B, 1, 576)
    grid = torch.randn(B, 1, 576, 768, device="cuda").half()
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    ff.delete_old_features_from_camera_frustum(dfull, pos, hd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if t >= 9:
        pr.enable()
    ff.update_feature_fields(d24, grid, None, pos, hd, patch_segm=fr.patch_segm)
    if t >= 9:
        pr.disable()
    torch.cuda.synchronize()
    if t >= 9:
        print(f"step {t}: update_feature_fields {1e3 * (time.perf_counter() - t0):.2f} ms")
    ff.get_environment_features(pos, hd)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
