"""Pretrain novel-view render path (a20-a23) at full size: B views of 12 x 12 rays x 501 samples against N stored patches per environment
(72 144 queries x N points per view), then the tcnn-style networks on 1 152 samples per view.  Prints wall time per stage (synchronised) and
the fused-vs-unfused MLP time; under rocprofv3 --kernel-trace the per-kernel durations (profiles/r03_render_*)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynam3d_amd import tcnn
from dynam3d_amd.ops import HipOps, Pools
from dynam3d_amd.render import FieldRenderer
from dynam3d_amd.weights import ff_param_spec, render_param_spec, synth_state_dict
B = int(os.environ.get("B", "8")); N = int(os.environ.get("N", "9216"))
ops = HipOps()
rng = np.random.default_rng(0)
pools = Pools.allocate(B, N + 64, 8, 8, "cuda")
for b in range(B):
    pos = rng.uniform(-4, 4, (N, 3)).astype(np.float32); pos[:, 2] = rng.uniform(-1, 1.5, N)
    pools.rows_pos[b, :N] = torch.from_numpy(pos).cuda()
    pools.rows_dir[b, :N] = torch.from_numpy(rng.uniform(0, 6.28, N).astype(np.float32)).cuda()
    pools.rows_scale[b, :N] = torch.from_numpy(rng.uniform(0.01, 0.2, N).astype(np.float32)).cuda()
    f = rng.standard_normal((N, 768)).astype(np.float32); f /= np.linalg.norm(f, axis=1, keepdims=True)
    pools.rows_fts[b, :N] = torch.from_numpy(f).cuda().half()
sd = synth_state_dict(ff_param_spec() + render_param_spec(), seed=0)
r = FieldRenderer(sd, "cuda")
posn = [[float(rng.uniform(-1, 1)), 0.0, float(rng.uniform(-1, 1))] for _ in range(B)]
head = [float(rng.uniform(0, 6.28)) for _ in range(B)]
def run():
    return r.render(pools, list(range(B)), [N] * B, posn, head, ops)
def timed_calls():
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
if os.environ.get("RENDER_AB", "1") == "1":                      # radius-limited KNN (default) against the exact brute-force query, alternating
    ab = {True: [], False: []}
    for rep in range(3):
        for flag in (False, True):
            FieldRenderer.RADIUS_KNN = flag
            ab[flag].append(timed_calls())
    FieldRenderer.RADIUS_KNN = True
    same = all(torch.equal(a, b) for a, b in zip(run()[:2], (lambda: (setattr(FieldRenderer, "RADIUS_KNN", False), run(), setattr(FieldRenderer, "RADIUS_KNN", True))[1])()[:2]))
    print(f"render call, brute-force d3d_knn: {'/'.join(f'{t:.2f}' for t in ab[False])} ms | d3d_knn_radius: {'/'.join(f'{t:.2f}' for t in ab[True])} ms | identical feature map + positions: {same}")
ms = timed_calls()
print(f"render_view_3d_patch B={B} views, N={N} patches/env: {ms:.2f} ms per call ({ms / B:.2f} ms per view)")
x = (torch.randn(B * 1152, 768, device="cuda") * 0.5).half()
def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rows in (1152, B * 1152):
    xs = x[:rows]
    res = {}
    for fused in (True, False, True, False):
        tcnn.FUSED = fused
        with torch.no_grad():
            res.setdefault(fused, []).append(tm(lambda: r.encoder(xs)))
    fl = 2.0 * rows * (768 * 768 * 2 + 768 * 769)
    print(f"tcnn encoder 768-768-768-769 on {rows} rows: fused (1 launch) {min(res[True]):.1f} us ({fl / min(res[True]) / 1e6:.0f} TF/s), "
          f"3 GEMM launches {min(res[False]):.1f} us ({fl / min(res[False]) / 1e6:.0f} TF/s)")
tcnn.FUSED = False
