"""Pre-training step (PRE-FF:843-1345 is_training=True + PRE-TR:479-526) on one MI355X: ms per step and where it goes.

    python tools/bench_pretrain.py [--batch 8] [--steps 6] [--n-gt 200000]  ->  one JSON line (profiles/rNN_bench_pretrain.json)

The step = frustum delete + 4-view memory update with loss collection (set encoders forward on the float32 MFMA GEMM, GT labelling by
d3d_knn over the GT cloud, ground-truth merges) + backward (dx / dW GEMMs, LayerNorm / GELU / set-attention backward kernels) + gradient
all-reduce (world 1: none) + AdamW + weight sync.  The GEMM share is measured with HIP events around every d3d_gemm_nt_f32 call
(forward, dx, dW) of the timed steps; its FLOPs are 2 M N K of each call (algorithmic: the split kernel issues three MFMAs per product)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynam3d_amd import train_ff as TF  # noqa: E402
from dynam3d_amd.feature_fields import Feature_Fields  # noqa: E402
from dynam3d_amd.ops import HipOps  # noqa: E402
from dynam3d_amd.synthetic import SyntheticEpisodes  # noqa: E402
from dynam3d_amd.weights import ff_param_spec, synth_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-gt", type=int, default=200000)
    a = ap.parse_args()
    dev, B, V, vids = "cuda", a.batch, 4, [0, 3, 6, 9]
    sd = synth_state_dict(ff_param_spec(), seed=0)
    ff = Feature_Fields(B, device=dev, state_dict=sd, ops=HipOps(), max_steps=(a.steps + a.warmup + 1) * V, variant="pretrain")
    ff.initialize_camera_setting(90.0, 90.0)
    model = TF.TrainableFF(sd, dev)
    rng = np.random.default_rng(0)
    gts = []
    for b in range(B):
        xyz = np.stack([rng.uniform(-9, 9, a.n_gt), rng.uniform(-9, 9, a.n_gt), rng.uniform(-3, 4, a.n_gt)], 1).astype(np.float32)
        gts.append((xyz, (np.floor(xyz[:, 0] / 1.5).astype(np.int64) + 8) * 64 + (np.floor(xyz[:, 1] / 1.5).astype(np.int64) + 8)))
    trainer = TF.FFTrainer(model, [g[0] for g in gts], [g[1] for g in gts])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    eps = [SyntheticEpisodes(B, seed=40 + v, image_hw=32, depth_hw=64) for v in range(V)]
    from oracle_free_inputs import step_inputs  # noqa: E402  (below: no oracle import in a bench tool)
    gemm = dict(ms=0.0, flop=0.0, n=0, on=False, ev=[])
    orig = TF.gemm_nt_f32

    def timed_gemm(x, w):
        if not gemm["on"]:
            return orig(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, w)
        e1.record()
        gemm["ev"].append((e0, e1))
        gemm["flop"] += 2.0 * x.shape[0] * w.shape[0] * x.shape[1]
        gemm["n"] += 1
        return y

    TF.gemm_nt_f32 = timed_gemm
    times, losses, phases = [], [], {}
    for t in range(a.warmup + a.steps):
        inp = step_inputs(eps, rng, B, V)
        gemm["on"] = t >= a.warmup
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ff.delete_old_features_from_camera_frustum(torch.from_numpy(inp["depth_full"]), inp["positions"], inp["headings"], view_ids=vids)
        res = TF.pretrain_step(ff, trainer, opt, dict(batch_depth=inp["depth24"], batch_grid_ft=inp["grid"], batch_position=inp["positions"],
                                                      batch_heading=inp["headings"], patch_segm=inp["patch_segm"], view_ids=vids, batch_image_ft=inp["img"]),
                               timing=phases if (os.environ.get("BENCH_PRETRAIN_PHASES") == "1" and t >= a.warmup) else None)
        torch.cuda.synchronize()
        if t >= a.warmup:
            times.append((time.perf_counter() - t0) * 1e3)
            losses.append(res["loss"])
    gemm["ms"] = sum(e0.elapsed_time(e1) for e0, e1 in gemm["ev"])
    ms = float(np.mean(times))
    n_par = sum(p.numel() for p in model.parameters())
    print(json.dumps(dict(what="pre-training step of the feature field (f-1): delete + 4-view update with losses + backward + AdamW, float32, 1 MI355X",
                          batch=B, views=V, patches_per_step=B * V * 576, gt_points_per_env=a.n_gt, steps=a.steps, ms_per_step=round(ms, 2),
                          ms_per_step_each=[round(x, 2) for x in times], frames_per_s=round(B * V / ms * 1e3, 1), trainable_parameters=n_par,
                          gradient_bytes=n_par * 4, train={"gemm_calls_per_step": gemm["n"] // a.steps, "gemm_ms_per_step": round(gemm["ms"] / a.steps, 2),
                                                           "gemm_fraction_of_step": round(gemm["ms"] / a.steps / ms, 3),
                                                           "gemm_gflop_per_step": round(gemm["flop"] / a.steps / 1e9, 1),
                                                           "gemm_tflops_fp32": round(gemm["flop"] / (gemm["ms"] * 1e-3) / 1e12, 1),
                                                           "fp32_mfma_peak_tflops": 157.3, "gemm_frac_of_fp32_mfma_peak": round(gemm["flop"] / (gemm["ms"] * 1e-3) / 157.3e12, 3),
                                                           "kernel": "d3d_gemm_nt_f32x3 (row-scaled fp16 hi + lo split, three MFMAs) incl. its two d3d_row_exponents launches; D3D_TRAIN_GEMM_SPLIT=0: d3d_gemm_nt_f32 -- forward, dx and dW of every Linear"},
                          phases_ms_per_step=({k: round(v / a.steps, 2) for k, v in phases.items()} if phases else "BENCH_PRETRAIN_PHASES=1 (synchronising) to measure"),
                          loss_first_last=[round(losses[0], 4), round(losses[-1], 4)], collectives_per_step=res["collectives"])))


if __name__ == "__main__":
    main()
