"""GPU: the benchmark's golden parity point (tests/golden/g22_bench_point.npz: B = 8, memory step 13, full configuration, CPU oracle in
float32 and with the reference's 16-bit rounding points) replayed exactly as bench.py's default command replays it behind its timed
region -- the product path inside the 16-bit noise band of the lowp oracle (hard), the float32 verification mode within north_star's 1e-3
of the float32 oracle, exact bookkeeping.  Reference: VLN-POL:329-363, 430-463."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bench_golden_point_product_in_band_and_float32_mode_within_1e3():
    import bench
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
    cfg = PolicyConfig()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(32, max(8, os.cpu_count() or 8)))
    try:
        sd = synth_policy_weights(cfg, 0)
    finally:
        torch.set_num_threads(threads_before)
    D.enable_hip_kernels(["all"])
    was = D.STRICT
    D.strict(True)
    try:
        net = Dynam3D_VLN(cfg, sd, device="cuda", batch_size=8, max_steps=16)
        out = bench.golden_point_parity(net, cfg, sd, "cuda", 0)
    finally:
        D.strict(was)
    print(out)
    assert out["bookkeeping_exact"] and out["f32_mode"]["bookkeeping_exact"], out
    assert out["within_band_vs_lowp"] and out["within_band_vs_f32"], out
    assert out["f32_mode"]["north_star_1e3_met"] and out["f32_mode"]["fallbacks"] == 0, out
    assert out["ok"] is True
