"""GPU: the step AT THE BENCHMARK'S CONFIGURATION -- B = 8, ViT-L/14@336 x 2, Phi-3-mini at full width and all 32 layers -- held to
properties that do not need a second implementation (the oracle cannot run this size in test time; per-layer parity at these shapes
is tests/test_gpu_depth_parity.py):

  * `PRUNE_LAST_LAYER` on == off on the B logit rows up to GEMM-kernel summation order (the last layer's o_proj / MLP are row-wise:
    evaluating them on the last row only must not change that row; 8 rows take the weight-streaming kernel, 6.9 k rows the 256-tile
    kernel, so single bf16 roundings may flip) -- VLN-POL:448-463 reads only those rows;
  * determinism: the same packed prompt twice -> identical logits (split-K partial sums are reduced in slice order);
  * packed == per-prompt: every prompt prefilled alone gives its packed logits up to GEMM tile-shape effects (different M ->
    different split of K), inside the 16-bit band of a 32-layer run, and the same first token wherever the top-2 margin is outside it;
  * config[3] at one rank's size: 8 concurrent episodes x 50 steps of the full model with episode pops and pool growth
    (VLN-TR:389-408, 778-784) -- every episode ends, the 3D memory of the survivors stays consistent, prompts stay finite."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_net():
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
    saved, was = dict(D.BACKEND), D.STRICT
    D.enable_hip_kernels(["all"])
    D.strict(True)
    cfg = PolicyConfig()
    net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device="cuda"), device="cuda", batch_size=8, max_steps=52)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    yield net
    D.strict(was)
    D.BACKEND.update(saved)


def _frame(ep, dev="cuda"):
    fr = ep.next()
    return (dict(rgb=torch.from_numpy(fr.rgb).to(dev), depth=torch.from_numpy(fr.depth).to(dev)), [p.tolist() for p in fr.positions],
            list(fr.headings), fr.patch_segm)


def test_full_config_step_prune_determinism_packed_vs_single(full_net):
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    net, B = full_net, 8
    net.feature_fields.reset(B)
    ep = SyntheticEpisodes(B, seed=11)
    instr = [INSTRUCTION_64] * B
    D.reset_counts()
    for step in range(2):
        obs, pos, hd, segm = _frame(ep)
        x, lens = net.build_inputs(obs, instr, pos, hd, patch_segm=segm, return_rows="packed")
        llm = net.llm
        assert llm.cfg.layers == 32 and llm.cfg.hidden == 3072 and x.shape[0] % 256 == 0
        llm.PRUNE_LAST_LAYER = True
        lo_a = llm.prefill_logits_packed(x, lens)
        lo_b = llm.prefill_logits_packed(x, lens)
        llm.PRUNE_LAST_LAYER = False
        try:
            lo_full = llm.prefill_logits_packed(x, lens)
        finally:
            llm.PRUNE_LAST_LAYER = True
        assert torch.isfinite(lo_a).all()
        assert torch.equal(lo_a, lo_b), "the packed prefill is not deterministic"
        # pruned: the last layer's o_proj / MLP run on 8 rows through the weight-streaming M <= 16 GEMM kernel, unpruned on 6.9 k rows through
        # the 256-tile kernel: same arithmetic per row, another float32 summation order -> a few bf16 roundings flip by one ulp
        r_prune = float((lo_a - lo_full).norm() / lo_full.norm())
        print(f"step {step}: PRUNE_LAST_LAYER on vs off: logits rel-L2 {r_prune:.2e}, max |d| {float((lo_a - lo_full).abs().max()):.4f}")
        assert r_prune < 3e-3 and torch.equal(lo_a.argmax(-1), lo_full.argmax(-1)), r_prune
        cu = np.concatenate([[0], np.cumsum(lens)])
        worst = 0.0
        for b in range(B):
            xs = torch.zeros(((lens[b] + 255) // 256 * 256, x.shape[1]), dtype=x.dtype, device=x.device)
            xs[:lens[b]] = x[cu[b]:cu[b + 1]]
            lo_1 = llm.prefill_logits_packed(xs, [lens[b]])[0]
            r = float((lo_1 - lo_a[b]).norm() / lo_a[b].norm())
            worst = max(worst, r)
            top2 = torch.sort(lo_a[b]).values[-2:]
            if float(top2[1] - top2[0]) > 0.05 * float(lo_a[b].abs().max()):
                assert int(lo_1.argmax()) == int(lo_a[b].argmax())
        print(f"step {step}: S = {lens}; packed vs per-prompt logits rel-L2 (worst of {B}) {worst:.2e}")
        assert worst < 3e-2, worst        # (32 layers deep: GEMM tile shapes depend on M -> bf16 roundings flip and decorrelate, like any two 16-bit runs)
    c = D.counts()
    assert not c["fallback"], c


def test_packed_prompt_kernel_equals_torch_assembly(full_net):
    """`d3d_assemble_prompt` (one pass over the row table) writes exactly the rows the PyTorch expressions of `_assemble_packed` build
    (embedding gather, float32 patch add rounded once, cat, row gather -- VLN-POL:448-456), padding rows included, at the bench shapes."""
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    net, B = full_net, 8
    net.feature_fields.reset(B)
    ep = SyntheticEpisodes(B, seed=4)
    instr = [INSTRUCTION_64[: 40 + 7 * b] for b in range(B)]                     # ragged instruction lengths
    for step in range(3):
        obs, pos, hd, segm = _frame(ep)
        net.feature_fields.history_actions = [["forward\n"] * (step + b % 2) + ["none\n"] * 2 for b in range(B)]       # ragged tails
        calls = {}
        orig = net._assemble_packed

        def spy(*a, **k):
            calls["args"] = a
            return orig(*a, **k)
        net._assemble_packed = spy
        try:
            x_k, lens_k = net.build_inputs(obs, instr, pos, hd, patch_segm=segm, return_rows="packed")
        finally:
            net._assemble_packed = orig
        type(net).ASSEMBLE_KERNEL = False
        try:
            x_t, lens_t = orig(*calls["args"])
        finally:
            type(net).ASSEMBLE_KERNEL = True
        assert lens_k == lens_t and x_k.shape == x_t.shape and x_k.dtype == x_t.dtype
        assert torch.equal(x_k, x_t), float((x_k.float() - x_t.float()).abs().max())
        assert float(x_k[sum(lens_k):].abs().max()) == 0.0


def test_whole_step_is_bitwise_repeatable(full_net):
    """RGB-D -> logits, three warm steps at the benchmark's configuration, run twice from a reset memory on the same frames: identical
    logits bit for bit.  Covers what the per-kernel tests cannot: stream hand-offs (frustum cull / llava tower on side streams), the
    3D-token update's host round trips, split-precision GEMMs, LDS-DMA attention, split-K reductions -- any race, uninitialised read or
    missing hazard wait between them shows up as a last-bit difference here (the inline-asm MFMA hazard of round 3 did)."""
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    net, B = full_net, 8
    instr = [INSTRUCTION_64] * B
    runs = []
    for rep in range(2):
        net.feature_fields.reset(B)
        ep = SyntheticEpisodes(B, seed=21)
        outs = []
        for step in range(3):
            obs, pos, hd, segm = _frame(ep)
            outs.append(net.forward_logits(obs, instr, pos, hd, patch_segm=segm).clone())
        runs.append(outs)
    for step, (a, b) in enumerate(zip(*runs)):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (step, float((a - b).abs().max()))


def test_config3_rollout_8_episodes_50_steps_full_model(full_net):
    """BASELINE configs[3] on one rank: 8 concurrent episodes, max_traj_len 50 (VLN/scripts/iter_train.yaml:41), the full model."""
    from dynam3d_amd.rollout import run_rollout
    net = full_net
    sums, done = run_rollout(net, episodes=8, max_steps=50, seed=5, stop_token_mod=41)      # ~2.4 % stop chance per step: pops at scattered steps
    assert done == 8
    assert 8 <= sums["steps_taken"] <= 8 * 50
    assert net.feature_fields.batch_size == 0                                                # every episode was popped (VLN-TR:778-784)
    # full-length episodes: the pools grow to 50 x 576 rows per environment and the prompts stay inside the model's window
    sums2, done2 = run_rollout(net, episodes=8, max_steps=50, seed=6, stop_token_mod=10 ** 9)
    assert done2 == 8 and sums2["steps_taken"] == 8 * 50
    assert max(net.last_lengths) < net.llm.SLIDING_WINDOW
    print("config[3] one-rank rollout: early-stop run", sums, "; full-length run: last S =", net.last_lengths)
