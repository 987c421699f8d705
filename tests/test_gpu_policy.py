"""GPU: whole hot path (policy.forward_logits through the C ABI kernels + towers) against the CPU step oracle.

Tolerances (relative L2 of the logits vector vs the float32 oracle):
  * float32 towers: 1e-3  (north_star's bound; measured ~1e-5) -- proves the pipeline/kernels, not the rounding;
  * fp16 CLIP + bf16 llava/Phi-3 (the reference's own dtypes): 3e-2.  bf16 has 8 mantissa bits (unit roundoff
    3.9e-3), so NO bf16 pipeline -- the reference's included -- can meet 1e-3 against a float32 oracle; the
    bound here is what 2 bf16 layers + bf16 storage of a 96-wide LM produce and is checked together with the
    argmax of every row."""
import numpy as np
import pytest
import torch

from tests.test_policy_cpu import SMALL, run_policy_vs_oracle

pytestmark = pytest.mark.gpu



def test_policy_fp32_matches_oracle():
    from dynam3d_amd.ops import HipOps
    worst = run_policy_vs_oracle(HipOps(), "cuda", SMALL, steps=3, B=2, tol=1e-3)
    assert worst < 1e-3


def test_policy_reference_dtypes_close_to_oracle():
    import dataclasses
    from dynam3d_amd.ops import HipOps
    cfg = dataclasses.replace(SMALL, clip_dtype=torch.float16, llava_dtype=torch.bfloat16)
    worst = run_policy_vs_oracle(HipOps(), "cuda", cfg, steps=2, B=2, tol=3e-2)
    assert worst < 3e-2


def test_policy_strict_hip_towers_match_lowp_oracle():
    """The whole step with EVERY dense primitive on a HIP kernel (strict mode: a PyTorch fallback raises; MID is the smallest
    configuration all of whose shapes are HIP-eligible), fp16 CLIP + bf16 llava / Phi-3, beside the step oracle in float32 and in
    `lowp` mode (the reference's rounding points in those dtypes, oracle/towers_ref.py).  A 16-bit evaluation sits a noise band away
    from float32 by construction (the lowp-vs-float32 distance, measured here); the HIP path must sit inside BAND x that band of
    BOTH oracles."""
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.ops import HipOps
    from tests.test_policy_cpu import MID, run_policy_three_way
    was = D.STRICT
    D.strict(True)
    D.reset_counts()
    try:
        d_lowp, d_f32, band = run_policy_three_way(HipOps(), "cuda", MID, steps=2, B=2)
        c = D.counts()
    finally:
        D.strict(was)
    assert not c["fallback"] and c["hip"]["linear"] > 0 and c["hip"]["attention"] > 0 and c["hip"]["vit_embed"] > 0, c
    print(f"MID step, strict HIP ({sum(c['hip'].values())} kernel dispatches, 0 fallbacks): vs lowp oracle {d_lowp:.2e}; vs float32 oracle {d_f32:.2e}; "
          f"lowp oracle vs float32 oracle (band) {band:.2e}")
    assert d_lowp < 1.25 * band and d_f32 < 1.25 * band, (d_lowp, d_f32, band)


def test_phi3_packed_varlen_prefill_matches_oracle():
    """Packed (no padding) Phi-3 prefill on the HIP kernels -- GEMM + fused epilogues, RMSNorm, RoPE with explicit positions,
    varlen flash attention -- vs the float32 oracle and vs the right-padded path."""
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
    from dynam3d_amd.weights import synth_state_dict
    from oracle import towers_ref as TR
    cfg = Phi3Config(vocab=512, hidden=384, layers=2, heads=4, kv_heads=4, mlp=512)       # head_dim 96 like Phi-3-mini
    sd = synth_state_dict(phi3_param_spec(cfg), seed=0)
    lens = [37, 300, 129, 64]
    g = torch.Generator().manual_seed(7)
    rows = [torch.randn(n, cfg.hidden, generator=g) * 0.5 for n in lens]
    emb = torch.zeros(len(lens), max(lens), cfg.hidden)
    for b, r in enumerate(rows):
        emb[b, :lens[b]] = r.to(torch.bfloat16).float()                       # oracle sees the same bf16-rounded inputs
    ref = TR.phi3_prefill_logits(emb, lens, sd, cfg.layers, cfg.heads, cfg.kv_heads, cfg.rms_eps, cfg.rope_theta).numpy()
    saved = dict(D.BACKEND)
    try:
        D.enable_hip_kernels(["all"])
        dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
        assert dec.interleave_gu and D.packed_ok(torch.bfloat16, cfg.head_dim)
        got = dec.prefill_logits_rows([r.cuda() for r in rows]).cpu().numpy()
        assert dec.last_packed_rows == 768                                   # 530 tokens -> 3 x 256 rows, no per-row padding
        pad = dec.prefill_logits(emb.cuda(), torch.tensor(lens)).cpu().numpy()
    finally:
        D.BACKEND.update(saved)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(got, ref) < 3e-2 and rel(pad, ref) < 3e-2 and rel(got, pad) < 2e-2, (rel(got, ref), rel(pad, ref), rel(got, pad))
    assert np.array_equal(got.argmax(-1), ref.argmax(-1))


def test_rollout_driver_pops_finished_episodes():
    """Config-4 shaped driver on one rank: episodes of different length, pop() compaction, single metric gather."""
    import dataclasses
    from dynam3d_amd import dist as DD
    from dynam3d_amd.policy import Dynam3D_VLN, synth_policy_weights
    from dynam3d_amd.rollout import run_rollout
    cfg = dataclasses.replace(SMALL, clip_dtype=torch.float16, llava_dtype=torch.bfloat16)
    from tests.test_policy_cpu import toy_dense
    with toy_dense(cfg, "cuda"):
        net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0), device="cuda", batch_size=3, max_steps=8)
        sums, n = run_rollout(net, 3, 5, seed=4, stop_token_mod=3)
    assert n == 3 and net.feature_fields.batch_size == 0
    res = DD.gather_metrics(sums, n, device="cuda")
    assert res["episodes"] == 3.0 and 1.0 <= res["steps_taken"] <= 5.0


def test_phi3_kv_cache_generation_matches_oracle():
    """Greedy generation with the KV cache (`Phi3Decoder.generate_packed`: prompt K/V read in place from the prefill buffers,
    generated tokens in the side cache, `d3d_decode_attention`) vs the float32 oracle that re-runs the whole prefix for every
    token (oracle/towers_ref.py::phi3_greedy_decode), on ragged prompts.  Teacher forcing with the oracle's tokens keeps both on
    the same path: logits of every step within the bf16 tolerance of the prefill test; the free-running product must pick
    the oracle's tokens wherever the oracle's top-2 margin is not within that tolerance, and stops at `end_id`."""
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
    from dynam3d_amd.weights import synth_state_dict
    from oracle import towers_ref as TR
    cfg = Phi3Config(vocab=512, hidden=384, layers=2, heads=4, kv_heads=4, mlp=512)       # head_dim 96 like Phi-3-mini
    sd = synth_state_dict(phi3_param_spec(cfg), seed=0)
    lens = [37, 300, 129, 64]
    steps = 6
    g = torch.Generator().manual_seed(11)
    rows = [(torch.randn(n, cfg.hidden, generator=g) * 0.5).to(torch.bfloat16) for n in lens]
    emb = torch.zeros(len(lens), max(lens), cfg.hidden)
    for b, r in enumerate(rows):
        emb[b, :lens[b]] = r.float()
    sd_r = {k: (v.to(torch.bfloat16).float() if v.dim() == 2 else v) for k, v in sd.items()}     # oracle on the same bf16 weights
    ref_tok, ref_logits = TR.phi3_greedy_decode(emb, lens, sd_r, cfg.layers, cfg.heads, cfg.kv_heads, steps, None, cfg.rms_eps, cfg.rope_theta)
    forced = [[ref_tok[b][i] for b in range(len(lens))] for i in range(steps)]
    saved = dict(D.BACKEND)
    try:
        D.enable_hip_kernels(["all"])
        dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
        T = sum(lens)
        x = torch.zeros(((T + 255) // 256 * 256, cfg.hidden), dtype=torch.bfloat16, device="cuda")
        x[:T] = torch.cat(rows).cuda()
        tok_f, logits_f = dec.generate_packed(x, lens, max_new_tokens=steps, forced=forced, return_logits=True)
        tok_free = dec.generate_packed(x, lens, max_new_tokens=steps)
        end = ref_tok[1][2]                                                    # a token sequence 1 emits at step 2
        tok_end = dec.generate_packed(x, lens, max_new_tokens=steps, end_id=end, forced=forced)
    finally:
        D.BACKEND.update(saved)
    assert tok_f == ref_tok
    got, ref = logits_f.cpu().numpy(), ref_logits.numpy()
    for i in range(steps):
        r = np.linalg.norm(got[i] - ref[i]) / np.linalg.norm(ref[i])
        assert r < 3e-2, (i, r)
    top2 = np.sort(ref, -1)[..., -2:]
    margin = (top2[..., 1] - top2[..., 0]) / np.linalg.norm(ref, axis=-1) * np.sqrt(ref.shape[-1])   # in units of the per-logit rms
    for b in range(len(lens)):
        for i in range(steps):
            if margin[i, b] > 0.25:
                assert tok_free[b][i] == ref_tok[b][i], (b, i, margin[i, b])
            else:
                break                                                          # after a near-tie the free-running paths may diverge
    for b in range(len(lens)):                                                 # every sequence is cut at ITS first `end` (inclusive)
        cut = ref_tok[b].index(end) + 1 if end in ref_tok[b] else steps
        assert tok_end[b] == ref_tok[b][:cut], (b, tok_end[b], ref_tok[b])
    assert len(tok_end[1]) == 3


def test_policy_forward_generates_text_with_kv_cache():
    """The reference's per-step call (VLN-POL:329 -> List[str]) end to end on the GPU: packed prefill + KV-cache decode,
    history update (VLN-POL:466-468), text -> action."""
    import dataclasses
    from dynam3d_amd.policy import Dynam3D_VLN, synth_policy_weights
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    cfg = dataclasses.replace(SMALL, clip_dtype=torch.float16, llava_dtype=torch.bfloat16)
    from dynam3d_amd import dense_ops as D
    net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0), device="cuda", batch_size=2, max_steps=4)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    ep = SyntheticEpisodes(2, seed=3, image_hw=224, depth_hw=224)
    before = [list(h) for h in net.feature_fields.history_actions]
    for _ in range(2):
        fr = ep.next()
        obs = dict(rgb=torch.from_numpy(fr.rgb), depth=torch.from_numpy(fr.depth))
        with D.allow_fallback():                                        # toy widths (SMALL): below the HIP kernels' tiles
            texts = net(obs, [INSTRUCTION_64] * 2, [p.tolist() for p in fr.positions], list(fr.headings), patch_segm=fr.patch_segm, max_new_tokens=5)
        assert len(texts) == 2 and all(isinstance(t, str) for t in texts)
        acts = net.convert_text_to_action(texts)
        assert len(acts) == 2
    after = net.feature_fields.history_actions
    assert all(len(a) == len(b) for a, b in zip(after, before)) and all(a[-1] == t + "\n" for a, t in zip(after, texts))


def test_bench_two_ranks_share_one_gpu():
    """The driver's N > 1 launch of bench.py (torch.distributed.run, one rank per GPU) exercised on a ONE-GPU box: both ranks on
    cuda:0 over gloo (D3D_SHARE_DEVICE0 / D3D_DIST_BACKEND test hooks; RCCL refuses duplicate devices).  Checks the barrier /
    max-over-ranks timing / rank-0 JSON line path, not performance."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, D3D_SHARE_DEVICE0="1", D3D_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--warm-steps", "1"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 8 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-2 * d["value"]   # whole-job aggregate
    assert d["ranks_seen"] == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1] and d["devices_seen"] == 1     # (the hook: both on cuda:0)
    assert all(r["ms_per_step"] <= d["ms_per_step"] * 1.001 for r in d["per_rank"])                              # the line reports the slowest rank


def test_bench_gpus_n_without_launcher_spawns_n_ranks_or_refuses():
    """`python bench.py --gpus 2` with no launcher: on a box with fewer GPUs it exits non-zero and prints NO result line; with the
    shared-device test hook it re-launches itself under torch.distributed.run and the line shows two ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--warm-steps", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "D3D_SHARE_DEVICE0", "D3D_DIST_BACKEND")}
    if torch.cuda.device_count() < 2:
        out = subprocess.run(base, env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "refusing" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run(base, env=dict(env, D3D_SHARE_DEVICE0="1", D3D_DIST_BACKEND="gloo"), cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_a_cuda_call_without_a_hip_kernel_raises_by_default():
    """Strict is the DEFAULT of dense_ops: a CUDA tensor whose shape / dtype has no hand-written kernel raises instead of running the
    PyTorch expression (hipBLASLt / SDPA); `allow_fallback()` is the explicit opt-out and is counted; CPU tensors always take the
    expression."""
    import pytest
    from dynam3d_amd import dense_ops as D
    assert D.STRICT
    D.enable_hip_kernels(["all"])
    x, w = torch.randn(40, 100, device="cuda", dtype=torch.bfloat16), torch.randn(72, 100, device="cuda", dtype=torch.bfloat16)   # N % 128, K % 64 both fail
    with pytest.raises(D.DenseFallbackError):
        D.linear(x, w, None)
    with pytest.raises(D.DenseFallbackError):
        D.linear(x.float(), w.float(), None)                                      # float32 operands: no 16-bit GEMM
    with pytest.raises(D.DenseFallbackError):
        D.rms_norm(torch.randn(4, 100, device="cuda", dtype=torch.bfloat16), torch.ones(100, device="cuda"), 1e-5)
    with pytest.raises(D.DenseFallbackError):
        D.attention_qkv(torch.randn(1, 8, 6, 32, device="cuda", dtype=torch.bfloat16), 2, True)            # head_dim 32
    D.reset_counts()
    with D.allow_fallback():
        y = D.linear(x, w, None)
    assert D.STRICT and D.counts()["fallback"] == {"linear": 1}
    assert torch.allclose(y.float(), (x.float() @ w.float().T), atol=0.5, rtol=5e-2)
    assert D.linear(x.cpu().float(), w.cpu().float(), None).shape == (40, 72)      # CPU tensors: the expression, never counted
    assert D.counts()["fallback"] == {"linear": 1}
