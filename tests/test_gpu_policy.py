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
    net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0), device="cuda", batch_size=3, max_steps=8)
    sums, n = run_rollout(net, 3, 5, seed=4, stop_token_mod=3)
    assert n == 3 and net.feature_fields.batch_size == 0
    res = DD.gather_metrics(sums, n, device="cuda")
    assert res["episodes"] == 3.0 and 1.0 <= res["steps_taken"] <= 5.0
