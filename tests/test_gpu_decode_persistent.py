"""The persistent decode kernel (csrc/decode_kernels.hip: one cooperative launch per generated token, grid barriers between the
phases) at llava-phi-3-mini's real widths (hidden 3072, 32 heads x 96, mlp 8192, vocab 32064; 2 layers to keep the oracle fast):
against the launch-per-op path (`D3D_DECODE_PERSISTENT=0`, the kernels the other decode tests pin) and against the float32 oracle
that re-runs the whole prefix for every token (oracle/towers_ref.py::phi3_greedy_decode)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dec, x, lens, steps, forced, persistent):
    old = os.environ.get("D3D_DECODE_PERSISTENT")
    os.environ["D3D_DECODE_PERSISTENT"] = "1" if persistent else "0"
    try:
        return dec.generate_packed(x, lens, max_new_tokens=steps, forced=forced, return_logits=True)
    finally:
        if old is None:
            os.environ.pop("D3D_DECODE_PERSISTENT", None)
        else:
            os.environ["D3D_DECODE_PERSISTENT"] = old


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_persistent_decode_matches_launch_per_op_and_oracle(dt):
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
    from dynam3d_amd.weights import synth_state_dict
    from oracle import towers_ref as TR
    cfg = Phi3Config(vocab=32064, hidden=3072, layers=2, heads=32, kv_heads=32, mlp=8192)
    sd = synth_state_dict(phi3_param_spec(cfg), seed=3)
    lens = [37, 211, 129, 64, 5, 90, 300, 17]                                # 8 ragged prompts
    steps = 5
    g = torch.Generator().manual_seed(5)
    rows = [(torch.randn(n, cfg.hidden, generator=g) * 0.5).to(dt) for n in lens]
    emb = torch.zeros(len(lens), max(lens), cfg.hidden)
    for b, r in enumerate(rows):
        emb[b, :lens[b]] = r.float()
    sd_r = {k: (v.to(dt).float() if v.dim() == 2 else v) for k, v in sd.items()}
    ref_tok, ref_logits = TR.phi3_greedy_decode(emb, lens, sd_r, cfg.layers, cfg.heads, cfg.kv_heads, steps, None, cfg.rms_eps, cfg.rope_theta)
    forced = [[ref_tok[b][i] for b in range(len(lens))] for i in range(steps)]
    saved = dict(D.BACKEND)
    try:
        D.enable_hip_kernels(["all"])
        dec = Phi3Decoder(sd, cfg, dt, "cuda")
        T = sum(lens)
        x = torch.zeros(((T + 255) // 256 * 256, cfg.hidden), dtype=dt, device="cuda")
        x[:T] = torch.cat(rows).cuda()
        tok_p, log_p = _run(dec, x, lens, steps, forced, True)
        tok_q, log_q = _run(dec, x, lens, steps, forced, False)
        tok_p2, log_p2 = _run(dec, x, lens, steps, forced, True)
    finally:
        D.BACKEND.update(saved)
    assert torch.equal(log_p, log_p2)                                        # deterministic from launch to launch
    got_p, got_q, ref = log_p.cpu().numpy(), log_q.cpu().numpy(), ref_logits.numpy()
    for i in range(steps):
        # the two device paths differ only in the fp32 summation order of the K slices (8 waves here, 4-16 there)
        d = np.linalg.norm(got_p[i] - got_q[i]) / np.linalg.norm(got_q[i])
        assert d < 6e-3, (i, d)
        r = np.linalg.norm(got_p[i] - ref[i]) / np.linalg.norm(ref[i])
        assert r < 3e-2, (i, r)
    assert tok_p == ref_tok and tok_q == ref_tok


def test_persistent_decode_fewer_rows():
    """3 sequences (rows < 8: the MFMA's unused activation rows are duplicates that are never stored) and free-running tokens."""
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
    from dynam3d_amd.weights import synth_state_dict
    cfg = Phi3Config(vocab=32064, hidden=3072, layers=2, heads=32, kv_heads=32, mlp=8192)
    sd = synth_state_dict(phi3_param_spec(cfg), seed=4)
    lens = [50, 7, 131]
    g = torch.Generator().manual_seed(6)
    rows = [(torch.randn(n, cfg.hidden, generator=g) * 0.5).to(torch.bfloat16) for n in lens]
    saved = dict(D.BACKEND)
    try:
        D.enable_hip_kernels(["all"])
        dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
        x = torch.zeros((256, cfg.hidden), dtype=torch.bfloat16, device="cuda")
        x[:sum(lens)] = torch.cat(rows).cuda()
        tok_p, log_p = _run(dec, x, lens, 4, None, True)
        tok_q, log_q = _run(dec, x, lens, 4, None, False)
    finally:
        D.BACKEND.update(saved)
    a, b = log_p.cpu().numpy(), log_q.cpu().numpy()
    assert np.isfinite(a).all()
    if tok_p == tok_q:
        for i in range(4):
            assert np.linalg.norm(a[i] - b[i]) / np.linalg.norm(b[i]) < 6e-3
    else:                                                                     # a near-tie may send the free-running paths apart: first token must agree
        assert np.linalg.norm(a[0] - b[0]) / np.linalg.norm(b[0]) < 6e-3
