"""The WHOLE step at the FULL configuration, end to end, against golden g19 (tests/golden/gen_golden_full_step.py: the CPU oracle in
float32 and with the reference's 16-bit rounding points, ViT-L/14@336 x 2 + Phi-3-mini x 32, B = 2, memory steps 0-1, then 20 greedy
tokens).  Everything the piecewise tests cover separately -- towers, per-layer teacher forcing, 3D-token builder, prefix, splice --
composed once at full size: RGB-D in, logits / generated tokens out (VLN-POL:329-363, 430-463).

Criterion for the 16-bit logits = the noise band: g19 carries the distance between the reference-dtype evaluation and
float32; the HIP path must be within that band of the lowp oracle (hard) and within 1.05 x it of float32, choose the oracle's token wherever the oracle's own top-2 margin is
outside the band, and overlap its top-5.  Bookkeeping (prompt lengths, instance / zone counts) is exact."""
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLDEN_DIR

pytestmark = pytest.mark.gpu

# When is a token DECIDED?  The band (lowp vs float32, g19) is ~1.73e-2 relative L2 over the vocabulary = a per-logit noise of ~0.0173 rms units;
# the difference of two logits carries sqrt(2) of that (0.0245).  A top-2 margin above 5 of those sigmas cannot be flipped by 16-bit noise.
# (With seeded random weights the logits are nearly flat -- the oracle's own margins are 0.006 ... 0.23 rms -- so most, not all, positions qualify.)
MARGIN_RMS = 0.12


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_full_configuration_step_and_generation_vs_oracle_golden():
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    g = np.load(os.path.join(GOLDEN_DIR, "g19_full_step.npz"))
    B, steps = int(g["B"]), int(g["steps"])
    cfg = PolicyConfig()
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(32, max(8, os.cpu_count() or 8)))            # (restored below: 256 intra-op threads on the GPU box's host make the
    try:                                                                   #  small CPU oracles of the tests that follow crawl)
        sd = synth_policy_weights(cfg, int(g["weight_seed"]))              # CPU generator: the values the golden was computed with
    finally:
        torch.set_num_threads(threads_before)
    D.enable_hip_kernels(["all"])
    D.strict(True)
    D.reset_counts()
    try:
        net = Dynam3D_VLN(cfg, sd, device="cuda", batch_size=B, max_steps=steps + 1)
        del sd
        net.feature_fields.initialize_camera_setting(90.0, 90.0)
        ep = SyntheticEpisodes(B, seed=int(g["episode_seed"]), image_hw=224, depth_hw=224)
        instr = [INSTRUCTION_64] * B
        report = []
        for t in range(steps):
            fr = ep.next()
            obs = dict(rgb=torch.from_numpy(fr.rgb), depth=torch.from_numpy(fr.depth))
            pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
            if t < steps - 1:
                lo = net.forward_logits(obs, instr, pos, hd, patch_segm=fr.patch_segm).float().cpu().numpy()
            else:                                                           # last step: keep the packed prompt for the generation below
                x, lengths = net.build_inputs(obs, instr, pos, hd, patch_segm=fr.patch_segm, return_rows="packed")
                lo = net.llm.prefill_logits_packed(x, lengths).float().cpu().numpy()
            f32, lowp = g[f"logits_f32_{t}"], g[f"logits_lowp_{t}"]
            assert list(net.last_lengths) == g[f"lengths_{t}"].tolist()                                # exact bookkeeping: same merges, same prompt
            assert net.last_counts["Ni"] == g[f"ni_{t}"].tolist() and net.last_counts["Nz"] == g[f"nz_{t}"].tolist()
            band = _rel(lowp, f32)
            d32, d16 = _rel(lo, f32), _rel(lo, lowp)
            # hard: the HIP evaluation is at least as close to the reference-dtype (lowp) oracle as float32 is; its own distance from float32
            # is one more sample of the same 16-bit noise.  The 5 % slack is the measured spread of that sample: seven full-size measurements
            # at five operating points (g19 steps 0 / 1: 1.006 / 1.017 x band; bench B = 8 at memory steps 13 and 24 on the run's own state:
            # 0.996 / 0.996; golden g22 on three boxes: 1.005) -- rounds 4-6, DESIGN.md 5.3.  What catches a SMALL kernel error is not this
            # band but the float32 verification mode (tests/test_gpu_f32_mode.py: the same wiring at 1e-3, measured 1.2e-5).
            assert d16 <= band and d32 <= 1.05 * band, (t, d32, d16, band)
            srt = np.sort(f32, -1)
            margin = (srt[:, -1] - srt[:, -2]) / np.sqrt((f32.astype(np.float64) ** 2).mean(-1))
            top5 = []
            for b in range(B):
                if margin[b] > MARGIN_RMS:
                    assert lo[b].argmax() == f32[b].argmax(), (t, b, margin[b])
                top5.append(len(set(np.argsort(-lo[b])[:5].tolist()) & set(np.argsort(-f32[b])[:5].tolist())))
            assert min(top5) >= 3, top5
            report.append(f"step {t}: S {list(net.last_lengths)}: HIP vs float32 {d32:.2e}, vs lowp {d16:.2e}, band {band:.2e}; "
                          f"argmax HIP {lo.argmax(-1).tolist()} f32 {f32.argmax(-1).tolist()} lowp {lowp.argmax(-1).tolist()} (margins {np.round(margin, 3).tolist()}); top-5 overlap {top5}")
        c = D.counts()
        assert not c["fallback"], c
        # ---- 20 greedy tokens on the last step's prompts (KV-cache decode) against the oracle's greedy continuation ----
        ref_tok = g["gen_tokens"]                                            # (B, T)
        T = ref_tok.shape[1]
        top_ids, top_val, rms = g["gen_top8_ids"], g["gen_top8_logits"], g["gen_logit_rms"]       # (T, B, 8), (T, B, 8), (T, B)
        forced = [[int(ref_tok[b][i]) for b in range(B)] for i in range(T)]
        tok_f, logits_f = net.llm.generate_packed(x, lengths, max_new_tokens=T, forced=forced, return_logits=True)
        tok_free = net.llm.generate_packed(x, lengths, max_new_tokens=T)
        lf = logits_f.float().cpu().numpy()                                  # teacher-forced: position i saw the oracle's tokens 0..i-1
        agree = total = 0
        for i in range(T):
            for b in range(B):
                m = (top_val[i, b, 0] - top_val[i, b, 1]) / rms[i, b]
                got8 = lf[i, b][top_ids[i, b]]
                assert np.linalg.norm(got8 - top_val[i, b]) / np.linalg.norm(top_val[i, b]) < 6e-2, (i, b)   # the oracle's top-8 logits, 16-bit band
                if m > MARGIN_RMS:
                    total += 1
                    agree += int(lf[i, b].argmax() == ref_tok[b][i])
        assert agree == total, (agree, total)
        same_prefix = []
        for b in range(B):                                                   # free running: identical until the first near-tie of the oracle
            n = 0
            while n < T and tok_free[b][n] == int(ref_tok[b][n]):
                n += 1
            first_tie = next((i for i in range(T) if (top_val[i, b, 0] - top_val[i, b, 1]) / rms[i, b] <= MARGIN_RMS), T)
            assert n >= first_tie, (b, n, first_tie)
            same_prefix.append((n, first_tie))
        report.append(f"generation: teacher-forced argmax == oracle token at {agree}/{total} decided positions (of {T * B}); free-running tokens equal the "
                      f"oracle's for (n, first near-tie) = {same_prefix} of {T}")
    finally:
        D.strict(True)                                              # (the default)
    print("\n".join(report))
    out = os.path.join(os.path.dirname(GOLDEN_DIR), "..", "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "full_parity_report.txt"), "w").write("\n".join(report) + "\n")
