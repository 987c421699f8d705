"""GPU: the three dense towers ON THE HIP KERNELS AT THE BENCHMARK'S REAL SHAPES, strict mode (any PyTorch fallback raises),
against goldens produced by the reference's own modules in the reference's own dtypes (tests/golden/gen_golden_lowp.py):

  * ClipVisionTower  ViT-L/14@336 fp16, 8 frames (M = 4616 rows)  vs the reference `VisionTransformer` after `convert_weights`  (g11)
  * LlavaVisionTower ViT-L/14@336 bf16 + projector, 8 frames       vs transformers' CLIPVisionModel in bf16                      (g12)
  * Phi3Decoder      hidden 3072 / 32 heads / MLP 8192 / vocab 32064, 2 layers, 8 packed ragged prompts, 6850 rows
                     (tiles 257 / 258 / tail-128, paired-causal attention, skinny lm_head)     vs Phi3ForCausalLM in bf16       (g13)

Tolerances.  `*_lowp` = distance to the golden computed in the reference's dtype (what north_star's "within 1e-3" can mean
for a 16-bit pipeline); `*_f32` = distance to the same module evaluated in float32 (the sanity band: a 16-bit pipeline sits
there by construction -- the goldens record how far the REFERENCE's own 16-bit run is from its float32 run, and the HIP path
must not be further away than that by more than 25 %)."""
import numpy as np
import pytest
import torch

from tests.golden_io import load

pytestmark = pytest.mark.gpu

# Two 16-bit evaluations of one network -- the reference's own run, the lowp restatement, the HIP path -- agree no better than each
# of them agrees with float32: after a few layers every 16-bit store rounds a slightly different value (measured: the reference's
# fp16 ViT-L sits 1.39e-3 from its float32 run, HF's bf16 Phi-3 9e-3).  BAND is the slack on that distance.
BAND = 1.25


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture()
def strict_hip():
    from dynam3d_amd import dense_ops as D
    saved, was = dict(D.BACKEND), D.STRICT
    D.enable_hip_kernels(["all"])
    D.strict(True)
    D.reset_counts()
    yield D
    D.strict(was)
    D.BACKEND.update(saved)


def _frames(seed, n, hw=224):
    return np.random.default_rng(seed).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)


def test_clip_vit_l14_336_fp16_on_hip_vs_reference_fp16(strict_hip):
    from dynam3d_amd.towers import ClipVisionTower, VitConfig, clip_param_spec, preprocess_rgb
    from dynam3d_amd.weights import synth_state_dict
    g = load("g11_clip_fp16_full.npz")
    cfg = VitConfig()
    sd = synth_state_dict(clip_param_spec(cfg), seed=0)
    tower = ClipVisionTower(sd, cfg, torch.float16, "cuda")
    rgb = torch.from_numpy(_frames(int(g["rgb_seed"]), int(g["n_img"]))).cuda()
    cls, patch = tower.forward(preprocess_rgb(rgb, cfg.image))
    torch.cuda.synchronize()
    c = strict_hip.counts()
    assert not c["fallback"], c
    assert c["hip"]["linear"] == 1 + 4 * cfg.layers + 1 and c["hip"]["attention"] == cfg.layers and c["hip"]["vit_embed"] == 1, c
    cls, patch = cls.float().cpu(), patch.float().cpu()
    rows, rowsum = patch[:, ::72].numpy(), patch.double().sum(-1).numpy()
    r_lowp = max(rel(rows, g["f16_rows"]), rel(cls.numpy(), g["f16_cls"]))
    r_f32 = rel(rows, g["f32_rows"])
    ref_band = rel(g["f16_rows"], g["f32_rows"])                     # the reference's own fp16-vs-float32 distance
    from oracle import towers_ref as TR
    with torch.no_grad():                                            # the restatement in lowp mode on the same frames (2 of the 8: CPU seconds)
        _, emu = TR.clip_vit_forward(TR.preprocess_rgb(_frames(int(g["rgb_seed"]), int(g["n_img"]))[:2], cfg.image), sd, cfg.layers, cfg.heads,
                                     lowp=torch.float16)
    r_emu = rel(patch[:2].numpy(), emu.numpy())
    print(f"CLIP fp16: vs reference fp16 {r_lowp:.2e}; vs float32 {r_f32:.2e} (reference fp16 vs float32: {ref_band:.2e}); vs lowp restatement {r_emu:.2e}")
    assert r_lowp < BAND * ref_band and r_emu < BAND * ref_band, (r_lowp, r_emu, ref_band)
    assert r_f32 < BAND * ref_band, (r_f32, ref_band)
    assert np.abs(rowsum - g["f16_rowsum"]).max() < 2e-2 * np.abs(g["f16_rowsum"]).max() + 0.05


def test_llava_vision_tower_bf16_on_hip_vs_hf_bf16(strict_hip):
    from dynam3d_amd.towers import LlavaVisionTower, VitConfig, llava_vision_param_spec, preprocess_rgb
    from dynam3d_amd.weights import synth_state_dict
    g = load("g12_llava_bf16_full.npz")
    cfg = VitConfig()
    sd = synth_state_dict(llava_vision_param_spec(cfg), seed=0)
    tower = LlavaVisionTower(sd, cfg, torch.bfloat16, "cuda")
    rgb = torch.from_numpy(_frames(int(g["rgb_seed"]), int(g["n_img"]))).cuda()
    f = tower.forward(preprocess_rgb(rgb, cfg.image))
    torch.cuda.synchronize()
    c = strict_hip.counts()
    assert not c["fallback"], c
    assert c["hip"]["linear"] == 1 + 4 * (cfg.layers - 1) + 2 and c["hip"]["attention"] == cfg.layers - 1, c
    f = f.float().cpu()
    rows = f[:, ::72].numpy()
    r_lowp, r_f32 = rel(rows, g["bf16_rows"]), rel(rows, g["f32_rows"])
    ref_band = rel(g["bf16_rows"], g["f32_rows"])
    from oracle import towers_ref as TR
    with torch.no_grad():
        emu = TR.llava_image_features(TR.preprocess_rgb(_frames(int(g["rgb_seed"]), int(g["n_img"]))[:2], cfg.image), sd, cfg.layers, cfg.heads,
                                      lowp=torch.bfloat16)
    r_emu = rel(f[:2].numpy(), emu.numpy())
    print(f"llava ViT bf16: vs HF bf16 {r_lowp:.2e}; vs float32 {r_f32:.2e} (HF bf16 vs float32: {ref_band:.2e}); vs lowp restatement {r_emu:.2e}")
    assert r_f32 < BAND * ref_band, (r_f32, ref_band)
    assert r_lowp < BAND * ref_band and r_emu < BAND * ref_band, (r_lowp, r_emu, ref_band)


def test_phi3_full_width_packed_prefill_on_hip_vs_hf_bf16(strict_hip):
    from dynam3d_amd.towers import Phi3Config, Phi3Decoder, phi3_param_spec
    from dynam3d_amd.weights import synth_state_dict
    g = load("g13_phi3_bf16_fullwidth.npz")
    cfg = Phi3Config(layers=2)
    lens = [int(n) for n in g["lengths"]]
    sd = synth_state_dict(phi3_param_spec(cfg), seed=0)
    dec = Phi3Decoder(sd, cfg, torch.bfloat16, "cuda")
    assert dec.interleave_gu and dec.packed_ok()
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    rows = [(torch.randn(n, cfg.hidden, generator=gen) * 0.5).to(torch.bfloat16) for n in lens]
    T = sum(lens)
    x = torch.zeros(((T + 255) // 256 * 256, cfg.hidden), dtype=torch.bfloat16, device="cuda")
    x[:T] = torch.cat(rows).cuda()
    lo = dec.prefill_logits_packed(x, lens)
    torch.cuda.synchronize()
    c = strict_hip.counts()
    assert not c["fallback"], c
    assert c["hip"]["linear"] == 3 * cfg.layers + 1 and c["hip"]["linear_swiglu"] == cfg.layers and c["hip"]["attention"] == cfg.layers, c
    assert dec.last_packed_rows == 6912 and T == 6850
    lo = lo.cpu().numpy()
    ref16, ref32 = g["bf16_logits"], g["f32_logits"]
    r_lowp = [rel(lo[b], ref16[b]) for b in range(len(lens))]
    r_f32 = [rel(lo[b], ref32[b]) for b in range(len(lens))]
    band = [rel(ref16[b], ref32[b]) for b in range(len(lens))]
    from oracle import towers_ref as TR
    r_emu = []
    with torch.no_grad():
        for b in (0, 2):                                             # the shortest-ish and the longest prompt through the restatement
            e = TR.phi3_prefill_logits(rows[b].float()[None], [lens[b]], sd, cfg.layers, cfg.heads, cfg.kv_heads, cfg.rms_eps, cfg.rope_theta,
                                       lowp=torch.bfloat16)[0].numpy()
            r_emu.append(rel(lo[b], e))
    print("Phi-3 bf16 full width: vs HF bf16", np.round(r_lowp, 4), "vs float32", np.round(r_f32, 4), "HF bf16 vs float32", np.round(band, 4),
          "vs lowp restatement", np.round(r_emu, 5))
    assert max(r_emu) < BAND * max(band), (r_emu, band)
    assert max(r_f32) < BAND * max(band), (r_f32, band)
    assert max(r_lowp) < BAND * max(band), (r_lowp, band)
    # the first generated token: equal to HF-bf16's wherever HF-bf16's own top-2 margin is not inside the bf16 noise
    top2 = np.sort(ref16, -1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 0.05 * np.abs(ref16).max(-1)
    assert np.array_equal(lo.argmax(-1)[safe], ref16.argmax(-1)[safe])
