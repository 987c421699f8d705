"""CPU: the `lowp` mode of oracle/towers_ref.py (float32 arithmetic inside a module, a 16-bit store at every module output)
against the REAL modules run in the reference's dtypes (g14: the reference's fp16 `VisionTransformer`, transformers'
CLIPVisionModel / Phi3ForCausalLM in bf16; tests/golden/gen_golden_lowp.py).

Two independent 16-bit evaluations of one network do not agree to better than their common distance from float32 (each store
rounds a slightly different value), so the pin is statistical: the restatement must sit as close to the real 16-bit run as that
run sits to float32 (`band`), and as close to float32 as the real run does.  Module by module (RMSNorm, SwiGLU, rotary,
Linear) the restatement was checked bit-for-bit against the HF modules when it was written."""
import numpy as np
import torch

from oracle import towers_ref as TR
from tests.golden_io import load
from dynam3d_amd.towers import Phi3Config, VitConfig, clip_param_spec, llava_vision_param_spec, phi3_param_spec
from dynam3d_amd.weights import synth_state_dict

VIT = VitConfig(image=56, patch=14, width=128, layers=3, heads=4, mlp=512, out_dim=96, proj_dim=192)
PHI = Phi3Config(vocab=512, hidden=192, layers=3, heads=6, kv_heads=6, mlp=384)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _check(name, emu, real16, real32):
    band = rel(real16, real32)
    d16, d32 = rel(emu, real16), rel(emu, real32)
    print(f"{name}: restatement vs real 16-bit {d16:.2e}, vs float32 {d32:.2e}; real 16-bit vs float32 {band:.2e}")
    assert d16 < 1.25 * band and d32 < 1.25 * band, (name, d16, d32, band)


def test_lowp_restatement_matches_real_low_precision_modules():
    g = load("g14_lowp_small.npz")
    px = TR.preprocess_rgb(g["rgb"], VIT.image)
    with torch.no_grad():
        sd = synth_state_dict(clip_param_spec(VIT), seed=0)
        _, p = TR.clip_vit_forward(px, sd, VIT.layers, VIT.heads, lowp=torch.float16)
        _check("clip fp16", p.numpy(), g["clip_f16"], g["clip_f32"])
        sd = synth_state_dict(llava_vision_param_spec(VIT), seed=0)
        f = TR.llava_image_features(px, sd, VIT.layers, VIT.heads, lowp=torch.bfloat16)
        _check("llava bf16", f.numpy(), g["llava_bf16"], g["llava_f32"])
        sd = synth_state_dict(phi3_param_spec(PHI), seed=0)
        lo = TR.phi3_prefill_logits(torch.from_numpy(g["phi_embeds"]), g["phi_lengths"].tolist(), sd, PHI.layers, PHI.heads, PHI.kv_heads,
                                    PHI.rms_eps, PHI.rope_theta, lowp=torch.bfloat16)
        _check("phi3 bf16", lo.numpy(), g["phi_bf16"], g["phi_f32"])
        # float32 mode is untouched by the low-precision hooks
        lo32 = TR.phi3_prefill_logits(torch.from_numpy(g["phi_embeds"]), g["phi_lengths"].tolist(), sd, PHI.layers, PHI.heads, PHI.kv_heads,
                                      PHI.rms_eps, PHI.rope_theta)
        assert rel(lo32.numpy(), g["phi_f32"]) < 1e-5
