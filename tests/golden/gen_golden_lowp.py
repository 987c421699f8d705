"""Golden vectors of the dense towers IN THE REFERENCE'S OWN DTYPES, at the benchmark's real shapes.  Container-only.

The reference runs its OpenAI-CLIP encoder in fp16 (`clip.load(..., device=cuda)`, resnet_encoders.py:260) and llava /
Phi-3 in bf16 (`torch_dtype=torch.bfloat16`, VLN-POL:125).  North_star's tolerance -- 1e-3 relative on token features and
logits -- is only meaningful against THAT arithmetic (fp32 accumulation, a 16-bit round at every module output), not
against a float32 restatement: a bf16 pipeline sits ~1e-2 from float32 by construction.  So these goldens are produced by
the real modules in the real dtypes, on the CPU of the build container:

  g11 : the reference's `VisionTransformer` after the reference's `convert_weights` (fp16 Linear/Conv/MHA/proj, fp32
        LayerNorm + embeddings; encoders/clip/model.py:153-238, 373-395): ViT-L/14@336 on 8 seeded 224x224 frames
        (M = 8 x 577 = 4616 rows: the benchmark's GEMM shapes) + the float32 run of the same module.
  g12 : installed transformers' CLIPVisionModel in bf16 (llava's vision tower), hidden_states[-2] without CLS, + the 2-layer
        GELU projector in bf16 (VLN-POL:448-452), ViT-L/14@336 x 8 frames, + float32.
  g13 : installed transformers' Phi3ForCausalLM in bf16 at Phi-3-mini WIDTH (hidden 3072, 32 heads, MLP 8192, vocab 32064),
        2 layers, 8 ragged prompts with the benchmark's lengths (sum 6850 rows) -> last-position logits, + float32.
Weights are name-keyed synthetic tensors (dynam3d_amd/weights.py, CPU generator), inputs are seeded: only outputs are stored.
transformers version is recorded in every file (the reference pins 4.46.0; 5.x is what this image has)."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from oracle import towers_ref as TR  # noqa: E402
from dynam3d_amd.towers import Phi3Config, VitConfig, clip_param_spec, llava_vision_param_spec, phi3_param_spec  # noqa: E402
from dynam3d_amd.weights import synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PHI_LENS = [830, 835, 1108, 792, 969, 764, 779, 773]          # the S_tokens of BENCH_r01.json (sum 6850)
PHI2 = Phi3Config(layers=2)                                    # full width, 2 of the 32 identical layers
N_IMG = 8


def frames(seed, n=N_IMG, hw=224):
    return np.random.default_rng(seed).integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)


def phi_inputs(seed=130):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, PHI2.hidden, generator=g) * 0.5).to(torch.bfloat16) for n in PHI_LENS]


def _sample(patch):
    """(B,576,D) -> the rows every 72nd patch + the per-row sums (a checksum over every row)."""
    p = patch.float()
    return p[:, ::72].numpy(), p.double().sum(-1).numpy()


def g11():
    import transformers
    clipm = rh.load_ref_clip()
    cfg = VitConfig()
    sd = synth_state_dict(clip_param_spec(cfg), seed=0)
    vt = clipm.VisionTransformer(cfg.image, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim).eval()
    vt.load_state_dict({k[len("visual."):]: v for k, v in sd.items()}, strict=True)
    rgb_seed = 110
    px = TR.preprocess_rgb(frames(rgb_seed), cfg.image)
    t0 = time.time()
    with torch.no_grad():
        c32, p32 = vt(px)
    clipm.convert_weights(vt)                                  # the reference's fp16 conversion (model.py:373-395)
    with torch.no_grad():
        c16, p16 = vt(px.half())                               # CLIPEncoder feeds the model's dtype (resnet_encoders.py:273-284)
    out = dict(rgb_seed=rgb_seed, n_img=N_IMG, torch=torch.__version__, transformers=transformers.__version__)
    out["f32_cls"], (out["f32_rows"], out["f32_rowsum"]) = c32.numpy(), _sample(p32)
    out["f16_cls"], (out["f16_rows"], out["f16_rowsum"]) = c16.float().numpy(), _sample(p16)
    np.savez_compressed(os.path.join(OUT, "g11_clip_fp16_full.npz"), **out)
    print("g11 %.1fs  fp16 vs f32 rel %.2e" % (time.time() - t0, float((p16.float() - p32).norm() / p32.norm())))


def g12():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = VitConfig()
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=c.width, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                                          image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu", layer_norm_eps=1e-5)).eval()
    sd = synth_state_dict(llava_vision_param_spec(c), seed=0)
    own = {k[len("vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower.")}
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):      # transformers>=5 flattened the wrapper
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    missing, unexpected = hf.load_state_dict(own, strict=False)
    assert not unexpected and all("post_layernorm" in m or "position_ids" in m for m in missing), (missing, unexpected)
    rgb_seed = 120
    px = TR.preprocess_rgb(frames(rgb_seed), c.image)
    w1, b1 = sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"]
    w2, b2 = sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"]
    t0 = time.time()
    out = dict(rgb_seed=rgb_seed, n_img=N_IMG, torch=torch.__version__, transformers=transformers.__version__)
    with torch.no_grad():
        hs = hf(pixel_values=px, output_hidden_states=True).hidden_states[-2][:, 1:]
        f32 = F.linear(F.gelu(F.linear(hs, w1, b1)), w2, b2)
        hf = hf.to(torch.bfloat16)                              # llava is loaded with torch_dtype=torch.bfloat16 (VLN-POL:125)
        bf = torch.bfloat16
        hs16 = hf(pixel_values=px.to(bf), output_hidden_states=True).hidden_states[-2][:, 1:]
        f16 = F.linear(F.gelu(F.linear(hs16, w1.to(bf), b1.to(bf))), w2.to(bf), b2.to(bf))
    out["f32_rows"], out["f32_rowsum"] = _sample(f32)
    out["bf16_rows"], out["bf16_rowsum"] = _sample(f16)
    np.savez_compressed(os.path.join(OUT, "g12_llava_bf16_full.npz"), **out)
    print("g12 %.1fs  bf16 vs f32 rel %.2e" % (time.time() - t0, float((f16.float() - f32).norm() / f32.norm())))


def _to_bf16_like_4_46(hf):
    """`model.to(bfloat16)` of transformers 5.x also casts the rotary `inv_freq` BUFFER to bf16 (angles then come from rounded
    frequencies).  The reference's pinned transformers 4.46.0 recomputes `inv_freq` in float32 inside every
    `Phi3RotaryEmbedding.forward`, so its bf16 model rotates with float32 frequencies and bf16-cast cos / sin: restore that."""
    inv32 = hf.model.rotary_emb.inv_freq.clone().float()
    hf = hf.to(torch.bfloat16)
    hf.model.rotary_emb.inv_freq = inv32
    if hasattr(hf.model.rotary_emb, "original_inv_freq"):
        hf.model.rotary_emb.original_inv_freq = inv32
    return hf


def g13():
    import transformers
    from transformers import Phi3Config as HFPhi3Config, Phi3ForCausalLM
    c = PHI2
    hf = Phi3ForCausalLM(HFPhi3Config(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                                      num_attention_heads=c.heads, num_key_value_heads=c.kv_heads, rms_norm_eps=c.rms_eps, rope_theta=c.rope_theta,
                                      max_position_embeddings=c.max_pos, original_max_position_embeddings=c.max_pos, pad_token_id=0,
                                      tie_word_embeddings=False)).eval()
    sd = synth_state_dict(phi3_param_spec(c), seed=0)
    own = {k[len("language_model."):]: v for k, v in sd.items()}
    missing, unexpected = hf.load_state_dict(own, strict=False)
    assert not unexpected and not [m for m in missing if "rotary" not in m and "inv_freq" not in m], (missing, unexpected)
    rows = phi_inputs()
    t0 = time.time()
    lo32, lo16 = [], []
    with torch.no_grad():
        for r in rows:
            lo32.append(hf(inputs_embeds=r.float()[None]).logits[0, -1].numpy())
        hf = _to_bf16_like_4_46(hf)
        for r in rows:
            lo16.append(hf(inputs_embeds=r[None]).logits[0, -1].float().numpy())
    lo32, lo16 = np.stack(lo32), np.stack(lo16)
    np.savez_compressed(os.path.join(OUT, "g13_phi3_bf16_fullwidth.npz"), lengths=np.array(PHI_LENS), input_seed=130, f32_logits=lo32, bf16_logits=lo16,
                        attn_implementation=str(getattr(hf.config, "_attn_implementation", "?")), torch=torch.__version__,
                        transformers=transformers.__version__)
    rel = np.linalg.norm(lo16 - lo32, axis=-1) / np.linalg.norm(lo32, axis=-1)
    print("g13 %.1fs  bf16 vs f32 rel per row" % (time.time() - t0), np.round(rel, 4), "argmax equal", (lo16.argmax(-1) == lo32.argmax(-1)).tolist())


def g14():
    """Small configurations of the three towers in the reference's dtypes (seconds on the CPU): pins oracle/towers_ref.py's
    `lowp` mode in the CPU suite."""
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, Phi3Config as HFPhi3Config, Phi3ForCausalLM
    out = dict(torch=torch.__version__, transformers=transformers.__version__)
    vit = VitConfig(image=56, patch=14, width=128, layers=3, heads=4, mlp=512, out_dim=96, proj_dim=192)
    rgb = frames(140, 2, 40)
    out["rgb"] = rgb
    px = TR.preprocess_rgb(rgb, vit.image)
    # OpenAI CLIP, fp16
    clipm = rh.load_ref_clip()
    sd = synth_state_dict(clip_param_spec(vit), seed=0)
    vt = clipm.VisionTransformer(vit.image, vit.patch, vit.width, vit.layers, vit.heads, vit.out_dim).eval()
    vt.load_state_dict({k[len("visual."):]: v for k, v in sd.items()}, strict=True)
    with torch.no_grad():
        _, p32 = vt(px)
        clipm.convert_weights(vt)
        _, p16 = vt(px.half())
    out["clip_f32"], out["clip_f16"] = p32.numpy(), p16.float().numpy()
    # HF CLIP vision + projector, bf16
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=vit.width, intermediate_size=vit.mlp, num_hidden_layers=vit.layers, num_attention_heads=vit.heads,
                                          image_size=vit.image, patch_size=vit.patch, hidden_act="quick_gelu", layer_norm_eps=1e-5)).eval()
    sd = synth_state_dict(llava_vision_param_spec(vit), seed=0)
    own = {k[len("vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower.")}
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    hf.load_state_dict(own, strict=False)
    w1, b1 = sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"]
    w2, b2 = sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"]
    bf = torch.bfloat16
    with torch.no_grad():
        hs = hf(pixel_values=px, output_hidden_states=True).hidden_states[-2][:, 1:]
        out["llava_f32"] = F.linear(F.gelu(F.linear(hs, w1, b1)), w2, b2).numpy()
        hf = hf.to(bf)
        hs = hf(pixel_values=px.to(bf), output_hidden_states=True).hidden_states[-2][:, 1:]
        out["llava_bf16"] = F.linear(F.gelu(F.linear(hs, w1.to(bf), b1.to(bf))), w2.to(bf), b2.to(bf)).float().numpy()
    # HF Phi-3, bf16
    c = Phi3Config(vocab=512, hidden=192, layers=3, heads=6, kv_heads=6, mlp=384)
    hf = Phi3ForCausalLM(HFPhi3Config(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                                      num_attention_heads=c.heads, num_key_value_heads=c.kv_heads, rms_norm_eps=c.rms_eps, rope_theta=c.rope_theta,
                                      max_position_embeddings=c.max_pos, original_max_position_embeddings=c.max_pos, pad_token_id=0,
                                      tie_word_embeddings=False)).eval()
    sd = synth_state_dict(phi3_param_spec(c), seed=0)
    hf.load_state_dict({k[len("language_model."):]: v for k, v in sd.items()}, strict=False)
    g = torch.Generator().manual_seed(141)
    lens = [37, 50, 23]
    emb = (torch.randn(3, 50, c.hidden, generator=g) * 0.5).to(bf)
    with torch.no_grad():
        lo32 = np.stack([hf(inputs_embeds=emb[b:b + 1, :L].float()).logits[0, -1].numpy() for b, L in enumerate(lens)])
        hf = _to_bf16_like_4_46(hf)
        lo16 = np.stack([hf(inputs_embeds=emb[b:b + 1, :L]).logits[0, -1].float().numpy() for b, L in enumerate(lens)])
    out.update(phi_embeds=emb.float().numpy(), phi_lengths=np.array(lens), phi_f32=lo32, phi_bf16=lo16)
    np.savez_compressed(os.path.join(OUT, "g14_lowp_small.npz"), **out)
    r = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    print("g14 clip fp16 vs f32 %.2e | llava bf16 vs f32 %.2e | phi3 bf16 vs f32 %.2e" %
          (r(out["clip_f16"], out["clip_f32"]), r(out["llava_bf16"], out["llava_f32"]), r(lo16, lo32)))


if __name__ == "__main__":
    torch.set_num_threads(8)
    for w in (sys.argv[1:] or ["g11", "g12", "g13", "g14"]):
        globals()[w]()
