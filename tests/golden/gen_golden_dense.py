"""Golden vectors for the dense towers.  Container-only.
  g5  : the REFERENCE's own OpenAI-CLIP `VisionTransformer` (encoders/clip/model.py) on a seeded small config
        + the full ViT-L/14@336 on one 224x224 image (checksum rows), weights regenerated from seed.
  g8  : installed transformers' CLIPVisionModel (llava vision tower arithmetic; `transformers==4.46.0` is the
        reference's un-vendored dependency) hidden_states[-2] on a seeded small config.
  g9  : installed transformers' Phi3ForCausalLM prefill logits on a seeded small config.
Weights are name-keyed synthetic tensors (dynam3d_amd/weights.py), so only outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from oracle import towers_ref as TR  # noqa: E402
from dynam3d_amd.towers import Phi3Config, VitConfig, clip_param_spec, llava_vision_param_spec, phi3_param_spec  # noqa: E402
from dynam3d_amd.weights import synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SMALL_VIT = VitConfig(image=56, patch=14, width=128, layers=3, heads=4, mlp=512, out_dim=96, proj_dim=192)
SMALL_PHI = Phi3Config(vocab=512, hidden=192, layers=3, heads=6, kv_heads=6, mlp=384)


def g5():
    clipm = rh.load_ref_clip()
    out = {}
    for tag, cfg, hw in (("small", SMALL_VIT, 40), ("full", VitConfig(), 224)):
        sd = synth_state_dict(clip_param_spec(cfg), seed=0)
        vt = clipm.VisionTransformer(cfg.image, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim).eval()
        vt.load_state_dict({k[len("visual."):]: v for k, v in sd.items()}, strict=True)
        rng = np.random.default_rng(50)
        rgb = rng.integers(0, 256, (1 if tag == "full" else 2, hw, hw, 3), dtype=np.uint8)
        px = TR.preprocess_rgb(rgb, cfg.image)
        with torch.no_grad():
            cls, patch = vt(px)
        out[tag + "_rgb"] = rgb
        if tag == "small":
            out[tag + "_cls"], out[tag + "_patch"] = cls.numpy(), patch.numpy()
        else:
            out[tag + "_cls"] = cls.numpy()
            out[tag + "_patch_rows"] = patch[0, ::72].numpy()          # 8 of the 576 rows
            out[tag + "_patch_rowsum"] = patch[0].double().sum(-1).numpy()
        print("g5", tag, float(patch.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "g5_clip_vit.npz"), **out)


def g8():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    c = SMALL_VIT
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=c.width, intermediate_size=c.mlp, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                                          image_size=c.image, patch_size=c.patch, hidden_act="quick_gelu", layer_norm_eps=1e-5)).eval()
    sd = synth_state_dict(llava_vision_param_spec(c), seed=0)
    own = {k[len("vision_tower."):]: v for k, v in sd.items() if k.startswith("vision_tower.")}
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):      # transformers>=5 flattened the wrapper
        own = {k[len("vision_model."):]: v for k, v in own.items()}
    missing, unexpected = hf.load_state_dict(own, strict=False)
    assert not unexpected and all("post_layernorm" in m or "position_ids" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(80)
    rgb = rng.integers(0, 256, (2, 40, 40, 3), dtype=np.uint8)
    px = TR.preprocess_rgb(rgb, c.image)
    with torch.no_grad():
        hs = hf(pixel_values=px, output_hidden_states=True).hidden_states[-2][:, 1:]
        h = torch.nn.functional.gelu(torch.nn.functional.linear(hs, sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"]))
        feats = torch.nn.functional.linear(h, sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"])
    np.savez_compressed(os.path.join(OUT, "g8_llava_vision.npz"), rgb=rgb, hidden=hs.numpy(), feats=feats.numpy())
    print("g8", float(feats.abs().mean()))


def g9():
    from transformers import Phi3Config as HFPhi3Config, Phi3ForCausalLM
    c = SMALL_PHI
    hf = Phi3ForCausalLM(HFPhi3Config(vocab_size=c.vocab, hidden_size=c.hidden, intermediate_size=c.mlp, num_hidden_layers=c.layers,
                                      num_attention_heads=c.heads, num_key_value_heads=c.kv_heads, rms_norm_eps=c.rms_eps, rope_theta=c.rope_theta,
                                      max_position_embeddings=c.max_pos, original_max_position_embeddings=c.max_pos, pad_token_id=0,
                                      tie_word_embeddings=False, attn_implementation="eager")).eval()
    sd = synth_state_dict(phi3_param_spec(c), seed=0)
    own = {k[len("language_model."):]: v for k, v in sd.items()}
    missing, unexpected = hf.load_state_dict(own, strict=False)
    assert not unexpected and not [m for m in missing if "rotary" not in m and "inv_freq" not in m], (missing, unexpected)
    g = torch.Generator().manual_seed(90)
    lengths = [37, 50, 23]
    emb = torch.randn(3, 50, c.hidden, generator=g) * 0.5
    logits = []
    with torch.no_grad():
        for b, L in enumerate(lengths):
            o = hf(inputs_embeds=emb[b:b + 1, :L]).logits[0, -1]
            logits.append(o.numpy())
    np.savez_compressed(os.path.join(OUT, "g9_phi3.npz"), embeds=emb.numpy(), lengths=np.array(lengths), logits=np.stack(logits))
    print("g9", float(np.abs(np.stack(logits)).mean()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g5", "g8", "g9"]
    for w in which:
        globals()[w]()
