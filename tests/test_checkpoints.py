"""Real-weight I/O (dynam3d_amd/checkpoints.py) on synthetic tensors written in the reference's file formats: HF safetensors
shards + index for llava-phi-3-mini, a CLIP state dict, `dynam3d.pth` in converted AND unconverted form, a trainer checkpoint
with `net.` / `net.module.` prefixes -- loaded back, checked against the parameter spec, and run through the policy."""
import json
import os

import numpy as np
import pytest
import torch

from dynam3d_amd import checkpoints as CK
from dynam3d_amd.policy import Dynam3D_VLN, prefix_param_spec, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
from dynam3d_amd.towers import clip_param_spec, llava_vision_param_spec, phi3_param_spec
from dynam3d_amd.weights import ff_param_spec
from tests.cpu_ops import CpuOps
from tests.test_policy_cpu import SMALL


def _new_hf_key(k):
    """transformers >= 4.52 LlavaForConditionalGeneration layout."""
    if k == "language_model.lm_head.weight":
        return "lm_head.weight"
    if k.startswith("language_model.model."):
        return "model.language_model." + k[len("language_model.model."):]
    return "model." + k


def _write(tmp, sd, cfg, unconverted_ff, new_hf_layout=False):
    from safetensors.torch import save_file
    llava = {k: v.contiguous() for k, v in sd.items() if k.startswith(("language_model.", "vision_tower.", "multi_modal_projector."))}
    if new_hf_layout:
        llava = {_new_hf_key(k): v for k, v in llava.items()}
    names = sorted(llava)
    half = len(names) // 2
    d = os.path.join(tmp, "llava-phi-3-mini-hf")
    os.makedirs(d)
    wmap = {}
    for i, part in enumerate((names[:half], names[half:])):
        fn = f"model-{i + 1:05d}-of-00002.safetensors"
        save_file({k: llava[k] for k in part}, os.path.join(d, fn))
        wmap.update({k: fn for k in part})
    json.dump({"metadata": {}, "weight_map": wmap}, open(os.path.join(d, "model.safetensors.index.json"), "w"))
    clip = os.path.join(tmp, "ViT-L-14-336px.pt")
    torch.save({k: v for k, v in sd.items() if k.startswith("visual.")} | {"logit_scale": torch.ones(())}, clip)
    ff_names = [n for n, _ in ff_param_spec(768)]
    ffp = os.path.join(tmp, "dynam3d.pth")
    if unconverted_ff:                                     # the 3DFF trainer's own checkpoint, before convert_ckpt.py
        torch.save({"state_dict": {"net.module.feature_fields." + k: sd[k] for k in ff_names} | {"net.module.feature_fields.nerf_encoder.params": torch.zeros(3)}}, ffp)
    else:
        torch.save({k: sd[k] for k in ff_names} | {"freezed_aggregate_patch_to_instance_embedding": torch.zeros(1, 768)}, ffp)
    mlp_names = [n for n, _ in prefix_param_spec(768, cfg.llm.hidden)]
    tr = os.path.join(tmp, "ckpt.iter100.pth")
    torch.save({"state_dict": {"net." + k: sd[k] for k in mlp_names} | {"net.some_other_head.weight": torch.zeros(2)}, "iteration": 100}, tr)
    return d, clip, ffp, tr


def test_new_hf_llava_layout_and_prefix_only_ignore_list(tmp_path):
    """A llava checkpoint saved by transformers >= 4.52 (`model.language_model.layers.*`, top-level `lm_head.weight`) loads into the
    4.46 names of the spec; feature-field keys are ignored by PREFIX only (a key that merely contains `clip_` / `nerf_` is kept)."""
    cfg = SMALL
    sd = synth_policy_weights(cfg, seed=0)
    d, clip, ffp, tr = _write(str(tmp_path), sd, cfg, False, new_hf_layout=True)
    got = CK.load_reference_weights(d, clip, ffp, tr, cfg=cfg)
    for n, _ in phi3_param_spec(cfg.llm) + llava_vision_param_spec(cfg.vit):
        assert torch.equal(got[n], sd[n]), n
    assert CK.canonical_llava_key("lm_head.weight") == "language_model.lm_head.weight"
    p = str(tmp_path / "ff2.pth")
    torch.save({"patch_clip_gate.weight": torch.ones(2), "nerf_encoder.params": torch.zeros(3), "clip_text.weight": torch.zeros(1)}, p)
    assert set(CK.load_dynam3d_pth(p)) == {"patch_clip_gate.weight"}


@pytest.mark.parametrize("unconverted_ff", [False, True])
def test_reference_file_formats_round_trip(tmp_path, unconverted_ff):
    cfg = SMALL
    sd = synth_policy_weights(cfg, seed=0)
    d, clip, ffp, tr = _write(str(tmp_path), sd, cfg, unconverted_ff)
    got = CK.load_reference_weights(d, clip, ffp, tr, cfg=cfg)
    spec = (ff_param_spec(768) + prefix_param_spec(768, cfg.llm.hidden) + clip_param_spec(cfg.vit) + llava_vision_param_spec(cfg.vit)
            + phi3_param_spec(cfg.llm))
    for n, _ in spec:
        assert torch.equal(got[n], sd[n]), n
    assert not any(k.startswith(("nerf_", "freezed_", "some_other_head")) for k in got)
    # the loaded dict drives the policy exactly like the in-memory one
    outs = []
    for weights in (sd, got):
        net = Dynam3D_VLN(cfg, weights, device="cpu", batch_size=2, ops=CpuOps(), max_steps=2)
        net.feature_fields.initialize_camera_setting(90.0, 90.0)
        fr = SyntheticEpisodes(2, seed=9, image_hw=224, depth_hw=224).next()
        outs.append(net.forward_logits({"rgb": torch.from_numpy(fr.rgb), "depth": torch.from_numpy(fr.depth)}, [INSTRUCTION_64] * 2,
                                       [p.tolist() for p in fr.positions], list(fr.headings), patch_segm=fr.patch_segm).numpy())
    assert np.array_equal(outs[0], outs[1])


def test_incomplete_weights_are_reported(tmp_path):
    cfg = SMALL
    sd = synth_policy_weights(cfg, seed=0)
    sd.pop("language_model.model.layers.1.mlp.down_proj.weight")
    d, clip, ffp, tr = _write(str(tmp_path), sd | {"language_model.model.layers.1.mlp.down_proj.weight": torch.zeros(3, 3)}, cfg, False)
    with pytest.raises(KeyError, match="mis-shaped"):
        CK.load_reference_weights(d, clip, ffp, tr, cfg=cfg)
    with pytest.raises(KeyError, match="missing"):
        CK.load_reference_weights(d, clip, None, tr, cfg=cfg)                     # no feature field at all
