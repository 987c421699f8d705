"""Real-weight I/O: builds the state dict `Dynam3D_VLN` expects -- the reference checkpoints' OWN key names --
from the files a Dynam3D user has on disk:

  * llava-phi-3-mini-hf     HF safetensors shards (+ `model.safetensors.index.json`): `language_model.*`, `vision_tower.*`,
                            `multi_modal_projector.*`                                  (VLN-POL:119-131 `from_pretrained`)
  * OpenAI CLIP ViT-L/14@336  `ViT-L-14-336px.pt` (TorchScript archive or plain state dict): `visual.*`
                            (encoders/resnet_encoders.py:260 `clip.load`)
  * `dynam3d.pth`           the pre-trained 3D feature field after `convert_ckpt.py` (bare Feature_Fields keys)  (VLN-POL:77-80)
  * trainer checkpoint      `ckpt.iter*.pth` with `state_dict` entries `net.<name>` / `net.module.<name>`: the policy's position
                            MLPs and projectors, and `net.feature_fields.*` (what `convert_ckpt.py` strips)  (VLN-TR:200-230)

No checkpoint is available offline; tests/test_checkpoints.py round-trips synthetic tensors through these file formats.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterable, Optional

import torch

FF_PREFIXES = ("net.module.feature_fields.", "net.feature_fields.", "module.feature_fields.", "feature_fields.")
NET_PREFIXES = ("net.module.", "net.", "module.")
POLICY_MLPS = ("patch_position_embedding.", "instance_position_embedding.", "zone_position_embedding.", "instance_projector.", "zone_projector.")
# Pretrain-only / frozen-copy keys of the 3DFF checkpoint that the VLN step never reads (PRE-FF:221-243; convert_ckpt.py:17-28)
FF_IGNORED = ("nerf_", "patch_to_nerf", "aggregate_patch_to_nerf", "freezed_", "clip_", "FastSAM")


def load_safetensors_dir(path: str, wanted_prefixes: Iterable[str] = ("language_model.", "vision_tower.", "multi_modal_projector.")) -> Dict[str, torch.Tensor]:
    """A HF model directory (index json + shards, or a single model.safetensors) or one .safetensors file."""
    from safetensors import safe_open
    if os.path.isdir(path):
        idx = os.path.join(path, "model.safetensors.index.json")
        if os.path.isfile(idx):
            files = sorted({os.path.join(path, f) for f in json.load(open(idx))["weight_map"].values()})
        else:
            files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    else:
        files = [path]
    if not files:
        raise FileNotFoundError(f"no .safetensors under {path}")
    out = {}
    wanted = tuple(wanted_prefixes)
    for f in files:
        with safe_open(f, framework="pt", device="cpu") as sf:
            for k in sf.keys():
                kk = canonical_llava_key(k)
                if kk.startswith(wanted):
                    out[kk] = sf.get_tensor(k)
    return out


def canonical_llava_key(k: str) -> str:
    """Checkpoint key -> the transformers-4.46 llava layout the parameter spec uses (`language_model.model.layers.*`,
    `language_model.lm_head.weight`, `vision_tower.vision_model.*`, `multi_modal_projector.*`).  transformers >= 4.52 saves
    LlavaForConditionalGeneration as `model.language_model.layers.*` (the inner `.model` level is gone), `model.vision_tower.*`,
    `model.multi_modal_projector.*` and a top-level `lm_head.weight`."""
    if k == "lm_head.weight":
        return "language_model.lm_head.weight"
    if k.startswith("model.language_model."):
        return "language_model.model." + k[len("model.language_model."):]
    if k.startswith(("model.vision_tower.", "model.multi_modal_projector.")):
        return k[len("model."):]
    return k


def load_clip_pt(path: str) -> Dict[str, torch.Tensor]:
    """`visual.*` tensors of an OpenAI CLIP checkpoint (jit archive as distributed, or a plain state dict)."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:                                     # noqa: BLE001  (not a TorchScript archive)
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()
    return {k: v for k, v in sd.items() if k.startswith("visual.")}


def _strip(key: str, prefixes) -> Optional[str]:
    for p in prefixes:
        if key.startswith(p):
            return key[len(p):]
    return None


def load_dynam3d_pth(path: str) -> Dict[str, torch.Tensor]:
    """`dynam3d.pth` (bare Feature_Fields keys) or an unconverted 3DFF checkpoint (`net.[module.]feature_fields.*`): what
    convert_ckpt.py does, done on load.  Pretrain-only tensors are dropped."""
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd)
    out = {}
    for k, v in sd.items():
        kk = _strip(k, FF_PREFIXES)
        if kk is None:
            kk = k if not k.startswith(NET_PREFIXES) else None
        if kk is None or kk.startswith(FF_IGNORED):
            continue
        out[kk] = v
    return out


def load_trainer_ckpt(path: str) -> Dict[str, torch.Tensor]:
    """Policy MLPs (`net.patch_position_embedding.*` ...) and, if present, the fine-tuned feature field of a VLN trainer checkpoint."""
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd)
    out = {}
    for k, v in sd.items():
        ff = _strip(k, FF_PREFIXES)
        if ff is not None:
            if not ff.startswith(FF_IGNORED):
                out[ff] = v
            continue
        kk = _strip(k, NET_PREFIXES) or k
        if kk.startswith(POLICY_MLPS):
            out[kk] = v
    return out


def load_reference_weights(llava_dir: str, clip_pt: str, dynam3d_pth: Optional[str] = None, trainer_ckpt: Optional[str] = None,
                           cfg=None) -> Dict[str, torch.Tensor]:
    """Merge the four sources (later ones win: a trainer checkpoint's fine-tuned feature field overrides dynam3d.pth) and check
    the result against the parameter spec of `cfg` (default PolicyConfig()): missing or mis-shaped tensors raise."""
    from .policy import PolicyConfig, prefix_param_spec
    from .towers import clip_param_spec, llava_vision_param_spec, phi3_param_spec
    from .weights import ff_param_spec
    cfg = cfg or PolicyConfig()
    sd: Dict[str, torch.Tensor] = {}
    sd.update(load_safetensors_dir(llava_dir))
    sd.update(load_clip_pt(clip_pt))
    if dynam3d_pth:
        sd.update(load_dynam3d_pth(dynam3d_pth))
    if trainer_ckpt:
        sd.update(load_trainer_ckpt(trainer_ckpt))
    spec = (ff_param_spec(768) + prefix_param_spec(768, cfg.llm.hidden) + clip_param_spec(cfg.vit) + llava_vision_param_spec(cfg.vit)
            + phi3_param_spec(cfg.llm))
    missing = [n for n, _ in spec if n not in sd]
    bad = [(n, tuple(sd[n].shape), tuple(shp)) for n, shp in spec if n in sd and tuple(sd[n].shape) != tuple(shp)]
    if missing or bad:
        raise KeyError(f"reference weights incomplete: {len(missing)} missing (first: {missing[:5]}), {len(bad)} mis-shaped (first: {bad[:3]})")
    return sd
