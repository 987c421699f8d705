"""Deterministic, name-keyed synthetic weights.

No checkpoint of any kind is available offline, so every tensor is generated from
(seed, parameter name, shape) with a CPU generator.  The SAME function fills the product modules,
the CPU oracle and -- in the container -- the reference's own modules (oracle/ref_harness.py), so
all three compute with bit-identical parameters.

Parameter names follow the reference's state-dict keys (VLN-FF:139-161, VLN-POL:83-111,
clip/model.py:202-217) so a real `dynam3d.pth` / CLIP state-dict can be loaded through the same
dictionaries.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Iterable, Tuple

import torch


def _name_seed(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little") & 0x7FFF_FFFF_FFFF_FFFF


def make_tensor(seed: int, name: str, shape: Tuple[int, ...], kind: str = "auto", fan_in: int | None = None,
                device: str = "cpu") -> torch.Tensor:
    """kind: 'linear' (N(0, fan_in^-1/2)), 'bias' (N(0,0.02)), 'norm_w' (1+N(0,0.02)), 'embed' (N(0, d^-1/2)).
    device='cpu' values are the cross-machine reproducible ones used for parity; device='cuda' draws from the
    GPU generator (different numbers, same distribution) and exists so the 4.4 B-parameter benchmark model
    does not spend a minute in a single-threaded host RNG."""
    g = torch.Generator(device=device)
    g.manual_seed(_name_seed(seed, name))
    _randn = torch.randn
    torch_randn = lambda shape, generator, dtype: _randn(shape, generator=generator, dtype=dtype, device=device)
    if kind == "auto":
        if name.endswith("bias"):
            kind = "bias"
        elif len(shape) == 1:
            kind = "norm_w"
        else:
            kind = "linear"
    if kind == "linear":
        fi = fan_in if fan_in is not None else int(torch.tensor(shape[1:]).prod().item())
        return torch_randn(shape, generator=g, dtype=torch.float32) * (fi ** -0.5)
    if kind == "bias":
        return torch_randn(shape, generator=g, dtype=torch.float32) * 0.02
    if kind == "norm_w":
        return 1.0 + torch_randn(shape, generator=g, dtype=torch.float32) * 0.02
    if kind == "embed":
        return torch_randn(shape, generator=g, dtype=torch.float32) * (shape[-1] ** -0.5)
    raise ValueError(kind)


def _is_norm(name: str) -> bool:
    parts = name.split(".")
    return any(p.startswith("norm") or p.startswith("ln_") or p == "ln" or p.endswith("layernorm") or p.endswith("norm")
               for p in parts)


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, norm_names: Iterable[str] = (),
                     device: str = "cpu", dtype_for=None) -> Dict[str, torch.Tensor]:
    """spec: iterable of (name, shape).  1-D '...weight' tensors are norm gains, 1-D '...bias' are
    biases, >=2-D are linear/conv/embedding matrices."""
    out = {}
    norm_names = set(norm_names)
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        if name.endswith("bias"):
            kind = "bias"
        elif len(shape) == 1 and (name.endswith("weight") or name in norm_names):
            kind = "norm_w"
        elif "embedding" in name or name.endswith("proj") and len(shape) == 2 and "." not in name:
            kind = "embed"
        else:
            kind = "linear"
        out[name] = make_tensor(seed, name, shape, kind, device=device)
        if dtype_for is not None:
            out[name] = out[name].to(dtype_for(name))
        if name == "instance_merge_discriminator.0.weight":
            # synthetic-only: make the merge decision depend visibly on the 3-d position offset so
            # seeded episodes exercise BOTH the "new instance" and the "merge" branch (with plain
            # random weights the LayerNorm'ed features give near-constant logits).
            out[name][:, -3:] *= 160.0
        if name == "instance_merge_discriminator.3.bias":
            out[name] += torch.tensor([0.4, -0.4], device=out[name].device)   # ~25 % positive proposals on the synthetic episodes
    return out


def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> Dict[str, torch.Tensor]:
    """Overwrite every parameter of `module` in place with the name-keyed synthetic tensor."""
    spec = [(prefix + n, tuple(p.shape)) for n, p in module.named_parameters()]
    sd = synth_state_dict(spec, seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(sd[prefix + n].to(p.dtype))
    return sd


def ff_param_spec(width: int = 768):
    """(name, shape) for every parameter of the VLN Feature_Fields (VLN-FF:134-161)."""
    spec = []

    def seq(name, din, dh, dout):
        spec.extend([(f"{name}.0.weight", (dh, din)), (f"{name}.0.bias", (dh,)),
                     (f"{name}.1.weight", (dh,)), (f"{name}.1.bias", (dh,)),
                     (f"{name}.3.weight", (dout, dh)), (f"{name}.3.bias", (dout,))])

    def enc(name):
        for i in range(2):
            p = f"{name}.layers.{i}"
            spec.extend([(p + ".self_attn.in_proj_weight", (3 * width, width)), (p + ".self_attn.in_proj_bias", (3 * width,)),
                         (p + ".self_attn.out_proj.weight", (width, width)), (p + ".self_attn.out_proj.bias", (width,)),
                         (p + ".linear1.weight", (4 * width, width)), (p + ".linear1.bias", (4 * width,)),
                         (p + ".linear2.weight", (width, 4 * width)), (p + ".linear2.bias", (width,)),
                         (p + ".norm1.weight", (width,)), (p + ".norm1.bias", (width,)),
                         (p + ".norm2.weight", (width,)), (p + ".norm2.bias", (width,))])
        spec.extend([(name + ".norm.weight", (width,)), (name + ".norm.bias", (width,))])

    seq("patch_to_instance_position_embedding", 7, width, width)
    spec.append(("aggregate_patch_to_instance_embedding", (1, width)))
    enc("aggregate_patch_to_instance_encoder")
    seq("instance_to_zone_position_embedding", 4, width, width)
    spec.append(("aggregate_instance_to_zone_embedding", (1, width)))
    enc("aggregate_instance_to_zone_encoder")
    seq("instance_merge_discriminator", 2 * width + 3, 4 * width, 2)
    return spec


def render_param_spec(width: int = 768, k: int = 4):
    """Pretrain-only parameters of the novel-view renderer (PRE-FF:221-254).  The tcnn networks are stored as
    per-layer [out,in] matrices `nerf_*.layers.{i}.weight` (tcnn itself keeps one flat `params` vector)."""
    s = [("patch_to_nerf_position_embedding.0.weight", (width, 6)), ("patch_to_nerf_position_embedding.0.bias", (width,)),
         ("patch_to_nerf_position_embedding.1.weight", (width,)), ("patch_to_nerf_position_embedding.1.bias", (width,)),
         ("aggregate_patch_to_nerf_encoder.0.weight", (width, width * k)), ("aggregate_patch_to_nerf_encoder.0.bias", (width,)),
         ("aggregate_patch_to_nerf_encoder.1.weight", (width,)), ("aggregate_patch_to_nerf_encoder.1.bias", (width,))]
    s += [("nerf_encoder.layers.0.weight", (width, width)), ("nerf_encoder.layers.1.weight", (width, width)),
          ("nerf_encoder.layers.2.weight", (width + 1, width))]
    s += [(f"nerf_decoder.layers.{i}.weight", (width, width)) for i in range(3)]
    return s
