"""nn.Module plumbing of the drop-in surface.

The reference trainer treats the policy net as an ordinary `torch.nn.Module` BEFORE it runs a single step
(Dynam3D_VLN/vlnce_baselines/ss_trainer_Dynam3D.py:183-219, 305-311, 374, 568): `policy.to(device)`,
`Adafactor(policy.net.parameters())`, `policy.load_state_dict(ckpt["state_dict"], strict=False)`, `policy.train()/.eval()`,
`policy.net.rgb_encoder.eval()`, `policy.net.depth_encoder.eval()`, `hasattr(policy.net, "module")`.

The kernels, however, want their own weight layouts (fused QKV, gate/up rows interleaved per 16, K padded to 64, float32 copies of
16-bit norm gains).  So the weights live TWICE, by design:

  * `ParamTree` -- plain `nn.Module` containers that hold one `nn.Parameter` per reference tensor under the reference's own dotted
    state-dict key (`llava.language_model.model.layers.3.mlp.gate_up_proj.weight`, `rgb_encoder.model.visual.conv1.weight`,
    `feature_fields.aggregate_patch_to_instance_encoder.layers.0.linear1.weight`, ...): this is what `parameters()`, `state_dict()`,
    `load_state_dict()`, `to()` see;
  * the compute objects (`towers.ClipVisionTower`, `LlavaVisionTower`, `Phi3Decoder`, `ff_dense.FFDense`) hold the kernel layouts.
    Where a layout IS the reference's (same dtype, contiguous [out, in]) the compute object aliases the parameter's storage -- no copy;
    the rest is re-laid-out by `refresh()` whenever the parameters were replaced or overwritten (`_apply`, load-state-dict post hook).
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import torch
from torch import nn


class ParamTree(nn.Module):
    """Container of parameters under dotted names: `add("model.layers.0.mlp.down_proj.weight", t)` creates the nested containers
    `model` / `layers` / `0` / `mlp` / `down_proj` and registers `weight` on the last one, so `state_dict()` emits exactly that key."""

    def __init__(self, named: Optional[Dict[str, torch.Tensor]] = None, requires_grad: bool = False):
        super().__init__()
        for k, v in (named or {}).items():
            self.add(k, v, requires_grad)

    def add(self, dotted: str, tensor: torch.Tensor, requires_grad: bool = False):
        install_param(self, dotted, tensor, requires_grad)

    def flat(self) -> Dict[str, torch.Tensor]:
        """dotted name -> the parameter's data (detached view of the SAME storage)."""
        return flat_params(self)

    def forward(self, *a, **k):
        raise RuntimeError("ParamTree only holds parameters; the compute objects built from it run the kernels")


def own_copy(t: torch.Tensor, device, dtype) -> torch.Tensor:
    """`t` on `device` in `dtype`, contiguous, in storage the module OWNS: `.to()` / `.contiguous()` return the caller's own tensor when
    nothing has to change (weights synthesised on the device in the target dtype), and a parameter aliasing the caller's state dict
    would let `load_state_dict`, an optimizer step or `sync_weights` write through into it and into every sibling model built from the
    same dict -- cloned in that case."""
    src = t.detach()
    out = src.to(device, dtype).contiguous()
    if out.data_ptr() == src.data_ptr() and out.numel() > 0:
        out = out.clone()
    return out


def install_param(root: nn.Module, dotted: str, tensor: torch.Tensor, requires_grad: bool = False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        child = mod._modules.get(p)
        if child is None:
            child = ParamTree()
            mod.add_module(p, child)
        mod = child
    rg = bool(requires_grad) and tensor.is_floating_point()
    mod.register_parameter(parts[-1], nn.Parameter(tensor.detach(), requires_grad=rg))


def install_params(root: nn.Module, named: Dict[str, torch.Tensor], requires_grad: bool = False):
    for k, v in named.items():
        install_param(root, k, v, requires_grad)


def flat_params(root: nn.Module, prefix: str = "") -> Dict[str, torch.Tensor]:
    """name -> detached view of every parameter below `root` (optionally only those whose name starts with `prefix`)."""
    return {k: p.detach() for k, p in root.named_parameters() if k.startswith(prefix)}


def storage_signature(params: Iterable[torch.Tensor]):
    """Identity of a set of parameters: a change of storage / device / dtype (`to()`, `half()`, `assign=True` loads) or an in-place
    write (`load_state_dict`, an optimizer step: the version counter) changes it."""
    return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)


class RefreshOnChange(nn.Module):
    """Mixin: calls `self.refresh()` after everything that replaces or overwrites parameters through the nn.Module protocol:
    `_apply` (to / cuda / cpu / float / half / bfloat16) and `load_state_dict` -- also when the call arrives through a PARENT module
    (`policy.load_state_dict(...)` recurses with `_load_from_state_dict`, never calling the child's own `load_state_dict`; the
    post hook registered here runs for every module of that recursion)."""

    def _init_refresh_hooks(self):
        self._refresh_enabled = True
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._do_refresh())

    def _do_refresh(self):
        if getattr(self, "_refresh_enabled", False):
            self.refresh()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._do_refresh()
        return out

    def refresh(self):          # pragma: no cover - overridden
        raise NotImplementedError


class DepthEncoderSlot(nn.Module):
    """Placeholder for `net.depth_encoder` (VLN-POL:137-143, `VlnResnetDepthEncoder`): the DDPPO depth ResNet feeds only the
    waypoint predictor (`get_candidate_waypoints`, VLN-POL:188-292), which SURVEY.md 2 puts outside the hot path.  The trainer
    still touches the attribute (`policy.net.depth_encoder.eval()`, VLN-TR:307-311; checkpoints carry `net.depth_encoder.*`), so the slot
    exists, accepts and ignores those keys, and raises if someone tries to run it.  Plug the reference's module in with
    `Dynam3D_VLN(depth_encoder=VlnResnetDepthEncoder(...))` or `net.depth_encoder = ...` on a machine that has Habitat."""
    is_blind = False

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        return          # `depth_encoder.*` keys of a reference checkpoint are accepted and ignored (not reported as unexpected)

    def forward(self, observations):
        raise NotImplementedError("depth_encoder is outside the hot path (SURVEY.md section 2): pass the reference's VlnResnetDepthEncoder "
                                  "as Dynam3D_VLN(depth_encoder=...) to use get_candidate_waypoints")


def make_prefix_mlp(din: int, dh: int, dout: int, device, named: Optional[Dict[str, torch.Tensor]] = None, prefix: str = "") -> nn.Sequential:
    """nn.Sequential(Linear, LayerNorm, GELU, Linear) (VLN-POL:83-111), float32, filled from `named[prefix + '0.weight']` ..."""
    seq = nn.Sequential(nn.Linear(din, dh), nn.LayerNorm(dh), nn.GELU(), nn.Linear(dh, dout)).to(device=device, dtype=torch.float32)
    if named is not None:
        with torch.no_grad():
            for k, p in seq.named_parameters():
                p.copy_(named[prefix + k].to(p.device, p.dtype))
    return seq
