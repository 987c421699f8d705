"""`Dynam3D_VLN` -- drop-in for the reference policy net's per-step call
(Dynam3D_VLN/vlnce_baselines/models/Policy_Dynam3D_VLN.py:66-506, "VLN-POL"):

    net(observations, instructions, agent_positions, agent_heading_angles, depth_scale=(0.,10.),
        gt_text=None, delete_old_features=True, num_of_views=1, is_train=False)          (VLN-POL:329, VLN-TR:671)

plus `net.feature_fields.*`, `net.convert_text_to_action(texts)` and -- for the benchmark metric --
`net.forward_logits(...) -> (B, vocab)`: the LM logits at the last prompt position of the very same
`inputs_embeds` the reference hands to `llava.generate`.

MI355X-first differences from the reference's data flow (results unchanged):
  * nothing leaves the GPU between the camera frame and the logits: CLIP grid features stay on the
    device as fp16 (the reference round-trips them through numpy, VLN-POL:345), depth is resized and
    pre-processed by one HIP kernel instead of a per-image cv2 loop (VLN-POL:336-341);
  * one image pre-processing feeds both ViT towers;
  * the batch is right-padded with per-row offsets and per-row action history.
"""
from __future__ import annotations

import math
import re
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dense_ops as D
from .adapters import PromptTokenizer
from .feature_fields import Feature_Fields
from .modules import DepthEncoderSlot, RefreshOnChange, make_prefix_mlp
from .profiling import TIMER
from .towers import (ClipEncoder, LlavaModel, Phi3Config, VitConfig, clip_param_spec, llava_vision_param_spec, phi3_param_spec,
                     preprocess_rgb)
from .weights import ff_param_spec, synth_state_dict

PREFIX_MLPS = ("patch_position_embedding", "instance_position_embedding", "zone_position_embedding", "instance_projector", "zone_projector")


def prefix_param_spec(width: int = 768, hidden: int = None):
    """VLN-POL:83-111 (hidden = width*4 = the LM width in the reference; separately settable for small test configs)."""
    hidden = hidden or 4 * width
    s = []

    def seq(name, din, dh, dout):
        s.extend([(f"{name}.0.weight", (dh, din)), (f"{name}.0.bias", (dh,)), (f"{name}.1.weight", (dh,)), (f"{name}.1.bias", (dh,)),
                  (f"{name}.3.weight", (dout, dh)), (f"{name}.3.bias", (dout,))])

    seq("patch_position_embedding", 6, hidden, hidden)
    seq("instance_position_embedding", 3, width, width)
    seq("zone_position_embedding", 3, width, width)
    seq("instance_projector", 2 * width, hidden, hidden)
    seq("zone_projector", 2 * width, hidden, hidden)
    return s


class SyntheticTokenizer(PromptTokenizer):
    """Deterministic stand-in for the llava-phi-3-mini tokenizer (its files are not available offline:
    "parity unpinned" at this boundary; `adapters.HFTokenizerAdapter` wraps the real one).  Special tokens take the
    Phi-3 / llava ids; every other whitespace-delimited piece is hashed into the ordinary vocabulary range.  Like
    Phi-3's own tokenizer (`add_bos_token: true`) `encode` puts <s> in front of the text."""
    BOS, NEWLINE = 1, 13
    SPECIAL = {"<|user|>": 32010, "<|end|>": 32007, "<|assistant|>": 32001, "<image>": 32038, "<|endoftext|>": 32000}

    def __init__(self, vocab: int = 32064, add_bos: bool = True):
        self.vocab, self.add_bos = vocab, add_bos
        self._re = re.compile(r"(<\|[a-z]+\|>|<image>|\n|[^\s<]+|<)")
        self._memo = {}                 # text -> ids of the last few hundred distinct texts (an instruction is encoded once per episode, not per step)

    def encode(self, text: str) -> List[int]:
        hit = self._memo.get(text)
        if hit is not None:
            return list(hit)
        ids = self._encode(text)
        if len(self._memo) >= 512:
            self._memo.clear()
        self._memo[text] = tuple(ids)
        return ids

    def _encode(self, text: str) -> List[int]:
        out = [self.BOS] if self.add_bos else []
        hi = min(32000, self.vocab) - 100
        for piece in self._re.findall(text):
            if piece in self.SPECIAL:
                out.append(self.SPECIAL[piece] % self.vocab)
            elif piece == "\n":
                out.append(self.NEWLINE)
            else:
                out.append(100 + zlib.crc32(piece.encode()) % hi)
        return out

    def split_prompt(self, head: str, n_visual: int, tail: str):
        """Same result as the generic `PromptTokenizer.split_prompt` (asserted in tests/test_policy_cpu.py) without running the regex
        over hundreds of "<image>" placeholders per prompt and step: every piece of this tokenizer is context free, so
        ids(head + "<image>" * n + tail) = ids(head) + [image id] * n + ids(tail)[without BOS]."""
        h = self.encode(head)
        t = self.encode(tail)[1 if self.add_bos else 0:]
        full = h + [self.SPECIAL["<image>"] % self.vocab] * n_visual + t
        return full[:2], full[n_visual + 2:]

    def decode(self, ids: Sequence[int]) -> str:
        inv = {v % self.vocab: k for k, v in self.SPECIAL.items()}
        return " ".join(inv.get(int(i), "\n" if int(i) == self.NEWLINE else f"<{int(i)}>") for i in ids)


class ActionGrammarTokenizer(SyntheticTokenizer):
    """`SyntheticTokenizer` whose `decode` renders a GENERATED id sequence as a sentence of the policy's action language
    ("turn left 2 steps, move 3 steps." / "stop.", VLN-POL:294-327).  No trained llava weights exist offline, so the tokens the LM
    emits with seeded random weights mean nothing; this grammar gives the closed loop of config[3] (rollout.py) sentences that
    `convert_text_to_action` can act on -- every branch reachable: left / right turns of 0-4 steps, moves of 0-4 steps, a capped
    5-step turn (no move), stop, and a malformed sentence (-> -100 -> stop, like the trainer treats it, VLN-TR:695-696).
    A function of the ids only: the same tokens give the same sentence everywhere."""

    def __init__(self, vocab: int = 32064, add_bos: bool = True, stop_mod: int = 24):
        super().__init__(vocab, add_bos)
        self.stop_mod = stop_mod

    def decode(self, ids: Sequence[int]) -> str:
        end = self.SPECIAL["<|end|>"] % self.vocab
        ids = [int(i) for i in ids]
        if end in ids:
            ids = ids[: ids.index(end)]
        if not ids:
            return "stop."
        h = zlib.crc32(np.asarray(ids, np.int64).tobytes())            # all generated ids decide, well mixed
        a = h % self.stop_mod
        if a == 0:
            return "stop."
        if a == 1:
            return "turn around and go back"                           # malformed on purpose
        side = "left" if (h >> 8) & 1 == 0 else "right"
        return f"turn {side} {(h >> 12) % 6} steps, move {(h >> 20) % 5} steps."


@dataclass
class PolicyConfig:
    vit: VitConfig = field(default_factory=VitConfig)
    llm: Phi3Config = field(default_factory=Phi3Config)
    clip_dtype: torch.dtype = torch.float16      # reference: clip.load(..., device=cuda) -> fp16 (resnet_encoders.py:260)
    llava_dtype: torch.dtype = torch.bfloat16    # reference: torch_dtype=torch.bfloat16 (VLN-POL:125)
    depth_quirk: bool = False                    # True reproduces `observations['depth'][b][i]` row indexing
    compat: str = "reference"
    hip_dense: bool = True                       # on a GPU the towers run on the hand-written HIP kernels (dense_ops.enable_hip_kernels);
                                                 # False keeps whatever dense_ops.BACKEND says (PyTorch-ROCm libraries by default)


def synth_policy_weights(cfg: PolicyConfig, seed: int = 0, device: str = "cpu") -> Dict[str, torch.Tensor]:
    spec = (ff_param_spec(768) + prefix_param_spec(768, cfg.llm.hidden) + clip_param_spec(cfg.vit) + llava_vision_param_spec(cfg.vit)
            + phi3_param_spec(cfg.llm))
    dtype_for = None
    if device != "cpu":      # benchmark-sized model generated on the GPU: store tower weights directly in their compute dtype
        def dtype_for(name):
            if name.startswith("visual."):
                return cfg.clip_dtype if not name.split(".")[-2].startswith("ln_") else torch.float32
            if name.startswith(("vision_tower.", "multi_modal_projector.", "language_model.")):
                return cfg.llava_dtype if "norm" not in name else torch.float32
            return torch.float32
    return synth_state_dict(spec, seed, device=device, dtype_for=dtype_for)


class Dynam3D_VLN(RefreshOnChange):
    """A `torch.nn.Module` with the reference net's module tree (VLN-POL:66-155), so that the UNCHANGED trainer can hold it as
    `ILPolicy.net` (models/policy.py:12-19): `feature_fields.*`, the five prefix MLPs as `nn.Sequential`s, `llava.language_model.* /
    .vision_tower.* / .multi_modal_projector.*`, `rgb_encoder.model.visual.*`, `depth_encoder` (a slot, modules.DepthEncoderSlot) --
    `parameters()`, `to()`, `train()/eval()`, `state_dict()` / `load_state_dict(strict=False)` work as VLN-TR:183-219, 305-311, 374 use
    them; `requires_grad` flags follow VLN-POL:150-155.  The kernels read their own layouts of these tensors (modules.py); `refresh()`
    re-derives them after the parameters were replaced or overwritten and is called by the nn.Module hooks.  Call it yourself after
    writing to `param.data` directly (an optimizer step)."""

    def __init__(self, cfg: PolicyConfig = PolicyConfig(), weights: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0,
                 device="cuda", batch_size: int = 1, ops=None, tokenizer=None, segmenter=None, max_steps: int = 64,
                 depth_encoder: Optional[torch.nn.Module] = None, ff_planner: Optional[str] = None):
        """ff_planner: `Feature_Fields(planner=...)` -- "device" / "host" bookkeeping of the 3D memory (default: $D3D_FF_PLANNER, else device)."""
        super().__init__()
        self.cfg, self.device = cfg, torch.device(device)
        if self.device.type == "cuda" and cfg.hip_dense:
            D.enable_hip_kernels(["all"])             # before the towers are built: they lay their weights out for these kernels
        sd = weights if weights is not None else synth_policy_weights(cfg, seed)
        ff_sd = {k: sd[k] for k, _ in ff_param_spec(768)}
        self.feature_fields = Feature_Fields(batch_size, device, ff_sd, compat=cfg.compat, ops=ops, segmenter=segmenter, max_steps=max_steps,
                                             planner=ff_planner)
        self.feature_fields.requires_grad_(False)                                             # VLN-POL:150-151 ("Avoid DDP bug")
        self.ops = self.feature_fields.ops
        width, hidden = 768, cfg.llm.hidden
        for name, (din, dh, dout) in zip(PREFIX_MLPS, ((6, hidden, hidden), (3, width, width), (3, width, width),
                                                       (2 * width, hidden, hidden), (2 * width, hidden, hidden))):        # VLN-POL:83-111
            setattr(self, name, make_prefix_mlp(din, dh, dout, self.device, sd, name + "."))
        self.llava = LlavaModel(sd, cfg.vit, cfg.llm, cfg.llava_dtype, device)
        self.depth_encoder = depth_encoder if depth_encoder is not None else DepthEncoderSlot()
        self.space_pool_depth = torch.nn.Sequential(torch.nn.AdaptiveAvgPool2d((1, 1)), torch.nn.Flatten(start_dim=2))   # VLN-POL:144
        self.rgb_encoder = ClipEncoder(sd, cfg.vit, cfg.clip_dtype, device)
        self.pano_img_idxes = np.arange(0, 12, dtype=np.int64)                                # VLN-POL:147-149 (read by get_candidate_waypoints)
        ang = torch.from_numpy((1 - self.pano_img_idxes / 12) * 2 * math.pi)
        self.pano_angle_fts = torch.stack([torch.sin(ang), torch.cos(ang), torch.sin(torch.zeros_like(ang)), torch.cos(torch.zeros_like(ang))], 1).float()
        self.tokenizer = tokenizer or SyntheticTokenizer(cfg.llm.vocab)
        self.last_lengths = None
        self.keep_prompt, self.last_prompt = False, None
        self._mlp_sig = None
        self._init_refresh_hooks()
        self.refresh()

    # the compute objects over `self.llava`'s parameters
    @property
    def llava_vision(self):
        return self.llava.vision

    @property
    def llm(self):
        return self.llava.lm

    def refresh(self):
        """Re-derive every kernel-side weight layout from the current parameters (no-op for parts whose storages are unchanged)."""
        self.feature_fields.refresh()
        self.rgb_encoder.refresh()
        self.llava.refresh()
        mlp_params = [p for n in PREFIX_MLPS for p in getattr(self, n).parameters()]
        sig = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in mlp_params)
        if sig != self._mlp_sig:
            self._mlp_sig = sig
            self.mlp_w = {f"{n}.{k}": p.detach().to(self.device, torch.float32).contiguous() for n in PREFIX_MLPS for k, p in getattr(self, n).named_parameters()}
            self._lowp_w = {}

    # ---- habitat `Net` properties the trainer / registry reads (VLN-POL:158-169) --------------------------------------
    @property
    def output_size(self):
        return 1

    @property
    def is_blind(self):
        return False

    @property
    def num_recurrent_layers(self):
        return 1

    def _cull_stream(self):
        if getattr(self, "_cull", None) is None:
            self._cull = torch.cuda.Stream(device=self.device)
        return self._cull

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            # D3D_LLAVA_STREAM_PRIO=-1: the llava tower's stream at HIGH priority, so that its large grids win the dispatcher over the
            # token builder's small float32 kernels that run beside it (the builder has ~1.3 ms of slack under the tower).  Experiment knob.
            prio = int(__import__("os").environ.get("D3D_LLAVA_STREAM_PRIO", "0"))
            self._side = torch.cuda.Stream(device=self.device, priority=prio)
        return self._side

    # ---- reference surface -----------------------------------------------------------------------------
    def preprocess_depth(self, depth, depth_scale=(0.0, 10.0)):
        """VLN-POL:171-186; depth (B,H,W,1) -> same shape, metres."""
        d = depth.to(self.device)
        return self.ops.preprocess_depth(d.reshape(d.shape[0], d.shape[1], d.shape[2]), depth_scale[0], depth_scale[1]).view(d.shape)

    def _mlp(self, x, name, lowp=False):
        """nn.Sequential(Linear, LayerNorm, GELU, Linear) (VLN-POL:83-111).  `lowp`: run the second (hidden x hidden) layer
        as a 16-bit MFMA GEMM in the LM's dtype -- its result is added to LM-dtype features anyway (VLN-POL:453), and the
        reference evaluates these layers under autocast."""
        import torch.nn.functional as F
        w = self.mlp_w
        if x.is_cuda:                                    # float32 kernels of the token builder (csrc/f32_kernels.hip): Linear, then LN + GELU fused
            f = self.feature_fields.dense._f32()
            x2 = x.reshape(-1, x.shape[-1])
            h = f.layer_norm(f.linear(x2, w[name + ".0.weight"], w[name + ".0.bias"]), w[name + ".1.weight"], w[name + ".1.bias"], 1e-5, gelu=True)
            h = h.view(*x.shape[:-1], -1)
            if not (lowp and self.cfg.llava_dtype != torch.float32):
                return f.linear(h.reshape(-1, h.shape[-1]), w[name + ".3.weight"], w[name + ".3.bias"]).view(*x.shape[:-1], -1)
        else:
            h = F.linear(x, w[name + ".0.weight"], w[name + ".0.bias"])
            h = F.gelu(F.layer_norm(h, (h.shape[-1],), w[name + ".1.weight"], w[name + ".1.bias"], 1e-5))
        if lowp and x.is_cuda and self.cfg.llava_dtype != torch.float32:
            dt = self.cfg.llava_dtype
            key = name + ".3.weight." + str(dt)
            if key not in self._lowp_w:
                self._lowp_w[key] = (w[name + ".3.weight"].to(dt).contiguous(), w[name + ".3.bias"].to(dt).contiguous())
            w3, b3 = self._lowp_w[key]
            return D.linear(h.reshape(-1, h.shape[-1]).to(dt), w3, b3).view(*h.shape[:-1], -1)
        return F.linear(h, w[name + ".3.weight"], w[name + ".3.bias"])

    def _patch_position_tokens(self, depth24, B, V):
        """patch_position_embedding over [x, y, z, sin d, cos d, scale] of every patch (VLN-POL:432-433); needs the depth only."""
        rel_x, rel_y, rel_z, direction, scale = self.feature_fields.get_patch_3d_info(depth24.reshape(B * V, -1))
        info = torch.cat([rel_x, rel_y, rel_z, torch.sin(direction), torch.cos(direction), scale], dim=-1)
        return self._mlp(info, "patch_position_embedding", lowp=True)

    def _depth24(self, depth, V, depth_scale):
        B = depth.shape[0]
        a = self.feature_fields.args
        if not self.cfg.depth_quirk:
            return self.ops.resize_nearest_preprocess(depth[..., 0], a.input_height, a.input_width, *depth_scale).view(B // V, V, -1)
        # quirk F9: depth[b][i] is image ROW i (shape (W,1)); cv2 resizes that column vector to 24x24
        Wd = depth.shape[2]
        ri = torch.from_numpy(np.minimum(np.floor(np.arange(a.input_height) * (Wd / a.input_height)).astype(np.int64), Wd - 1)).to(depth.device)
        rows = torch.stack([depth[b // V, b % V, :, 0][ri] for b in range(B)])            # (B,24)
        img = rows[:, :, None].expand(B, a.input_height, a.input_width).contiguous()
        return self.ops.resize_nearest_preprocess(img, a.input_height, a.input_width, *depth_scale).view(B // V, V, -1)

    @torch.no_grad()
    def build_inputs(self, observations, instructions, agent_positions, agent_heading_angles, depth_scale=(0.0, 10.0),
                     delete_old_features=True, num_of_views=1, patch_segm=None, return_rows=False):
        """Everything up to the LM: returns (inputs_embeds (B,S,3072) right-padded, lengths (B,))  (VLN-POL:331-461)."""
        ff, V = self.feature_fields, num_of_views
        B = ff.batch_size
        rgb = observations["rgb"].to(self.device)
        depth = observations["depth"].to(self.device, torch.float32)
        depth24 = self._depth24(depth, V, (0.0, 10.0))                                        # (B,V,576) metres.  VLN-POL:341 calls
        # `self.preprocess_depth(batch_depth_fts)` WITHOUT depth_scale: the 24x24 depth is always scaled by the (0, 10) default;
        # only the full-resolution depth of the frustum cull takes the caller's depth_scale (VLN-POL:350)
        pixels = preprocess_rgb(rgb)                                                          # shared by both towers
        cuda = self.device.type == "cuda"
        dfull = patch_pos = None
        if delete_old_features:
            dfull = self.ops.preprocess_depth(depth[..., 0], *depth_scale).view(B, V, depth.shape[1], depth.shape[2])
        if cuda:
            main = torch.cuda.current_stream()
            ready = main.record_event()                                                       # depth / pools ready, CLIP not yet queued
        # The llava tower is released behind CLIP block `LLAVA_AFTER_CLIP_BLOCK` (not behind the whole CLIP tower): it is the longer of the two
        # things that follow CLIP (the 3D-token update's host-paced chain and the llava tower), so it gets a head start under CLIP's tail.
        llava_go = None
        clip = self.rgb_encoder.tower
        if cuda and 0 <= self.LLAVA_AFTER_CLIP_BLOCK < clip.cfg.layers - 1:
            llava_go = torch.cuda.Event()
            clip.after_block = (self.LLAVA_AFTER_CLIP_BLOCK, lambda: llava_go.record(main))
        try:
            _, grid = clip.forward(pixels)                                                    # (B*V,576,768) fp16, stays on device
        finally:
            clip.after_block = None
        self.last_grid = grid                                                                 # (bench.py hands these to the CPU oracle's memory advance)
        # The frustum cull needs the depth and the stored rows, not the CLIP features: it runs -- with its host round trip for the
        # hit lists -- on a third stream UNDER the CLIP tower, which the host has only queued at this point.
        if delete_old_features:
            if cuda:
                cull = self._cull_stream()
                cull.wait_event(ready)
                with torch.cuda.stream(cull):
                    ff.delete_old_features_from_camera_frustum(dfull, agent_positions, agent_heading_angles, num_of_views=V)
                    patch_pos = self._patch_position_tokens(depth24, B, V)                   # depth only: also under the CLIP tower
                dfull.record_stream(cull)
                depth24.record_stream(cull)
                main.wait_stream(cull)
                patch_pos.record_stream(main)
            else:
                ff.delete_old_features_from_camera_frustum(dfull, agent_positions, agent_heading_angles, num_of_views=V)
        # The llava vision tower only needs `pixels`: run it on a second HIP stream underneath the 3D-token
        # update, whose host round trips (merge decisions, Ni/Nz) would otherwise idle the GPU.
        side = None
        if cuda:
            side = self._side_stream()
            if llava_go is not None:
                side.wait_event(llava_go)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                patch_feat = self.llava_vision.forward(pixels)
            pixels.record_stream(side)
        ff.update_feature_fields(depth24, grid.view(B, V, ff.P, -1), rgb, agent_positions, agent_heading_angles, num_of_views=V,
                                 patch_segm=patch_segm)
        with TIMER.range("prefix.query"):
            env = ff.get_environment_features(agent_positions, agent_heading_angles)
        with TIMER.range("prefix.mlps"):
            if patch_pos is None:
                patch_pos = self._patch_position_tokens(depth24, B, V)                            # (B*V,576,3072)
            ni = [int(t.shape[0]) for t in env["batch_instance_fts"]]
            nz = [int(t.shape[0]) for t in env["batch_zone_fts"]]
            ifts, irel = torch.cat(env["batch_instance_fts"]), torch.cat(env["batch_instance_relative_position"])
            zfts, zrel = torch.cat(env["batch_zone_fts"]), torch.cat(env["batch_zone_relative_position"])
            inst_tok = self._mlp(torch.cat([ifts, self._mlp(irel, "instance_position_embedding")], -1), "instance_projector", lowp=True)   # VLN-POL:434
            zone_tok = self._mlp(torch.cat([zfts, self._mlp(zrel, "zone_position_embedding")], -1), "zone_projector", lowp=True)           # VLN-POL:435
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            patch_feat.record_stream(torch.cuda.current_stream())
        else:
            patch_feat = self.llava_vision.forward(pixels)
        with TIMER.range("prefix.rows"):
            if return_rows == "packed":
                return self._assemble_packed(patch_feat, patch_pos, inst_tok, zone_tok, ni, nz, instructions, B, V)
            return self._assemble_rows(patch_feat, patch_pos, inst_tok, zone_tok, ni, nz, instructions, B, V, return_rows)

    @torch.no_grad()
    def advance_memory(self, observations, agent_positions, agent_heading_angles, grid, depth_scale=(0.0, 10.0), delete_old_features=True,
                       num_of_views=1, patch_segm=None):
        """One step of the 3D memory only -- frustum delete + update (VLN-POL:349-354) -- on GIVEN CLIP grid features `grid` (B*V, 576, 768)
        instead of the image tower's: brings the memory to a warm operating point that does not depend on 16-bit tower arithmetic
        (bench.py's golden parity point, tests).  The depth path is `build_inputs`'s."""
        ff, V = self.feature_fields, num_of_views
        B = ff.batch_size
        depth = observations["depth"].to(self.device, torch.float32)
        depth24 = self._depth24(depth, V, (0.0, 10.0))
        if delete_old_features:
            dfull = self.ops.preprocess_depth(depth[..., 0], *depth_scale).view(B, V, depth.shape[1], depth.shape[2])
            ff.delete_old_features_from_camera_frustum(dfull, agent_positions, agent_heading_angles, num_of_views=V)
        grid = grid.to(self.device)
        ff.update_feature_fields(depth24, grid.view(B, V, ff.P, -1), observations["rgb"].to(self.device), agent_positions, agent_heading_angles,
                                 num_of_views=V, patch_segm=patch_segm)

    PROMPT_HEAD = "<|user|>\n"                                    # VLN-POL:436

    def _prompt_text(self, b, instructions):
        return ("\nInstruction:\n" + instructions[b] + "\nHistory actions:\n" + "".join(self.feature_fields.history_actions[b])
                + "<|end|>\n<|assistant|>\nNext action:\n")

    def _prompt_ids(self, b, instructions, n_visual):
        """(ids in front of the visual prefix, ids behind it) = the reference's `inputs_embeds[:, :2]` / `[:, n_visual + 2:]` of the
        prompt tokenised with one "<image>" per visual token (VLN-POL:436-438, 456): `PromptTokenizer.split_prompt`."""
        return self.tokenizer.split_prompt(self.PROMPT_HEAD, n_visual, self._prompt_text(b, instructions))

    # CLIP block (0-based) behind which the llava tower may start; -1 = behind the whole CLIP tower (D3D_LLAVA_AFTER_CLIP_BLOCK)
    LLAVA_AFTER_CLIP_BLOCK = int(__import__("os").environ.get("D3D_LLAVA_AFTER_CLIP_BLOCK", "-1"))

    ASSEMBLE_KERNEL = True          # packed prompt rows by d3d_assemble_prompt (one pass); False: the PyTorch expressions below (the test's reference)

    def _assemble_packed(self, patch_feat, patch_pos, inst_tok, zone_tok, ni, nz, instructions, B, V):
        """Same rows as `_assemble_rows`, written once, in the LM's dtype, back to back (no per-environment tensors, no
        padding): one id upload + one embedding gather for all prompts, one add for all patch tokens, one row gather into
        the packed buffer.  Returns (x (Tp, hidden) with Tp = sum(lengths) rounded up to 256 rows, lengths)."""
        ff, dt = self.feature_fields, self.cfg.llava_dtype
        P = V * ff.P
        parts = [self._prompt_ids(b, instructions, P + ni[b] + nz[b]) for b in range(B)]        # (head ids, tail ids) per prompt
        emb_w = self.llm.embed_w
        if (self.ASSEMBLE_KERNEL and patch_feat.is_cuda and D.BACKEND["linear"] == "hip" and dt in (torch.bfloat16, torch.float16)
                and all(t.dtype == dt for t in (emb_w, patch_feat, patch_pos, inst_tok, zone_tok))):
            # one kernel writes every row from its source (csrc/tower_kernels.hip k_assemble_prompt): the host only builds the row table
            desc, lengths = [], []
            i_off, z_off = 0, 0
            for b in range(B):                                                                 # VLN-POL:456 row order
                h_ids, t_ids = parts[b]
                desc.append(np.concatenate([np.asarray(h_ids, np.int64), (1 << 28) + b * P + np.arange(P), (2 << 28) + i_off + np.arange(ni[b]),
                                            (3 << 28) + z_off + np.arange(nz[b]), np.asarray(t_ids, np.int64)]))
                lengths.append(len(h_ids) + P + ni[b] + nz[b] + len(t_ids))
                i_off, z_off = i_off + ni[b], z_off + nz[b]
            T = int(sum(lengths))
            Tp = (T + 255) // 256 * 256
            table = np.full(Tp, 7 << 28, np.int64)                                             # zero rows behind the last prompt
            table[:T] = np.concatenate(desc)
            x = D._hip.assemble_prompt(torch.from_numpy(table.astype(np.uint32).view(np.int32)).to(self.device), emb_w,
                                        patch_feat.reshape(B * P, -1).contiguous(), patch_pos.reshape(B * P, -1).contiguous(),
                                        inst_tok.contiguous(), zone_tok.contiguous(), Tp)
            self.last_lengths = lengths
            self.last_counts = dict(Ni=ni, Nz=nz)
            return x, lengths
        if (self.ASSEMBLE_KERNEL and patch_feat.is_cuda and D.BACKEND["linear"] == "hip" and dt == torch.float32
                and all(t.dtype == dt for t in (emb_w, patch_feat, patch_pos, inst_tok, zone_tok))):
            # float32 verification mode: the same row table through the float32 twin of the kernel
            desc, lengths = [], []
            i_off, z_off = 0, 0
            for b in range(B):
                h_ids, t_ids = parts[b]
                desc.append(np.concatenate([np.asarray(h_ids, np.int64), (1 << 28) + b * P + np.arange(P), (2 << 28) + i_off + np.arange(ni[b]),
                                            (3 << 28) + z_off + np.arange(nz[b]), np.asarray(t_ids, np.int64)]))
                lengths.append(len(h_ids) + P + ni[b] + nz[b] + len(t_ids))
                i_off, z_off = i_off + ni[b], z_off + nz[b]
            T = int(sum(lengths))
            Tp = (T + 255) // 256 * 256
            table = np.full(Tp, 7 << 28, np.int64)
            table[:T] = np.concatenate(desc)
            x = D._hip.assemble_prompt_f32(torch.from_numpy(table.astype(np.uint32).view(np.int32)).to(self.device), emb_w,
                                            patch_feat.reshape(B * P, -1).contiguous(), patch_pos.reshape(B * P, -1).contiguous(),
                                            inst_tok.contiguous(), zone_tok.contiguous(), Tp)
            self.last_lengths = lengths
            self.last_counts = dict(Ni=ni, Nz=nz)
            return x, lengths
        ids = torch.tensor([i for h, t in parts for i in h + t], device=self.device)
        emb = self.llm.embed_tokens(ids).to(dt)                                                # rows [0, E): head_0, tail_0, head_1, ...
        patch_tok = (patch_feat.reshape(B * P, -1).float() + patch_pos.reshape(B * P, -1).float()).to(dt)   # VLN-POL:448-453
        src = torch.cat([emb, patch_tok, inst_tok.to(dt), zone_tok.to(dt)], 0)
        E = emb.shape[0]
        o_patch, o_inst, o_zone = E, E + B * P, E + B * P + int(sum(ni))
        idx, lengths = [], []
        e_off, i_off, z_off = 0, 0, 0
        for b in range(B):                                                                     # VLN-POL:456 row order
            n_h, n_t = len(parts[b][0]), len(parts[b][1])
            idx.append(np.concatenate([e_off + np.arange(n_h), o_patch + b * P + np.arange(P), o_inst + i_off + np.arange(ni[b]),
                                       o_zone + z_off + np.arange(nz[b]), e_off + n_h + np.arange(n_t)]))
            lengths.append(n_h + P + ni[b] + nz[b] + n_t)
            e_off, i_off, z_off = e_off + n_h + n_t, i_off + ni[b], z_off + nz[b]
        T = int(sum(lengths))
        Tp = (T + 255) // 256 * 256
        x = torch.zeros((Tp, src.shape[1]), dtype=dt, device=self.device)
        torch.index_select(src, 0, torch.from_numpy(np.concatenate(idx)).to(self.device), out=x[:T])
        self.last_lengths = lengths
        self.last_counts = dict(Ni=ni, Nz=nz)
        return x, lengths

    def _assemble_rows(self, patch_feat, patch_pos, inst_tok, zone_tok, ni, nz, instructions, B, V, return_rows):
        ff = self.feature_fields
        patch_tok = patch_feat.float() + patch_pos                                             # VLN-POL:448-453
        patch_tok = patch_tok.view(B, V * ff.P, -1)
        # prompt (VLN-POL:436): ids 0..1 are kept in front of the visual prefix, the text follows it
        rows, lengths = [], []
        io, zo = np.concatenate([[0], np.cumsum(ni)]), np.concatenate([[0], np.cumsum(nz)])
        for b in range(B):
            h_ids, t_ids = self._prompt_ids(b, instructions, V * ff.P + ni[b] + nz[b])
            head_e = self.llm.embed_tokens(torch.tensor(h_ids, device=self.device)).float()
            te = self.llm.embed_tokens(torch.tensor(t_ids, device=self.device)).float()
            row = torch.cat([head_e, patch_tok[b], inst_tok[io[b]:io[b + 1]], zone_tok[zo[b]:zo[b + 1]], te], 0)   # VLN-POL:456
            rows.append(row)
            lengths.append(row.shape[0])
        self.last_lengths = lengths
        self.last_counts = dict(Ni=ni, Nz=nz)
        if return_rows:
            return rows, lengths
        S = max(lengths)
        embeds = torch.zeros((B, S, self.cfg.llm.hidden), dtype=self.cfg.llava_dtype, device=self.device)
        for b, r in enumerate(rows):
            embeds[b, :r.shape[0]] = r.to(self.cfg.llava_dtype)
        return embeds, torch.tensor(lengths, device=self.device)

    @torch.no_grad()
    def forward_logits(self, observations, instructions, agent_positions, agent_heading_angles, depth_scale=(0.0, 10.0),
                       gt_text=None, delete_old_features=True, num_of_views=1, is_train=False, patch_segm=None) -> torch.Tensor:
        if self.llm.packed_ok():
            x, lengths = self.build_inputs(observations, instructions, agent_positions, agent_heading_angles, depth_scale,
                                           delete_old_features, num_of_views, patch_segm, return_rows="packed")
            if self.keep_prompt:                             # (bench.py's parity block: the packed rows handed to Phi-3; the prefill does not write them)
                self.last_prompt = (x, list(lengths))
            return self.llm.prefill_logits_packed(x, lengths)
        rows, _ = self.build_inputs(observations, instructions, agent_positions, agent_heading_angles, depth_scale,
                                    delete_old_features, num_of_views, patch_segm, return_rows=True)
        return self.llm.prefill_logits_rows(rows)

    @torch.no_grad()
    def forward(self, observations, instructions, agent_positions, agent_heading_angles, depth_scale=(0.0, 10.0), gt_text=None,
                delete_old_features=True, num_of_views=1, is_train=False, patch_segm=None, max_new_tokens: int = 20):
        """Eval branch of VLN-POL:430-469: greedy text generation + per-row history update.  On the HIP backend the prompt
        is prefilled once (packed) and every further token is an 8-row pass over a KV cache (`Phi3Decoder.generate_packed`,
        SURVEY.md 8f-4); elsewhere (CPU tests) the prefix is recomputed per token.  The headline metric is `forward_logits`."""
        if is_train:
            raise NotImplementedError("training branch (loss over gt_text) is outside the hot path: SURVEY.md 8f-1")
        end_id = self.tokenizer.SPECIAL["<|end|>"] % self.cfg.llm.vocab
        if self.llm.packed_ok() and self.cfg.llm.kv_heads == self.cfg.llm.heads:
            x, lengths = self.build_inputs(observations, instructions, agent_positions, agent_heading_angles, depth_scale,
                                           delete_old_features, num_of_views, patch_segm, return_rows="packed")
            gen = self.llm.generate_packed(x, lengths, max_new_tokens=max_new_tokens, end_id=end_id)       # KV-cache decode
            return self._finish_generation(gen)
        embeds, lengths = self.build_inputs(observations, instructions, agent_positions, agent_heading_angles, depth_scale,
                                            delete_old_features, num_of_views, patch_segm)
        B = embeds.shape[0]
        gen = [[] for _ in range(B)]
        done = [False] * B
        end_id = self.tokenizer.SPECIAL["<|end|>"] % self.cfg.llm.vocab
        for _ in range(max_new_tokens):
            nxt = self.llm.prefill_logits(embeds, lengths).argmax(-1)
            S = int(lengths.max()) + 1
            if S > embeds.shape[1]:
                embeds = torch.cat([embeds, torch.zeros((B, S - embeds.shape[1], embeds.shape[2]), dtype=embeds.dtype, device=self.device)], 1)
            e = self.llm.embed_tokens(nxt)
            for b in range(B):
                if done[b]:
                    continue
                gen[b].append(int(nxt[b]))
                done[b] = int(nxt[b]) == end_id
                embeds[b, int(lengths[b])] = e[b]
            lengths = lengths + torch.tensor([0 if d and len(g) and g[-1] != end_id else 1 for d, g in zip(done, gen)], device=self.device)
            if all(done):
                break
        return self._finish_generation(gen)

    def _finish_generation(self, gen):
        texts = []
        for b in range(len(gen)):
            t = self.tokenizer.decode(gen[b])
            cut = t.find("<|end|>")
            t = t[:cut] if cut >= 0 else t
            texts.append(t)
            self.feature_fields.history_actions[b].pop(0)                                       # VLN-POL:466-468
            self.feature_fields.history_actions[b].append(t + "\n")
        return texts

    # ---- teacher text (VLN-POL:294-327; called by the trainer in 'train' mode, VLN-TR:664, 676) ----------------------------
    def get_gt_text(self, target_angles, target_distances, stop_actions):
        """(angle rad CCW, distance m, stop flag) per environment -> the action sentence the LM is trained to emit, and the side
        effect the reference has: a turn of >= 4 steps is capped at 4 and the REMAINING turn + the move are parked in
        `feature_fields.keep_target_waypoint[b]` for the next step.  A sentence whose first 17 characters repeat the history the way the
        reference checks it (entries -2 and -4 against this sentence, entry -3 against the LAST environment's sentence -- the
        reference's own indexing, reproduced) becomes "error.<|end|>".  Pinned by tests/golden/g20_gt_text.json."""
        ff = self.feature_fields
        step_deg, step_m, cap = 15, 0.25, 4
        degs = [round(np.degrees(a)) for a in target_angles]
        texts = ["" for _ in degs]
        for b, (deg, dist) in enumerate(zip(degs, target_distances)):
            if stop_actions[b] == True:      # noqa: E712  (the reference compares with ==)
                texts[b] = "stop.<|end|>"
            else:
                turns = round(deg / step_deg)
                move = " move " + str(round(dist / step_m)) + " steps.<|end|>"
                left = "turn left " + str(turns) + " steps," + move
                right = "turn right " + str(round((360 - deg) / step_deg)) + " steps," + move
                if cap <= turns < 360 // step_deg:
                    go_left = turns < 180 // step_deg
                    texts[b] = left if go_left else right
                    rest = deg - cap * step_deg if go_left else deg + cap * step_deg
                    ff.keep_target_waypoint[b] = [(np.radians(rest) + math.pi * 2) % (math.pi * 2), dist]
                else:
                    texts[b] = left if turns < cap else right
                    ff.keep_target_waypoint[b] = None
            n = len("turn left 4 steps")
            h = ff.history_actions[b]
            if h[-2][:n] == texts[b][:n] and h[-4][:n] == texts[b][:n] and h[-3][:n] == texts[-1][:n]:
                texts[b] = "error.<|end|>"
        return texts

    def get_candidate_waypoints(self, waypoint_predictor=None, observations=None):
        """VLN-POL:188-292: the waypoint predictor over the DDPPO depth encoder's 12-view embedding -- outside the hot path
        (SURVEY.md section 2).  The attributes that function reads exist on this module (`depth_encoder`, `space_pool_depth`,
        `pano_angle_fts`, `pano_img_idxes`), so on a machine with Habitat the reference's own function can be bound unchanged:
        `net.get_candidate_waypoints = types.MethodType(RefNet.get_candidate_waypoints, net)` with
        `Dynam3D_VLN(depth_encoder=VlnResnetDepthEncoder(...))`."""
        raise NotImplementedError("get_candidate_waypoints is outside the hot path (SURVEY.md section 2); bind the reference's function, see the docstring")

    # ---- a18 (VLN-POL:472-506) ----------------------------------------------------------------------------------
    @staticmethod
    def convert_text_to_action(generated_text: Sequence[str]):
        """'turn left/right k steps, move m steps.' -> (angle rad, distance m); stop / error / malformed -> -100.
        Where the reference raises (unparsable integers, 'move' without a turn: NameError/ValueError) this
        returns -100."""
        angle_per_step, distance_per_step, max_turn_steps = 15, 0.25, 4
        out = []
        for text in generated_text:
            try:
                if "stop" in text or "error" in text:
                    out.append(-100)
                    continue
                angle, distance = 0.0, 0.0
                side = "left" if "left" in text else ("right" if "right" in text else None)
                if side is None:
                    if "move" in text:
                        out.append(-100)       # reference: unbound `start`/`end` -> NameError
                        continue
                    out.append((angle, distance))
                    continue
                start = text.find(side) + len(side)
                end = text.find("steps,")
                if end == -1:
                    out.append(-100)
                    continue
                k = int(text[start:end])
                turn = math.radians(min(max_turn_steps, k) * angle_per_step)
                angle = turn if side == "left" else math.pi * 2.0 - turn
                if "move" in text and k < max_turn_steps:
                    s2 = text.find("move") + len("move")
                    e2 = text.find("steps.")
                    distance = int(text[s2:e2]) * distance_per_step
                out.append((angle, distance))
            except ValueError:
                out.append(-100)
        return out
