"""The three dense towers of the step (the only true dense contractions, all on MFMA):

  * `ClipVisionTower`  -- OpenAI CLIP ViT-L/14@336 image encoder returning (cls, 576 x 768 patch features)
                          (reference: encoders/clip/model.py:202-238 `VisionTransformer`, driven by
                          encoders/resnet_encoders.py:273-284 `CLIPEncoder.forward`), fp16 like the reference.
  * `LlavaVisionTower` -- llava-phi-3-mini's HF CLIP vision model, hidden state of layer -2 without CLS,
                          + 2-layer GELU projector -> (B,576,3072)   (VLN-POL:448-452), bf16.
  * `Phi3Decoder`      -- Phi-3-mini decoder stack prefill over `inputs_embeds` -> logits at the last prompt
                          position (VLN-POL:463), bf16.

Weights arrive under the reference checkpoints' own key names (OpenAI CLIP `visual.*`, HF llava
`vision_tower.*` / `multi_modal_projector.*` / `language_model.*`) and are re-laid-out once at load
(fused QKV, contiguous [out,in]) for the GEMM kernels.  All matmuls/attention/norms go through
`dense_ops`, which dispatches to the hand-written HIP kernels where they exist.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import nn

from . import dense_ops as D
from .modules import ParamTree, install_param, own_copy
from .profiling import TIMER

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class VitConfig:
    image: int = 336
    patch: int = 14
    width: int = 1024
    layers: int = 24
    heads: int = 16
    mlp: int = 4096
    out_dim: int = 768           # CLIP `proj`
    proj_dim: int = 3072         # llava projector width

    @property
    def grid(self):
        return self.image // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1


@dataclass
class Phi3Config:
    vocab: int = 32064
    hidden: int = 3072
    layers: int = 32
    heads: int = 32
    kv_heads: int = 32
    mlp: int = 8192
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_pos: int = 4096

    @property
    def head_dim(self):
        return self.hidden // self.heads


# ---------------------------------------------------------------------------------------------------
# parameter specs (names == the reference checkpoints' keys)
# ---------------------------------------------------------------------------------------------------
def clip_param_spec(c: VitConfig):
    s = [("visual.conv1.weight", (c.width, 3, c.patch, c.patch)), ("visual.class_embedding", (c.width,)),
         ("visual.positional_embedding", (c.tokens, c.width)), ("visual.ln_pre.weight", (c.width,)), ("visual.ln_pre.bias", (c.width,))]
    for i in range(c.layers):
        p = f"visual.transformer.resblocks.{i}"
        s += [(p + ".attn.in_proj_weight", (3 * c.width, c.width)), (p + ".attn.in_proj_bias", (3 * c.width,)),
              (p + ".attn.out_proj.weight", (c.width, c.width)), (p + ".attn.out_proj.bias", (c.width,)),
              (p + ".ln_1.weight", (c.width,)), (p + ".ln_1.bias", (c.width,)),
              (p + ".mlp.c_fc.weight", (c.mlp, c.width)), (p + ".mlp.c_fc.bias", (c.mlp,)),
              (p + ".mlp.c_proj.weight", (c.width, c.mlp)), (p + ".mlp.c_proj.bias", (c.width,)),
              (p + ".ln_2.weight", (c.width,)), (p + ".ln_2.bias", (c.width,))]
    s += [("visual.ln_post.weight", (c.width,)), ("visual.ln_post.bias", (c.width,)), ("visual.proj", (c.width, c.out_dim))]
    return s


def llava_vision_param_spec(c: VitConfig):
    v = "vision_tower.vision_model"
    s = [(v + ".embeddings.class_embedding", (c.width,)), (v + ".embeddings.patch_embedding.weight", (c.width, 3, c.patch, c.patch)),
         (v + ".embeddings.position_embedding.weight", (c.tokens, c.width)),
         (v + ".pre_layrnorm.weight", (c.width,)), (v + ".pre_layrnorm.bias", (c.width,))]
    for i in range(c.layers):
        p = f"{v}.encoder.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s += [(f"{p}.self_attn.{n}.weight", (c.width, c.width)), (f"{p}.self_attn.{n}.bias", (c.width,))]
        s += [(p + ".layer_norm1.weight", (c.width,)), (p + ".layer_norm1.bias", (c.width,)),
              (p + ".mlp.fc1.weight", (c.mlp, c.width)), (p + ".mlp.fc1.bias", (c.mlp,)),
              (p + ".mlp.fc2.weight", (c.width, c.mlp)), (p + ".mlp.fc2.bias", (c.width,)),
              (p + ".layer_norm2.weight", (c.width,)), (p + ".layer_norm2.bias", (c.width,))]
    s += [("multi_modal_projector.linear_1.weight", (c.proj_dim, c.width)), ("multi_modal_projector.linear_1.bias", (c.proj_dim,)),
          ("multi_modal_projector.linear_2.weight", (c.proj_dim, c.proj_dim)), ("multi_modal_projector.linear_2.bias", (c.proj_dim,))]
    return s


def phi3_param_spec(c: Phi3Config):
    m = "language_model.model"
    s = [(m + ".embed_tokens.weight", (c.vocab, c.hidden))]
    qkv = (c.heads + 2 * c.kv_heads) * c.head_dim
    for i in range(c.layers):
        p = f"{m}.layers.{i}"
        s += [(p + ".self_attn.qkv_proj.weight", (qkv, c.hidden)), (p + ".self_attn.o_proj.weight", (c.hidden, c.heads * c.head_dim)),
              (p + ".mlp.gate_up_proj.weight", (2 * c.mlp, c.hidden)), (p + ".mlp.down_proj.weight", (c.hidden, c.mlp)),
              (p + ".input_layernorm.weight", (c.hidden,)), (p + ".post_attention_layernorm.weight", (c.hidden,))]
    s += [(m + ".norm.weight", (c.hidden,)), ("language_model.lm_head.weight", (c.vocab, c.hidden))]
    return s


# ---------------------------------------------------------------------------------------------------
# shared ViT trunk (pre-LN blocks, QuickGELU MLP): both CLIP flavours are this network
# ---------------------------------------------------------------------------------------------------
class _VitTrunk:
    def __init__(self, cfg: VitConfig, dtype, device):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.blocks = []
        self.patch_w = self.cls = self.pos = None
        self.ln_pre = None

    def _t(self, x):
        return x.detach().to(self.device, self.dtype).contiguous()

    def _f(self, x):   # norm gains/biases that are consumed in float32
        return x.detach().to(self.device, torch.float32).contiguous()

    def _fq(self, x):  # float32 copies of parameters the REFERENCE holds in the tower's 16-bit dtype (HF `model.to(bfloat16)` casts the
        return x.detach().to(self.device, self.dtype).to(torch.float32).contiguous()     # norm gains too): same values, fp32 container

    def _patch_weight(self, w4):
        """conv1.weight (W,3,P,P) -> (W, Kp): the GEMM's [out, in] operand, K zero-padded to a multiple of 64 (588 -> 640)."""
        w = w4.reshape(w4.shape[0], -1)
        Kp = (w.shape[1] + 63) // 64 * 64
        out = torch.zeros((w.shape[0], Kp), dtype=self.dtype, device=self.device)
        out[:, :w.shape[1]] = w.to(self.device, self.dtype)
        return out

    def embed(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels (B,3,336,336) normalised float32 -> (B,577,width) after patch embedding, cls/pos and ln_pre (clip/model.py:222-228)."""
        return D.vit_embed(pixels.float(), self.patch_w, self.cls, self.pos, self.ln_pre[0], self.ln_pre[1], self.cfg.patch, 1e-5)

    def block(self, i: int, x: torch.Tensor) -> torch.Tensor:
        """Residual block i on x (B, L, W) (clip/model.py:166-187 `ResidualAttentionBlock` == HF `CLIPEncoderLayer`); also the entry point of
        the per-layer teacher-forced parity tests."""
        c, blk = self.cfg, self.blocks[i]
        B, L, W = x.shape
        h = D.layer_norm(x, blk["ln1_w"], blk["ln1_b"], 1e-5)
        qkv = D.linear(h.view(B * L, W), blk["qkv_w"], blk["qkv_b"]).view(B, L, 3 * c.heads, W // c.heads)
        a = D.attention_qkv(qkv, c.heads, causal=False)                                    # (B,L,H,hd)
        x = D.linear(a.reshape(B * L, W), blk["out_w"], blk["out_b"], residual=x.view(B * L, W)).view(B, L, W)
        h = D.layer_norm(x, blk["ln2_w"], blk["ln2_b"], 1e-5)
        h = D.linear(h.view(B * L, W), blk["fc1_w"], blk["fc1_b"], act="quick_gelu")
        return D.linear(h, blk["fc2_w"], blk["fc2_b"], residual=x.view(B * L, W)).view(B, L, W)

    # (layer index, callable) set by the policy: called once, right after that block has been QUEUED -- it records the event behind which the
    # other vision tower starts (policy.build_inputs), so that the two towers overlap for the last blocks of this one
    after_block = None

    def run_blocks(self, x: torch.Tensor, n_layers: int) -> torch.Tensor:
        hook = self.after_block
        for i in range(n_layers):
            x = self.block(i, x)
            if hook is not None and i == hook[0]:
                hook[1]()
        return x


class ClipVisionTower(_VitTrunk):
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: VitConfig = VitConfig(), dtype=torch.float16, device="cuda"):
        super().__init__(cfg, dtype, device)
        self.load(sd)

    def load(self, sd: Dict[str, torch.Tensor]):
        """(Re-)derive the kernel layouts from the `visual.*` tensors; tensors that already have the tower's dtype / device / a contiguous
        [out, in] layout are ALIASED, not copied (modules.py)."""
        cfg = self.cfg
        t, f = self._t, self._f
        self.blocks = []
        self.patch_w = self._patch_weight(sd["visual.conv1.weight"])
        self.cls, self.pos = t(sd["visual.class_embedding"]), t(sd["visual.positional_embedding"])
        self.ln_pre = (f(sd["visual.ln_pre.weight"]), f(sd["visual.ln_pre.bias"]))
        self.ln_post = (f(sd["visual.ln_post.weight"]), f(sd["visual.ln_post.bias"]))
        self.proj_w = t(sd["visual.proj"].t())          # stored [out,in] for D.linear
        for i in range(cfg.layers):
            p = f"visual.transformer.resblocks.{i}"
            self.blocks.append(dict(
                qkv_w=t(sd[p + ".attn.in_proj_weight"]), qkv_b=t(sd[p + ".attn.in_proj_bias"]),
                out_w=t(sd[p + ".attn.out_proj.weight"]), out_b=t(sd[p + ".attn.out_proj.bias"]),
                ln1_w=f(sd[p + ".ln_1.weight"]), ln1_b=f(sd[p + ".ln_1.bias"]), ln2_w=f(sd[p + ".ln_2.weight"]), ln2_b=f(sd[p + ".ln_2.bias"]),
                fc1_w=t(sd[p + ".mlp.c_fc.weight"]), fc1_b=t(sd[p + ".mlp.c_fc.bias"]),
                fc2_w=t(sd[p + ".mlp.c_proj.weight"]), fc2_b=t(sd[p + ".mlp.c_proj.bias"])))

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor):
        """-> (cls (B,768), patches (B,576,768)) in tower dtype (clip/model.py:219-238)."""
        c = self.cfg
        x = self.run_blocks(self.embed(pixels), c.layers)
        x = D.layer_norm(x, self.ln_post[0], self.ln_post[1], 1e-5)          # ln_post on ALL tokens
        B, L, W = x.shape
        y = D.linear(x.view(B * L, W), self.proj_w, None).view(B, L, c.out_dim)
        return y[:, 0], y[:, 1:]


class LlavaVisionTower(_VitTrunk):
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: VitConfig = VitConfig(), dtype=torch.bfloat16, device="cuda",
                 feature_layer: int = -2):
        super().__init__(cfg, dtype, device)
        self.feature_layer = feature_layer
        self.load(sd)

    def load(self, sd: Dict[str, torch.Tensor]):
        cfg, feature_layer = self.cfg, self.feature_layer
        t, f = self._t, self._fq                         # HF `torch_dtype=bfloat16` casts the LayerNorm parameters as well
        self.blocks = []
        v = "vision_tower.vision_model"
        self.patch_w = self._patch_weight(sd[v + ".embeddings.patch_embedding.weight"])
        self.cls, self.pos = t(sd[v + ".embeddings.class_embedding"]), t(sd[v + ".embeddings.position_embedding.weight"])
        self.ln_pre = (f(sd[v + ".pre_layrnorm.weight"]), f(sd[v + ".pre_layrnorm.bias"]))
        self.n_run = cfg.layers + 1 + feature_layer          # hidden_states[-2] == output of layer (L-1)
        for i in range(cfg.layers):
            p = f"{v}.encoder.layers.{i}.self_attn"
            q = f"{v}.encoder.layers.{i}"
            self.blocks.append(dict(
                qkv_w=t(torch.cat([sd[p + ".q_proj.weight"], sd[p + ".k_proj.weight"], sd[p + ".v_proj.weight"]], 0)),
                qkv_b=t(torch.cat([sd[p + ".q_proj.bias"], sd[p + ".k_proj.bias"], sd[p + ".v_proj.bias"]], 0)),
                out_w=t(sd[p + ".out_proj.weight"]), out_b=t(sd[p + ".out_proj.bias"]),
                ln1_w=f(sd[q + ".layer_norm1.weight"]), ln1_b=f(sd[q + ".layer_norm1.bias"]),
                ln2_w=f(sd[q + ".layer_norm2.weight"]), ln2_b=f(sd[q + ".layer_norm2.bias"]),
                fc1_w=t(sd[q + ".mlp.fc1.weight"]), fc1_b=t(sd[q + ".mlp.fc1.bias"]),
                fc2_w=t(sd[q + ".mlp.fc2.weight"]), fc2_b=t(sd[q + ".mlp.fc2.bias"])))
        self.p1 = (t(sd["multi_modal_projector.linear_1.weight"]), t(sd["multi_modal_projector.linear_1.bias"]))
        self.p2 = (t(sd["multi_modal_projector.linear_2.weight"]), t(sd["multi_modal_projector.linear_2.bias"]))

    @torch.no_grad()
    def forward(self, pixels: torch.Tensor) -> torch.Tensor:
        """-> (B,576,3072): llava.get_image_features(layer -2, 'default') (VLN-POL:448-452)."""
        c = self.cfg
        x = self.run_blocks(self.embed(pixels), self.n_run)[:, 1:]
        B, L, W = x.shape
        h = D.linear(x.reshape(B * L, W), self.p1[0], self.p1[1], act="gelu")
        return D.linear(h, self.p2[0], self.p2[1]).view(B, L, c.proj_dim)


# ---------------------------------------------------------------------------------------------------
# Phi-3-mini decoder prefill
# ---------------------------------------------------------------------------------------------------
class Phi3Decoder:
    def __init__(self, sd: Dict[str, torch.Tensor], cfg: Phi3Config = Phi3Config(), dtype=torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self._rope_cache = {}
        self.load(sd)

    def load(self, sd: Dict[str, torch.Tensor]):
        """(Re-)derive the kernel layouts from the `language_model.*` tensors (aliasing where the layout is the reference's)."""
        cfg, dtype = self.cfg, self.dtype
        self._w_arrays = None                                                            # decode runner's weight pointer tables
        t = lambda x: x.detach().to(self.device, dtype).contiguous()
        f = lambda x: x.detach().to(self.device, dtype).to(torch.float32).contiguous()   # RMSNorm gains: values of the LM's dtype (HF casts them)
        m = "language_model.model"
        self.embed_w = t(sd[m + ".embed_tokens.weight"])
        # gate/up rows interleaved per 16 for the fused SwiGLU GEMM epilogue (only when the HIP GEMM is active)
        self.interleave_gu = D.BACKEND["linear"] == "hip" and cfg.mlp % 16 == 0 and dtype != torch.float32     # (float32 verification mode: plain rows)
        self.layers = []
        for i in range(cfg.layers):
            p = f"{m}.layers.{i}"
            gu = t(sd[p + ".mlp.gate_up_proj.weight"])
            if self.interleave_gu:
                from .hip_dense import interleave_gate_up
                gu = interleave_gate_up(gu)
            self.layers.append(dict(qkv_w=t(sd[p + ".self_attn.qkv_proj.weight"]), o_w=t(sd[p + ".self_attn.o_proj.weight"]),
                                    gu_w=gu, down_w=t(sd[p + ".mlp.down_proj.weight"]),
                                    n1=f(sd[p + ".input_layernorm.weight"]), n2=f(sd[p + ".post_attention_layernorm.weight"])))
        self.norm_w = f(sd[m + ".norm.weight"])
        self.lm_head_w = t(sd["language_model.lm_head.weight"])

    PRUNE_LAST_LAYER = True            # prefill_logits_packed: the last layer's o_proj / MLP on the B last rows only
    MAX_DECODE_ROWS = 16          # k_gemm_skinny: M <= 16
    MAX_DECODE_KEYS = 4096 + 64   # k_decode_attn keeps one score per key in LDS
    SLIDING_WINDOW = 2047         # Phi-3-mini-4k-instruct config.json: every layer attends to the last 2047 keys only

    def _check_lengths(self, lens, max_new_tokens: int = 0):
        """Prompts the kernels do not model.  The PREFILL masks Phi-3-mini's sliding attention window (every layer attends to the last
        2047 keys only) inside the flash kernel (`d3d_flash_attention_v3(window=...)`); the KV-cache decode kernel does not, so generation
        is limited to prompt + new tokens <= 2047.  The reference's prompts are 2 + 576 + Ni + Nz + text ~ 0.8-1.4 k tokens."""
        longest = max(lens) + max_new_tokens
        if max_new_tokens and longest > self.SLIDING_WINDOW:
            raise ValueError(f"prompt of {max(lens)} tokens (+{max_new_tokens} generated) exceeds Phi-3-mini's sliding window of {self.SLIDING_WINDOW} keys, "
                             "which the KV-cache decode kernel does not model")
        if longest > self.cfg.max_pos or longest > self.MAX_DECODE_KEYS:
            raise ValueError(f"sequence of {longest} tokens exceeds max_position_embeddings = {self.cfg.max_pos}")

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return self.embed_w.index_select(0, ids.reshape(-1)).view(*ids.shape, self.cfg.hidden)

    def _rope(self, S: int):
        n = S
        S = (S + 2047) // 2048 * 2048              # one cached table serves every prompt length up to S; callers get its first n rows
        if S not in self._rope_cache:
            c = self.cfg
            inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32, device=self.device) / c.head_dim))
            ang = torch.arange(S, dtype=torch.float32, device=self.device)[:, None] * inv[None]
            # HF Phi3RotaryEmbedding returns cos / sin cast to the activations' dtype: float32 containers of those values
            self._rope_cache[S] = (ang.cos().to(self.dtype).float().contiguous(), ang.sin().to(self.dtype).float().contiguous())   # (S, hd/2)
        cos, sin = self._rope_cache[S]
        return cos[:n], sin[:n]

    @torch.no_grad()
    def packed_ok(self) -> bool:
        c = self.cfg
        return bool(D.packed_ok(self.dtype, c.head_dim) and c.kv_heads == c.heads and self.device.type == "cuda")

    def prefill_logits_rows(self, rows) -> torch.Tensor:
        """rows: list of B tensors (S_b, hidden) -- the per-environment prompts.  With the HIP backend the batch is PACKED
        (no padding: sum(S_b) tokens, rounded up to a multiple of 256 rows so every GEMM is whole 256-row tiles) and
        attention / RoPE work off per-sequence offsets; otherwise it is right-padded and sent to `prefill_logits`."""
        c = self.cfg
        lens = [int(r.shape[0]) for r in rows]
        B = len(rows)
        if not (self.packed_ok() and rows[0].is_cuda):
            S = max(lens)
            emb = torch.zeros((B, S, c.hidden), dtype=self.dtype, device=self.device)
            for b, r in enumerate(rows):
                emb[b, :lens[b]] = r.to(self.dtype)
            return self.prefill_logits(emb, torch.tensor(lens, device=self.device))
        T = sum(lens)
        Tp = (T + 255) // 256 * 256
        with TIMER.range("prefill.pack"):
            x = torch.zeros((Tp, c.hidden), dtype=self.dtype, device=self.device)
            torch.cat([r.to(self.dtype) for r in rows], 0, out=x[:T])
        return self.prefill_logits_packed(x, lens)

    def packed_context(self, lens, Tp: int):
        """Per-call tables of the packed prefill: sequence offsets (device + host), per-row rotary positions, cos / sin, last rows."""
        cu_h = [0]
        for n in lens:
            cu_h.append(cu_h[-1] + n)
        cu = torch.tensor(cu_h, dtype=torch.int32, device=self.device)
        pos_h = torch.zeros(Tp, dtype=torch.int32)
        for b, n in enumerate(lens):
            pos_h[cu_h[b]:cu_h[b + 1]] = torch.arange(n, dtype=torch.int32)
        pos = pos_h.to(self.device, non_blocking=True)
        max_len = max(lens)
        cos, sin = self._rope(max_len)
        # one attention-output buffer for all layers (o_proj consumes it before the next layer's attention writes it); its padding rows
        # are zeroed here, once per prompt batch, instead of once per layer
        attn_out = torch.empty((Tp, self.cfg.heads, self.cfg.head_dim), dtype=self.dtype, device=self.device)
        if cu_h[-1] < Tp:
            attn_out[cu_h[-1]:].zero_()
        # D3D_ATTN_SCHED=1: one query block per attention workgroup, heaviest first (hip_dense.attention_schedule; bit-identical results).
        # Off by default: measured 1 ms SLOWER per step than the paired, head-adjacent launch (profiles/r05_attention_schedule_ab.txt)
        sched = D.attention_schedule(lens, self.cfg.heads, self.device) if self.ATTN_SCHED else None
        return dict(cu=cu, cu_h=cu_h, pos=pos, cos=cos, sin=sin, max_len=max_len, last_rows=(cu[1:] - 1).long(), B=len(lens), Tp=Tp, attn_out=attn_out,
                    sched=sched)

    ATTN_SCHED = os.environ.get("D3D_ATTN_SCHED", "0") == "1"
    TIME_EVERY = max(1, int(os.environ.get("D3D_BENCH_TIME_EVERY", "4")))

    # rotate the queries inside the attention kernel (d3d_flash_attention_v3_rope_q) when the backend can; D3D_FUSE_ROPE_Q=0: in place, with the keys
    FUSE_ROPE_Q = os.environ.get("D3D_FUSE_ROPE_Q", "1") != "0"

    @torch.no_grad()
    def layer_packed(self, li: int, x: torch.Tensor, ctx, keep_kv: Optional[list] = None, prune: bool = False) -> torch.Tensor:
        """One decoder layer over the packed rows x (Tp, hidden) -> (Tp, hidden) (HF `Phi3DecoderLayer.forward`: RMSNorm, fused QKV,
        rotary, causal attention per sequence, o_proj + residual, RMSNorm, gate_up + SwiGLU, down_proj + residual).  `prune`: behind
        the attention only each prompt's LAST row is evaluated -> (B, hidden) (see `prefill_logits_packed`).  Also the entry point of
        the per-layer teacher-forced parity tests (tests/test_gpu_depth_parity.py)."""
        c, L = self.cfg, self.layers[li]
        Tp, Ht = x.shape[0], c.heads + 2 * c.kv_heads
        h = D.rms_norm(x, L["n1"], c.rms_eps)
        qkv = D.linear(h, L["qkv_w"], None)
        fuse_q = self.FUSE_ROPE_Q and D.can_fuse_rope_q(self.dtype)
        if fuse_q:
            # keys rotated in place (the KV cache keeps rotated keys), queries rotated inside the attention kernel: half of k_rope's traffic
            D.rope_packed_(qkv[:, c.heads * c.head_dim:], c.kv_heads, c.head_dim, ctx["cos"], ctx["sin"], ctx["pos"])
        else:
            D.rope_packed_(qkv, c.heads + c.kv_heads, c.head_dim, ctx["cos"], ctx["sin"], ctx["pos"])
        if keep_kv is not None:
            keep_kv.append(qkv)                                        # (only its k / v heads are read afterwards)
        a = D.attention_packed(qkv.view(Tp, Ht, c.head_dim), c.heads, True, ctx["cu"], ctx["B"], ctx["max_len"], n_valid=ctx["cu_h"][-1],
                               window=self.SLIDING_WINDOW if ctx["max_len"] > self.SLIDING_WINDOW else 0, out=ctx.get("attn_out"),
                               rope_q=(ctx["cos"], ctx["sin"]) if fuse_q else None, sched=ctx.get("sched"))
        a = a.view(Tp, c.heads * c.head_dim)
        if prune:
            a, x = a[ctx["last_rows"]].contiguous(), x[ctx["last_rows"]].contiguous()
        x = D.linear(a, L["o_w"], None, residual=x)
        h = D.rms_norm(x, L["n2"], c.rms_eps)
        if h.shape[0] == Tp and li % self.TIME_EVERY == 0:
            # bench.py's roofline launch (full-row GEMMs only), HIP-event timed in every TIME_EVERY-th layer: an event pair costs ~6 us on
            # the stream and the bracket below as much again -- timing all 31 full-row launches of a step put 0.4 ms of instrumentation
            # into the step it measures (D3D_BENCH_TIME_EVERY=1: every layer, as rounds 1-4 did)
            with TIMER.range("phi3.gate_up_proj", rows=ctx["cu_h"][-1], rows_launched=Tp):
                act = D.linear_swiglu(h, L["gu_w"], self.interleave_gu)
            with TIMER.range("phi3.event_pair_overhead"):                   # an EMPTY bracket right behind it: what two event records
                pass                                                        # cost on this stream at this point (bench.py subtracts it)
        else:
            act = D.linear_swiglu(h, L["gu_w"], self.interleave_gu)
        return D.linear(act, L["down_w"], None, residual=x)

    @torch.no_grad()
    def final_logits(self, last: torch.Tensor) -> torch.Tensor:
        """(B, hidden) last-position rows -> (B, vocab) float32: final RMSNorm + lm_head."""
        return D.linear(D.rms_norm(last, self.norm_w, self.cfg.rms_eps), self.lm_head_w, None).float()

    @torch.no_grad()
    def prefill_logits_packed(self, x: torch.Tensor, lens, keep_kv: Optional[list] = None) -> torch.Tensor:
        """x (Tp, hidden) in the LM's dtype: the B prompts back to back (lens[b] rows each), zero rows up to Tp (a multiple
        of 256).  -> logits (B, vocab) float32 at each prompt's last position.  `keep_kv`: a list that receives every layer's
        fused-QKV buffer after RoPE -- the prompt part of the KV cache, read in place by `generate_packed`."""
        B, Tp = len(lens), x.shape[0]
        self._check_lengths(lens)
        ctx = self.packed_context(lens, Tp)
        n = len(self.layers)
        for li in range(n):
            # Behind the last layer's attention every operation is row-wise and only each prompt's LAST row is read (the logits of the
            # next token; the layer's keys / values -- all rows -- are already in `qkv`): o_proj, the MLP, the final norm and the
            # lm_head run on B rows instead of Tp.  Same arithmetic per row, 1/32 of the stack's o_proj + MLP GEMM time saved.
            x = self.layer_packed(li, x, ctx, keep_kv, prune=(li == n - 1 and self.PRUNE_LAST_LAYER))
        self.last_packed_rows = Tp
        self._last_cu = ctx["cu"]
        return self.final_logits(x if x.shape[0] == B else x[ctx["last_rows"]])

    @torch.no_grad()
    def generate_packed(self, x: torch.Tensor, lens, max_new_tokens: int = 20, end_id: Optional[int] = None, forced=None,
                        return_logits: bool = False):
        """Greedy generation with a KV cache (`llava.generate(..., max_new_tokens=20, do_sample=False)`, VLN-POL:463).
        Prefill as `prefill_logits_packed`; every later token costs one 8-row pass: the prompt keys/values are read in place
        from the prefill's post-RoPE QKV buffers, the generated tokens' from a (B, max_new, H, hd) side cache per layer
        (`d3d_decode_attention`).  Returns the token ids per sequence (up to and including `end_id`), and with
        `return_logits` the (steps, B, vocab) float32 logits.  `forced` (steps x B ids) replaces the argmax (teacher forcing)."""
        c = self.cfg
        B = len(lens)
        if B > self.MAX_DECODE_ROWS:
            # the weight-streaming decode kernels take at most 16 rows: larger batches are generated in groups of 16 sequences
            out_tok, out_logits, cu = [], [], [0]
            for n in lens:
                cu.append(cu[-1] + n)
            for i0 in range(0, B, self.MAX_DECODE_ROWS):
                i1 = min(B, i0 + self.MAX_DECODE_ROWS)
                T = cu[i1] - cu[i0]
                xs = torch.zeros(((T + 255) // 256 * 256, x.shape[1]), dtype=x.dtype, device=x.device)
                xs[:T] = x[cu[i0]:cu[i1]]
                r = self.generate_packed(xs, lens[i0:i1], max_new_tokens, end_id, None if forced is None else [f[i0:i1] for f in forced], return_logits)
                out_tok += r[0] if return_logits else r
                if return_logits:
                    out_logits.append(r[1])
            if return_logits:
                steps = max(l.shape[0] for l in out_logits)          # groups may stop at different steps (end_id): pad with NaN rows
                pad = [torch.cat([l, torch.full((steps - l.shape[0],) + tuple(l.shape[1:]), float("nan"), device=l.device)], 0) for l in out_logits]
                return out_tok, torch.cat(pad, 1)
            return out_tok
        self._check_lengths(lens, max_new_tokens)
        kv = []
        logits = self.prefill_logits_packed(x, lens, keep_kv=kv)
        cu = self._last_cu
        H, hd = c.heads, c.head_dim
        Tmax = max(1, max_new_tokens - 1)
        side = torch.empty((c.layers, 2, B, Tmax, H, hd), dtype=self.dtype, device=self.device)
        cos, sin = self._rope(max(lens) + max_new_tokens + 1)
        lens_d = torch.tensor(lens, dtype=torch.int32, device=self.device)
        all_logits, toks, run = [], [], None
        done = torch.zeros(B, dtype=torch.bool, device=self.device)
        n_steps = 0
        for i in range(max_new_tokens):
            if return_logits:
                all_logits.append(logits)
            nxt = logits.argmax(-1) if forced is None else torch.as_tensor(forced[i], device=self.device)
            toks.append(nxt)
            n_steps = i + 1
            if i == max_new_tokens - 1:
                break
            if end_id is not None:
                done |= nxt == end_id
                # the tokens stay on the device; the host only looks every 4th token whether every sequence has finished
                # (a per-token .tolist() stalled the GPU for the ~0.5 ms the next token takes to issue)
                if i % 4 == 3 and bool(done.all()):
                    break
            xt = self.embed_tokens(nxt.long()).to(self.dtype).contiguous()        # (B, hidden): generated token i at position lens + i
            pos = (lens_d + i).contiguous()
            if run is None:
                run = self._decode_runner(B, kv, cu, side, cos, sin, max(lens))
            logits = run(xt, pos, i)
        if run is not None:
            run.status()                                                          # (synchronises: the persistent decode kernel's error flag)
        tok_h = torch.stack(toks).tolist()                                        # (steps, B), one transfer
        gen = []
        for b in range(B):
            seq = [int(tok_h[i][b]) for i in range(n_steps)]
            if end_id is not None and end_id in seq:
                seq = seq[:seq.index(end_id) + 1]
            gen.append(seq)
        return (gen, torch.stack(all_logits)) if return_logits else gen

    def _decode_runner(self, B, kv, cu, side, cos, sin, max_prompt_len):
        """One decode token = one C call (d3d_phi3_decode_token issues the ~260 launches from C++; from Python they cost more
        host time than the 8-row kernels run).  Returns run(x, pos, t_new) -> logits (B, vocab) float32."""
        from .hip_dense import HipDense, Phi3DecodeArgs, ptr_array
        c = self.cfg
        if not (self.interleave_gu and B <= 16):
            raise RuntimeError("KV-cache decode needs the HIP GEMM backend (interleaved gate/up weights) and <= 16 sequences")
        hdn = HipDense()
        dev, dt = self.device, self.dtype
        buf = dict(h=torch.empty((B, c.hidden), dtype=dt, device=dev), qkv=torch.empty((B, 3 * c.hidden), dtype=dt, device=dev),
                   attn=torch.empty((B, c.hidden), dtype=dt, device=dev), act=torch.empty((B, c.mlp), dtype=dt, device=dev),
                   logits=torch.empty((B, c.vocab), dtype=dt, device=dev))
        if getattr(self, "_w_arrays", None) is None:                            # weight pointer tables: built once per decoder
            self._w_arrays = {k: ptr_array([L[k] for L in self.layers]) for k in ("qkv_w", "o_w", "gu_w", "down_w", "n1", "n2")}
        wa = self._w_arrays
        kv_arr = ptr_array(kv)
        a = Phi3DecodeArgs()
        a.n_layers, a.rows, a.hidden, a.heads, a.head_dim, a.mlp, a.vocab = c.layers, B, c.hidden, c.heads, c.head_dim, c.mlp, c.vocab
        a.dtype, a.rms_eps = (0 if dt == torch.bfloat16 else 1), c.rms_eps
        a.h, a.qkv, a.attn, a.act, a.logits = (buf[k].data_ptr() for k in ("h", "qkv", "attn", "act", "logits"))
        a.qkv_w, a.o_w, a.gate_up_w, a.down_w, a.n1, a.n2 = wa["qkv_w"], wa["o_w"], wa["gu_w"], wa["down_w"], wa["n1"], wa["n2"]
        a.norm_w, a.lm_head_w = self.norm_w.data_ptr(), self.lm_head_w.data_ptr()
        a.cos_t, a.sin_t = cos.data_ptr(), sin.data_ptr()
        a.prompt_qkv, a.cu_seqlens = kv_arr, cu.data_ptr()
        a.knew, a.vnew = side[0, 0].data_ptr(), side[0, 1].data_ptr()
        a.cache_layer_stride_bytes = side.stride(0) * side.element_size()
        a.t_max, a.max_prompt_len = side.shape[3], max_prompt_len
        keep = (buf, kv_arr, kv, side, cos, sin, cu)                           # referenced by raw pointers above

        def run(x, pos, t_new):
            a.x, a.pos, a.t_new = x.data_ptr(), pos.data_ptr(), t_new
            hdn.phi3_decode_token(a)
            return buf["logits"].float() if keep else None
        run.status = hdn.phi3_decode_status
        return run

    @torch.no_grad()
    def prefill_logits(self, inputs_embeds: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """inputs_embeds (B,S,3072) right-padded, lengths (B,) real lengths -> logits (B,vocab) float32 at the last
        real position of each row.  Causal attention makes right padding invisible to real tokens."""
        c = self.cfg
        B, S, Hd = inputs_embeds.shape
        x = inputs_embeds.to(self.dtype).contiguous()
        cos, sin = self._rope(S)
        for L in self.layers:
            h = D.rms_norm(x, L["n1"], c.rms_eps)
            qkv = D.linear(h.view(B * S, Hd), L["qkv_w"], None).view(B, S, c.heads + 2 * c.kv_heads, c.head_dim)
            qkv = D.rope_qk_(qkv, c.heads + c.kv_heads, cos, sin)
            if c.kv_heads == c.heads:
                a = D.attention_qkv(qkv, c.heads, causal=True)
            else:
                q, k, v = qkv[:, :, :c.heads], qkv[:, :, c.heads:c.heads + c.kv_heads], qkv[:, :, c.heads + c.kv_heads:]
                a = D.attention(q, k, v, causal=True)
            x = D.linear(a.reshape(B * S, c.heads * c.head_dim), L["o_w"], None, residual=x.view(B * S, Hd)).view(B, S, Hd)
            h = D.rms_norm(x, L["n2"], c.rms_eps)
            with TIMER.range("phi3.gate_up_proj"):
                act = D.linear_swiglu(h.view(B * S, Hd), L["gu_w"], self.interleave_gu)
            x = D.linear(act, L["down_w"], None, residual=x.view(B * S, Hd)).view(B, S, Hd)
        last = x[torch.arange(B, device=x.device), (lengths.to(x.device).long() - 1)]
        last = D.rms_norm(last, self.norm_w, c.rms_eps)
        return D.linear(last, self.lm_head_w, None).float()


# ---------------------------------------------------------------------------------------------------
# image preprocessing shared by both towers (a3: resnet_encoders.py:267-271)
# ---------------------------------------------------------------------------------------------------
def preprocess_rgb(rgb_u8: torch.Tensor, size: int = 336) -> torch.Tensor:
    """rgb (B,h,w,3) uint8 -> (B,3,336,336) float32 normalised: CHW, bicubic resize in float with the result
    rounded back to uint8 (torchvision's tensor path), /255, CLIP mean/std."""
    return D.resize_normalize(rgb_u8, size, CLIP_MEAN, CLIP_STD)


# ---------------------------------------------------------------------------------------------------
# nn.Module holders of the towers' parameters (see modules.py for the two-copies design)
# ---------------------------------------------------------------------------------------------------
def _is_ln(name: str) -> bool:
    return any(p.startswith("ln_") for p in name.split("."))


def clip_storage_dtype(name: str, clip_dtype):
    """What OpenAI CLIP's `convert_weights` (clip/model.py:373-395) leaves a `visual.*` tensor in: conv / linear / attention weights and
    `proj` in fp16, LayerNorm parameters and the class / positional embeddings in float32 (cast at use, clip/model.py:225-226)."""
    if _is_ln(name) or name in ("visual.class_embedding", "visual.positional_embedding"):
        return torch.float32
    return clip_dtype


class ClipEncoder(nn.Module):
    """`net.rgb_encoder` (encoders/resnet_encoders.py:245-284 `CLIPEncoder`): an nn.Module whose parameters are the image tower's under
    the reference's keys `model.visual.*` (`self.model` = the CLIP model there), frozen like the reference's (:262-264).  The kernels run
    in `self.tower` (`ClipVisionTower`); attributes / methods not found here (`cfg`, `blocks`, `block`, `embed`, ...) resolve there."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: VitConfig = VitConfig(), dtype=torch.float16, device="cuda"):
        super().__init__()
        dev = torch.device(device)
        names = [n for n, _ in clip_param_spec(cfg)]
        self.model = ParamTree({n: own_copy(sd[n], dev, clip_storage_dtype(n, dtype)) for n in names})
        self.tower = ClipVisionTower(self.flat(), cfg, dtype, dev)
        self._sig = self._signature()

    def flat(self) -> Dict[str, torch.Tensor]:
        return self.model.flat()

    def _signature(self):
        return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in self.model.parameters())

    def refresh(self):
        sig = self._signature()
        if sig != self._sig:
            self._sig = sig
            self.tower.device = next(self.model.parameters()).device
            self.tower.load(self.flat())

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            tower = self.__dict__.get("tower")
            if tower is None or name.startswith("__"):
                raise
            return getattr(tower, name)

    @torch.no_grad()
    def forward(self, x):
        """`CLIPEncoder.forward(observations)` -> (view_fts (B,768), grid_fts (B,576,768)) (resnet_encoders.py:273-284); also accepts the
        already pre-processed (B,3,336,336) float32 pixels the policy shares between its two vision towers."""
        if isinstance(x, dict):
            x = preprocess_rgb(x["rgb"].to(self.tower.device))
        return self.tower.forward(x)


class LlavaModel(nn.Module):
    """`net.llava` (VLN-POL:119-131 `LlavaForConditionalGeneration`, transformers 4.46 layout): parameters under `language_model.*`,
    `vision_tower.*`, `multi_modal_projector.*`, all in the model's dtype (`torch_dtype=torch.bfloat16` casts the norm gains too).
    `vision` (`LlavaVisionTower`) and `lm` (`Phi3Decoder`) are the compute objects over them.  Trainable flags as the reference sets
    them: vision tower and projector frozen (VLN-POL:152-155), the language model trainable."""

    def __init__(self, sd: Dict[str, torch.Tensor], vit: VitConfig, llm: Phi3Config, dtype=torch.bfloat16, device="cuda"):
        super().__init__()
        dev = torch.device(device)
        st = lambda n: own_copy(sd[n], dev, dtype)
        for n, _ in phi3_param_spec(llm):
            self._add(n, st(n), True)
        for n, _ in llava_vision_param_spec(vit):
            self._add(n, st(n), False)
        flat = self.flat()
        self.vision = LlavaVisionTower(flat, vit, dtype, dev)
        self.lm = Phi3Decoder(flat, llm, dtype, dev)
        self._sig = self._signature()

    def _add(self, dotted, tensor, requires_grad):
        install_param(self, dotted, tensor, requires_grad)

    def flat(self) -> Dict[str, torch.Tensor]:
        return {k: p.detach() for k, p in self.named_parameters()}

    def _signature(self):
        return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in self.parameters())

    def refresh(self):
        sig = self._signature()
        if sig != self._sig:
            self._sig = sig
            flat = self.flat()
            self.vision.device = self.lm.device = next(self.parameters()).device
            self.vision.load(flat)
            self.lm.load(flat)

    def forward(self, *a, **k):
        raise RuntimeError("LlavaModel holds the parameters; Dynam3D_VLN drives `vision` / `lm` (packed prefill, KV-cache decode)")
