"""Dense primitive dispatch for the towers.  Each primitive names the hand-written gfx950 kernel that implements it
(csrc/gemm_kernels.hip, attn3_kernels.hip, dense_kernels.hip, tower_kernels.hip).  On a CUDA tensor the HIP kernel is the
ONLY path by default (`STRICT = True`: a shape / dtype without a kernel raises `DenseFallbackError`); the PyTorch
expression beside each primitive exists for CPU tensors (the CPU suite's host-logic path) and behind the explicit
opt-out `allow_fallback()` (toy test configurations, `bench.py --hip-dense` A/B baselines), counted per primitive.
`BACKEND` records what each primitive is bound to so the benchmark line reports exactly what ran.

Shapes: activations are row-major (tokens, features); weights are [out, in] like nn.Linear.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn.functional as F

BACKEND = {"linear": "torch/hipBLASLt", "attention": "torch/SDPA", "layer_norm": "torch", "rms_norm": "torch",
           "rope": "torch", "swiglu": "torch", "resize_normalize": "torch", "vit_embed": "torch"}

_hip = None  # set by enable_hip_kernels()

# ---- strict mode + dispatch accounting ---------------------------------------------------------------------------
# Every primitive below either launches a hand-written kernel ("hip") or -- when a shape / dtype predicate fails, or on
# the CPU -- runs the PyTorch expression next to it.  On a GPU that second branch would be a silent change of backend
# (hipBLASLt / SDPA instead of the product's kernels), so STRICT IS THE DEFAULT: a CUDA tensor that cannot take the HIP
# kernel RAISES.  The PyTorch expressions stay for CPU tensors (the CPU suite drives the host logic through them by design;
# never counted) and for callers that opt out explicitly -- `strict(False)` / `with allow_fallback():` (A/B baselines of
# bench.py --hip-dense, toy configurations below the kernels' tile sizes in tests); opted-out calls are counted per primitive
# and the benchmark line prints both tables (`fallbacks` must be empty).
STRICT = True
COUNTS = {"hip": {}, "fallback": {}}


class DenseFallbackError(RuntimeError):
    pass


def strict(on: bool = True):
    """Raise `DenseFallbackError` whenever a CUDA tensor would take a PyTorch expression instead of a HIP kernel."""
    global STRICT
    STRICT = bool(on)


class allow_fallback:
    """`with allow_fallback():` -- the explicit opt-out of strict mode for a block (restores the previous mode)."""

    def __enter__(self):
        global STRICT
        self.was, STRICT = STRICT, False
        return self

    def __exit__(self, *exc):
        global STRICT
        STRICT = self.was
        return False


def reset_counts():
    COUNTS["hip"].clear()
    COUNTS["fallback"].clear()


def counts():
    return {"hip": dict(COUNTS["hip"]), "fallback": dict(COUNTS["fallback"])}


def _hit(prim: str):
    COUNTS["hip"][prim] = COUNTS["hip"].get(prim, 0) + 1


def _miss(prim: str, t: torch.Tensor, why: str):
    if not t.is_cuda:
        return
    COUNTS["fallback"][prim] = COUNTS["fallback"].get(prim, 0) + 1
    if STRICT:
        raise DenseFallbackError(f"dense_ops.{prim}: no HIP kernel for this call on a CUDA tensor ({why}); the PyTorch expression is a different "
                                 f"backend and is not taken silently -- opt out with dense_ops.strict(False) / `with dense_ops.allow_fallback():`")


def enable_hip_kernels(which: Sequence[str] = ("all",)):
    """Switch primitives to the hand-written HIP kernels (requires libdynam3d_hip.so + a GPU)."""
    global _hip
    from . import hip_dense
    _hip = hip_dense.HipDense()
    names = list(BACKEND) if "all" in which else list(which)
    for n in names:
        if _hip.has(n):
            BACKEND[n] = "hip"
    return dict(BACKEND)


def _f32(x: torch.Tensor) -> bool:
    """The float32 VERIFICATION MODE (csrc/verify_f32_kernels.hip): a float32 CUDA tensor on the HIP backend runs the float32 twin of the
    primitive's kernel -- same host wiring as the 16-bit product path, float32 arithmetic, so that north_star's 1e-3 against the float32
    oracle can be asserted on it under strict dispatch (towers built with dtype=float32)."""
    return x.is_cuda and x.dtype == torch.float32 and _hip is not None


def _act(y, act):
    if act is None:
        return y
    if act == "quick_gelu":               # clip/model.py:162-164
        return y * torch.sigmoid(1.702 * y)
    if act == "gelu":
        return F.gelu(y)
    raise ValueError(act)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: Optional[str] = None,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x w^T + b) [+ residual].  x (M,K), w (N,K).  On the HIP backend bias / activation / residual are
    fused into the GEMM epilogue (csrc/gemm_kernels.hip)."""
    if BACKEND["linear"] == "hip" and x.is_cuda and _hip.gemm_ok(x, w):
        _hit("linear")
        return _hip.linear(x, w, b, act, residual)
    if BACKEND["linear"] == "hip" and _f32(x) and _hip.gemm_f32_ok(x, w):
        _hit("linear")
        return _hip.linear_f32(x, w, b, act, residual)
    _miss("linear", x, f"backend {BACKEND['linear']}, x {tuple(x.shape)} {x.dtype}, w {tuple(w.shape)} {w.dtype}")
    y = _act(F.linear(x, w, b), act)
    return y if residual is None else y + residual.reshape(y.shape)


def linear_swiglu(x: torch.Tensor, w_gate_up: torch.Tensor, interleaved: bool) -> torch.Tensor:
    """Phi-3 MLP front half: up * silu(gate) of x w^T.  `interleaved` says whether w's rows were re-laid-out by
    hip_dense.interleave_gate_up (per-16 gate/up blocks) for the fused-epilogue kernel."""
    if interleaved and BACKEND["linear"] == "hip" and x.is_cuda and _hip.gemm_ok(x, w_gate_up):
        _hit("linear_swiglu")
        return _hip.linear_swiglu(x, w_gate_up)
    if not interleaved and BACKEND["linear"] == "hip" and _f32(x) and _hip.gemm_f32_ok(x, w_gate_up):
        _hit("linear_swiglu")
        return _hip.swiglu_f32(_hip.linear_f32(x, w_gate_up, None, None))
    _miss("linear_swiglu", x, f"interleaved {interleaved}, x {tuple(x.shape)} {x.dtype}, w {tuple(w_gate_up.shape)}")
    gu = F.linear(x, w_gate_up)
    if interleaved:
        I = gu.shape[-1] // 2
        gu = gu.view(-1, I // 16, 2, 16)
        g, u = gu[:, :, 0].reshape(-1, I), gu[:, :, 1].reshape(-1, I)
        return u * F.silu(g)                                  # HF Phi3MLP on the activations' dtype: silu and the product are stored
    return swiglu(gu)


def r16(x: torch.Tensor, dtype) -> torch.Tensor:
    """A 16-bit store-and-reload at one of the reference's module boundaries (no-op for float32 towers)."""
    return x if dtype == torch.float32 else x.to(dtype).to(x.dtype)


def vit_embed(pixels: torch.Tensor, patch_w: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor,
              patch: int, eps: float = 1e-5) -> torch.Tensor:
    """ViT embeddings + ln_pre (clip/model.py:222-228): conv(patch, stride patch, no bias) as a GEMM over unfolded patches,
    [cls; patches] + positional embedding, LayerNorm.  pixels (B,3,S,S) f32; patch_w (W, Kp) with Kp = 3*patch^2 rounded up to
    a multiple of 64 (zero columns); cls (W), pos (L, W) in the tower's dtype.  -> (B, L, W)."""
    B, _, S, _ = pixels.shape
    G = S // patch
    dt = patch_w.dtype
    K = 3 * patch * patch
    if BACKEND["vit_embed"] == "hip" and pixels.is_cuda and dt in (torch.bfloat16, torch.float16) and _hip.gemm_ok(patch_w[:1], patch_w):
        _hit("vit_embed")
        rows = _hip.patchify(pixels, patch, patch_w.shape[1], dt)
        x = linear(rows, patch_w, None)
        return _hip.vit_embed_ln(x, cls, pos, ln_w, ln_b, B, eps)
    if BACKEND["vit_embed"] == "hip" and pixels.is_cuda and dt == torch.float32 and _hip.gemm_f32_ok(patch_w[:1], patch_w):
        _hit("vit_embed")
        x = linear(_hip.patchify_f32(pixels, patch, patch_w.shape[1]), patch_w, None)
        return _hip.vit_embed_ln_f32(x, cls, pos, ln_w, ln_b, B, eps)
    _miss("vit_embed", pixels, f"dtype {dt}, patch_w {tuple(patch_w.shape)}")
    pt = pixels.to(dt).view(B, 3, G, patch, G, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, K)
    x = F.linear(pt, patch_w[:, :K]).view(B, G * G, -1)
    x = torch.cat([cls.expand(B, 1, -1).to(x.dtype), x], dim=1) + pos.to(x.dtype)          # a 16-bit add in the 16-bit towers
    return layer_norm(x, ln_w, ln_b, eps)


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """float32 statistics and affine, result in x.dtype (clip/model.py:153-159)."""
    if BACKEND["layer_norm"] == "hip" and x.is_cuda and _hip.norm_ok(x):
        _hit("layer_norm")
        return _hip.layer_norm(x, w, b, eps)
    if BACKEND["layer_norm"] == "hip" and _f32(x) and x.shape[-1] % 4 == 0 and x.shape[-1] <= 3072:
        _hit("layer_norm")
        return _hip.layer_norm_f32(x, w, b, eps)
    _miss("layer_norm", x, f"x {tuple(x.shape)} {x.dtype}")
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).to(x.dtype)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    if BACKEND["rms_norm"] == "hip" and x.is_cuda and _hip.norm_ok(x):
        _hit("rms_norm")
        return _hip.rms_norm(x, w, eps)
    if BACKEND["rms_norm"] == "hip" and _f32(x) and x.shape[-1] % 4 == 0:
        _hit("rms_norm")
        return _hip.rms_norm_f32(x, w, eps)
    _miss("rms_norm", x, f"x {tuple(x.shape)} {x.dtype}")
    xf = x.float()
    xh = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)          # HF Phi3RMSNorm: weight * x_hat.to(input_dtype)
    return (xh.float() * w).to(x.dtype)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool) -> torch.Tensor:
    """q,k,v (B,L,H,hd) (any strides) -> (B,L,H,hd) contiguous; softmax scale 1/sqrt(hd)."""
    _miss("attention", q, f"SDPA on q {tuple(q.shape)} {q.dtype}")
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=causal)
    return o.transpose(1, 2).contiguous()


def attention_qkv(qkv: torch.Tensor, n_heads: int, causal: bool) -> torch.Tensor:
    """Self-attention over the fused projection qkv (B,S,3H,hd) laid out [q heads | k heads | v heads] -> (B,S,H,hd).
    HIP backend: flash kernel reading q/k/v in place (csrc/attn_kernels.hip); otherwise SDPA on strided views."""
    B, S, Ht, hd = qkv.shape
    if BACKEND["attention"] == "hip" and qkv.is_cuda and Ht == 3 * n_heads and _hip.attention_ok(qkv, hd):
        _hit("attention")
        return _hip.attention_qkv(qkv, n_heads, causal)
    if BACKEND["attention"] == "hip" and _f32(qkv) and Ht == 3 * n_heads and hd in (64, 96) and qkv.is_contiguous():
        _hit("attention")
        return _hip.attention_qkv_f32(qkv, n_heads, causal)
    return attention(qkv[:, :, :n_heads], qkv[:, :, n_heads:2 * n_heads], qkv[:, :, 2 * n_heads:], causal)


def packed_ok(dtype, head_dim: int) -> bool:
    """True when the packed (variable-length, no padding) decoder path is available: HIP rope + flash attention."""
    return (BACKEND["attention"] == "hip" and BACKEND["rope"] == "hip" and dtype in (torch.bfloat16, torch.float16, torch.float32)
            and head_dim in (64, 96) and (head_dim // 2) % 8 == 0)


def rope_packed_(qkv2d: torch.Tensor, n_rot_heads: int, head_dim: int, cos, sin, pos: torch.Tensor):
    _hit("rope")
    _hip.rope_inplace(qkv2d, cos, sin, 1, n_rot_heads, head_dim, pos)
    return qkv2d


def attention_packed(qkv3: torch.Tensor, n_heads: int, causal: bool, cu_seqlens: torch.Tensor, n_seq: int, max_len: int, n_valid=None, window: int = 0,
                     out=None, rope_q=None, sched=None):
    """`rope_q` = (cos, sin): the buffer's queries are NOT rotated yet, the kernel rotates them (see `can_fuse_rope_q`).
    `sched`: workgroup table from `attention_schedule` (one query block per workgroup, heaviest first)."""
    _hit("attention")
    if qkv3.dtype == torch.float32:
        assert rope_q is None and sched is None          # (the float32 mode rotates q and k in place: `can_fuse_rope_q(float32)` is False)
        return _hip.attention_packed_f32(qkv3, n_heads, causal, cu_seqlens, n_seq, max_len, n_valid, window, out)
    return _hip.attention_packed(qkv3, n_heads, causal, cu_seqlens, n_seq, max_len, n_valid, window, out, rope_q, sched)


def attention_schedule(lens, n_heads: int, device) -> torch.Tensor:
    return torch.from_numpy(_hip.attention_schedule(lens, n_heads)).to(device)


def can_fuse_rope_q(dtype=None) -> bool:
    if dtype == torch.float32:
        return False
    return BACKEND["attention"] == "hip" and BACKEND["rope"] == "hip" and _hip is not None and _hip.can_fuse_rope_q()


def decode_attention(qkv_new, prompt_qkv, cu_seqlens, knew, vnew, n_heads: int, t_new: int, max_prompt_len: int, rope=None):
    return _hip.decode_attention(qkv_new, prompt_qkv, cu_seqlens, knew, vnew, n_heads, t_new, max_prompt_len, rope)


def rope_qk_(qkv: torch.Tensor, n_rot_heads: int, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Rotate the first `n_rot_heads` heads (q heads then k heads) of the fused projection qkv (B,S,Htot,hd);
    in place on the HIP backend (one pass over q,k instead of slice/float/cat round trips)."""
    B, S, Ht, hd = qkv.shape
    if (BACKEND["rope"] == "hip" and qkv.is_cuda and qkv.is_contiguous() and qkv.dtype in (torch.bfloat16, torch.float16, torch.float32)
            and (hd // 2) % 8 == 0):
        _hit("rope")
        _hip.rope_inplace(qkv.view(B * S, Ht * hd), cos, sin, S, n_rot_heads, hd)
        return qkv
    _miss("rope", qkv, f"qkv {tuple(qkv.shape)} {qkv.dtype}")
    return torch.cat([rope(qkv[:, :, :n_rot_heads], cos, sin), qkv[:, :, n_rot_heads:]], dim=2)


def rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Half-split rotary embedding (HF `rotate_half`): pairs (i, i+hd/2).  x (B,S,H,hd); cos/sin (S,hd/2) f32."""
    hd = x.shape[-1]
    x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
    c, s = cos[None, :, None, :].to(x.dtype), sin[None, :, None, :].to(x.dtype)          # HF: cos/sin cast to the activations' dtype,
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)                           # products and sums stored in it


def swiglu(gu: torch.Tensor) -> torch.Tensor:
    """Phi-3 MLP: gate, up = chunk(gate_up, 2); up * silu(gate)."""
    if BACKEND["swiglu"] == "hip" and gu.is_cuda and gu.dtype in (torch.bfloat16, torch.float16) and gu.shape[-1] % 16 == 0:
        _hit("swiglu")
        return _hip.swiglu(gu)
    if BACKEND["swiglu"] == "hip" and _f32(gu) and gu.shape[-1] % 8 == 0:
        _hit("swiglu")
        return _hip.swiglu_f32(gu)
    _miss("swiglu", gu, f"gu {tuple(gu.shape)} {gu.dtype}")
    g, u = gu.chunk(2, dim=-1)
    return u * F.silu(g)


def resize_normalize(rgb_u8: torch.Tensor, size: int, mean, std) -> torch.Tensor:
    if BACKEND["resize_normalize"] == "hip" and rgb_u8.is_cuda:
        _hit("resize_normalize")
        return _hip.resize_normalize(rgb_u8, size, mean, std)
    _miss("resize_normalize", rgb_u8, "backend")
    x = rgb_u8.permute(0, 3, 1, 2).float()
    if x.shape[-1] != size or x.shape[-2] != size:
        x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=False)
        x = x.round().clamp(0, 255)                   # torchvision casts the interpolated image back to uint8
    x = x / 255.0
    m = torch.tensor(mean, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32, device=x.device).view(1, 3, 1, 1)
    return (x - m) / s
