"""ctypes wrappers of the hand-written dense kernels (csrc/gemm_kernels.hip, dense_kernels.hip, attn_kernels.hip)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import f32, i32, i64, vp

EPI = dict(none=0, bias=1, bias_quick_gelu=2, bias_gelu=3, res=4, bias_res=5, swiglu=6, lrelu=7, lrelu_bwd=8)

_lib.register("d3d_gemm_nt", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, vp])
_lib.register("d3d_gemm_nt_rmsnorm", [vp, vp, f32, vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, vp])
_lib.register("d3d_gemm_nt_tile", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, i32, i32, vp])
_lib.register("d3d_gemm_reserve_workspace", [vp])
_lib.register("d3d_norm", [vp, vp, vp, vp, i32, i32, i64, i64, f32, i32, i32, vp])
_lib.register("d3d_rope_inplace", [vp, vp, vp, i32, i32, i32, i32, i64, vp, i32, vp])
_lib.register("d3d_set_attention", [vp, vp, i32, i32, i32, i32, vp, vp])
_lib.register("d3d_flash_attention_v3", [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, vp, i32, i32, vp])
_lib.register("d3d_flash_attention_v3_rope_q", [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp, i32, vp])
_lib.register("d3d_flash_attention_v3_sched", [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, i32, i32, vp])
_lib.register("d3d_flash_attention_v4", [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp])
_lib.register("d3d_swiglu", [vp, vp, i64, i32, i32, vp])
_lib.register("d3d_resize_normalize", [vp, vp, i32, i32, i32, i32, vp, vp, vp])
_lib.register("d3d_decode_attention", [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp])
_lib.register("d3d_phi3_decode_token", [vp])
_lib.register("d3d_phi3_decode_status", [vp])
_lib.register("d3d_patchify", [vp, vp, i32, i32, i32, i32, i32, vp])
_lib.register("d3d_vit_embed_ln", [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp])
_lib.register("d3d_assemble_prompt", [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp])
# float32 verification mode (csrc/verify_f32_kernels.hip; GEMMs / LayerNorm: d3d_gemm_nt_f32 / d3d_layer_norm_f32, registered in f32_ops)
_lib.register("d3d_attention_f32", [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, vp, i32, vp])
_lib.register("d3d_rms_norm_f32", [vp, vp, vp, i32, i32, i64, i64, f32, vp])
_lib.register("d3d_rope_inplace_f32", [vp, vp, vp, i32, i32, i32, i32, i64, vp, vp])
_lib.register("d3d_swiglu_f32", [vp, vp, i64, i32, vp])
_lib.register("d3d_patchify_f32", [vp, vp, i32, i32, i32, i32, vp])
_lib.register("d3d_vit_embed_ln_f32", [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp])
_lib.register("d3d_assemble_prompt_f32", [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp])


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def interleave_gate_up(w: torch.Tensor, block: int = 16) -> torch.Tensor:
    """[gate(I); up(I)] rows -> [g16|u16|g16|u16|...] so one wave's MFMA tiles 2j / 2j+1 hold matching
    gate / up columns and SwiGLU is applied on accumulator registers (gemm_kernels.hip EPI_SWIGLU)."""
    I = w.shape[0] // 2
    g, u = w[:I].view(I // block, block, -1), w[I:].view(I // block, block, -1)
    return torch.stack([g, u], dim=1).reshape(2 * I, -1).contiguous()


class Phi3DecodeArgs(C.Structure):
    """ctypes mirror of `d3d_phi3_decode_args` (include/dynam3d_hip.h)."""
    _fields_ = [("n_layers", C.c_int32), ("rows", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
                ("mlp", C.c_int32), ("vocab", C.c_int32), ("dtype", C.c_int32), ("rms_eps", C.c_float),
                ("x", C.c_void_p), ("h", C.c_void_p), ("qkv", C.c_void_p), ("attn", C.c_void_p), ("act", C.c_void_p), ("logits", C.c_void_p),
                ("qkv_w", C.POINTER(C.c_void_p)), ("o_w", C.POINTER(C.c_void_p)), ("gate_up_w", C.POINTER(C.c_void_p)),
                ("down_w", C.POINTER(C.c_void_p)), ("n1", C.POINTER(C.c_void_p)), ("n2", C.POINTER(C.c_void_p)),
                ("norm_w", C.c_void_p), ("lm_head_w", C.c_void_p), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("pos", C.c_void_p),
                ("prompt_qkv", C.POINTER(C.c_void_p)), ("cu_seqlens", C.c_void_p), ("knew", C.c_void_p), ("vnew", C.c_void_p),
                ("cache_layer_stride_bytes", C.c_int64), ("t_new", C.c_int32), ("t_max", C.c_int32), ("max_prompt_len", C.c_int32),
                ("stream", C.c_void_p)]


def ptr_array(tensors):
    """Host array of device pointers (kept alive by the caller together with the tensors)."""
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class HipDense:
    PRIMS = {"linear", "layer_norm", "rms_norm", "rope", "swiglu", "resize_normalize", "attention", "vit_embed"}

    def __init__(self):
        self.lib = _lib.load()

    def has(self, name):
        return name in self.PRIMS

    @staticmethod
    def _stream():
        return _lib.current_stream_ptr()

    @staticmethod
    def gemm_ok(x, w):
        K = x.shape[-1]
        rows = x.numel() // max(K, 1)
        shape_ok = (w.shape[0] % 128 == 0 and K % 64 == 0) or (rows <= 16 and w.shape[0] % 32 == 0 and K % 32 == 0)   # skinny kernel
        return (shape_ok and x.dtype == w.dtype and x.dtype in (torch.bfloat16, torch.float16)
                and x.stride(-1) == 1 and w.is_contiguous())

    TILE = 0   # 0 = library heuristic, 128 / 130 / 132 / 256 / 257 / 258 = force a kernel variant (benchmarking, tests)

    def gemm(self, x, w, bias, residual, epi: str):
        M, K = x.shape
        N = w.shape[0]
        n_out = N // 2 if epi == "swiglu" else N
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
        if bias is not None and bias.dtype != x.dtype:
            bias = bias.to(x.dtype)
        dt = 0 if x.dtype == torch.bfloat16 else 1
        if self.TILE:
            tile = self.TILE if (self.TILE < 256 or N % 256 == 0) else 128
            _lib.check(self.lib.d3d_gemm_nt_tile(_p(x), _p(w), _p(out), _p(bias), _p(residual), M, N, K, x.stride(0), w.stride(0), n_out,
                                                 dt, EPI[epi], tile, self._stream()))
        else:
            _lib.check(self.lib.d3d_gemm_nt(_p(x), _p(w), _p(out), _p(bias), _p(residual), M, N, K, x.stride(0), w.stride(0), n_out,
                                            dt, EPI[epi], self._stream()))
        return out

    def linear(self, x, w, b, act, residual=None):
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1 or (x2.stride(0) % 8) != 0:
            x2 = x2.contiguous()
        if residual is not None:
            residual = residual.reshape(-1, w.shape[0])
            if not residual.is_contiguous():
                residual = residual.contiguous()
        if act is None:
            epi = ("bias_res" if b is not None else "res") if residual is not None else ("bias" if b is not None else "none")
        else:
            assert residual is None and b is not None
            epi = {"quick_gelu": "bias_quick_gelu", "gelu": "bias_gelu"}[act]
        return self.gemm(x2, w, b, residual, epi)

    def linear_swiglu(self, x, w_interleaved):
        return self.gemm(x, w_interleaved, None, None, "swiglu")

    def linear_rmsnorm(self, x, norm_w, eps, w, swiglu=False):
        """epilogue(RMSNorm(x) w^T) for <= 16 rows, the norm applied inside the weight-streaming GEMM (d3d_gemm_nt_rmsnorm)."""
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty((M, N // 2 if swiglu else N), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.d3d_gemm_nt_rmsnorm(_p(x), _p(norm_w), float(eps), _p(w), _p(out), M, N, K, x.stride(0), w.stride(0), out.stride(0),
                                                0 if x.dtype == torch.bfloat16 else 1, 6 if swiglu else 0, self._stream()))
        return out

    # ---- row kernels (csrc/dense_kernels.hip) -------------------------------------------------------------
    @staticmethod
    def norm_ok(x):
        D = x.shape[-1]
        return x.dtype in (torch.bfloat16, torch.float16) and D % 8 == 0 and D <= 4096 and ((D + 511) // 512) in (1, 2, 3, 4, 6, 8)

    def _norm(self, x, w, b, eps, rms):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1 or x2.stride(0) % 8:
            x2 = x2.contiguous()
        y = torch.empty((x2.shape[0], x2.shape[1]), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.d3d_norm(_p(x2), _p(w), _p(b), _p(y), x2.shape[0], x2.shape[1], x2.stride(0), y.stride(0), eps, 1 if rms else 0,
                                     0 if x.dtype == torch.bfloat16 else 1, self._stream()))
        return y.view(x.shape)

    def layer_norm(self, x, w, b, eps):
        return self._norm(x, w, b, eps, False)

    def rms_norm(self, x, w, eps):
        return self._norm(x, w, None, eps, True)

    def rope_inplace(self, qkv2d, cos, sin, S, n_rot_heads, hd, pos=None):
        """qkv2d (rows, >= n_rot_heads*hd) bf16 / fp16 (float32: the verification mode's kernel), rotated in place; position = pos[row] if
        given else row % S."""
        if qkv2d.dtype == torch.float32:
            _lib.check(self.lib.d3d_rope_inplace_f32(_p(qkv2d), _p(cos), _p(sin), qkv2d.shape[0], S, n_rot_heads, hd, qkv2d.stride(0), _p(pos), self._stream()))
            return
        _lib.check(self.lib.d3d_rope_inplace(_p(qkv2d), _p(cos), _p(sin), qkv2d.shape[0], S, n_rot_heads, hd, qkv2d.stride(0), _p(pos),
                                             0 if qkv2d.dtype == torch.bfloat16 else 1, self._stream()))

    def decode_attention(self, qkv_new, prompt_qkv, cu_seqlens, knew, vnew, n_heads, t_new, max_prompt_len, rope=None):
        """One KV-cache decode step: qkv_new (B, 3H*hd) rotated (or un-rotated with rope=(cos, sin, pos)); prompt_qkv (T, 3H*hd) the
        layer's prefill buffer; knew / vnew (B, Tmax, H, hd) side caches (token t_new is appended here).  -> (B, H*hd)."""
        B = qkv_new.shape[0]
        Tmax, hd = knew.shape[1], knew.shape[3]
        out = torch.empty((B, n_heads * hd), dtype=qkv_new.dtype, device=qkv_new.device)
        cos, sin, pos = rope if rope is not None else (None, None, None)
        _lib.check(self.lib.d3d_decode_attention(_p(qkv_new), _p(prompt_qkv), _p(cu_seqlens), _p(knew), _p(vnew), _p(out), B, n_heads, hd, t_new, Tmax,
                                                 max_prompt_len, _p(cos), _p(sin), _p(pos), 0 if qkv_new.dtype == torch.bfloat16 else 1, self._stream()))
        return out

    def phi3_decode_token(self, args: "Phi3DecodeArgs"):
        """All launches of one KV-cache decode token, issued from C++ (d3d_phi3_decode_token)."""
        args.stream = self._stream()
        _lib.check(self.lib.d3d_phi3_decode_token(C.byref(args)))

    def phi3_decode_status(self):
        """Synchronises the stream and raises if a grid barrier of the persistent decode kernel timed out (d3d_phi3_decode_status)."""
        _lib.check(self.lib.d3d_phi3_decode_status(self._stream()))

    def patchify(self, pixels, patch, Kp, dtype):
        """pixels (B,3,S,S) f32 -> (B*(S/patch)^2, Kp) rows of unfolded patches in `dtype`, zero-padded columns (d3d_patchify)."""
        px = pixels.contiguous()
        B, _, S, _ = px.shape
        G = S // patch
        out = torch.empty((B * G * G, Kp), dtype=dtype, device=px.device)
        _lib.check(self.lib.d3d_patchify(_p(px), _p(out), B, S, patch, Kp, 0 if dtype == torch.bfloat16 else 1, self._stream()))
        return out

    def vit_embed_ln(self, patch_rows, cls, pos, ln_w, ln_b, B, eps):
        """[cls; patch rows] + pos (16-bit add) -> ln_pre -> (B, L, D)  (d3d_vit_embed_ln)."""
        L, D = pos.shape
        out = torch.empty((B, L, D), dtype=patch_rows.dtype, device=patch_rows.device)
        _lib.check(self.lib.d3d_vit_embed_ln(_p(patch_rows), _p(cls), _p(pos), _p(ln_w), _p(ln_b), _p(out), B, L, D, eps,
                                             0 if patch_rows.dtype == torch.bfloat16 else 1, self._stream()))
        return out

    def assemble_prompt(self, desc, embed, patch_feat, patch_pos, inst, zone, rows):
        """The packed prompt rows of all environments in one pass (d3d_assemble_prompt): desc (rows,) int32 = (source << 28) | row."""
        D, dt = embed.shape[1], embed.dtype
        for t in (patch_feat, patch_pos, inst, zone):
            assert t.dtype == dt and t.is_contiguous() and t.shape[-1] == D, (t.dtype, t.shape)
        out = torch.empty((rows, D), dtype=dt, device=embed.device)
        _lib.check(self.lib.d3d_assemble_prompt(_p(desc), _p(embed), _p(patch_feat), _p(patch_pos), _p(inst), _p(zone), _p(out), rows, D,
                                                0 if dt == torch.bfloat16 else 1, self._stream()))
        return out

    def resize_normalize(self, rgb_u8, size, mean, std):
        import numpy as np
        rgb = rgb_u8.contiguous()
        B, H, W, _ = rgb.shape
        out = torch.empty((B, 3, size, size), dtype=torch.float32, device=rgb.device)
        m, s = np.asarray(mean, np.float32), np.asarray(std, np.float32)
        _lib.check(self.lib.d3d_resize_normalize(_p(rgb), _p(out), B, H, W, size, m.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), self._stream()))
        return out

    def swiglu(self, gu):
        gu2 = gu.reshape(-1, gu.shape[-1]).contiguous()
        I = gu2.shape[1] // 2
        out = torch.empty((gu2.shape[0], I), dtype=gu.dtype, device=gu.device)
        _lib.check(self.lib.d3d_swiglu(_p(gu2), _p(out), gu2.shape[0], I, 0 if gu.dtype == torch.bfloat16 else 1, self._stream()))
        return out

    def set_attention(self, qkv, set_off, n_sets, n_heads, max_len, q_rows=0):
        """qkv (T, 3*H*64) f32 packed sets -> (T, H*64) f32 (rows outside the queried range are zero)."""
        out = torch.zeros((qkv.shape[0], n_heads * 64), dtype=torch.float32, device=qkv.device)
        _lib.check(self.lib.d3d_set_attention(_p(qkv), _p(set_off), n_sets, n_heads, max_len, q_rows, _p(out), self._stream()))
        return out

    # ---- float32 verification mode (csrc/verify_f32_kernels.hip + the float32-MFMA GEMM / LayerNorm of f32_kernels.hip) -------------------
    F32_EPI = dict(none=0, bias=1, bias_gelu=2, bias_res=3, bias_quick_gelu=4, res=5)

    @staticmethod
    def gemm_f32_ok(x, w):
        K = x.shape[-1]
        return (x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[0] % 4 == 0 and K % 16 == 0 and K == w.shape[1]
                and x.stride(-1) == 1 and w.is_contiguous())

    def linear_f32(self, x, w, b, act, residual=None):
        """act(x w^T + b) [+ residual] in float32 on v_mfma_f32_16x16x4_f32 (d3d_gemm_nt_f32: an exact float32 multiply-add chain)."""
        from . import f32_ops  # noqa: F401  (registers d3d_gemm_nt_f32 / d3d_layer_norm_f32)
        x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        if residual is not None:
            residual = residual.reshape(-1, N)
            if not residual.is_contiguous():
                residual = residual.contiguous()
        if act is None:
            epi = ("bias_res" if b is not None else "res") if residual is not None else ("bias" if b is not None else "none")
        else:
            assert residual is None and b is not None
            epi = {"quick_gelu": "bias_quick_gelu", "gelu": "bias_gelu"}[act]
        if b is not None and b.dtype != torch.float32:
            b = b.float()
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.d3d_gemm_nt_f32(_p(x2), _p(w), _p(y), _p(b), _p(residual), M, N, K, x2.stride(0), K, N, self.F32_EPI[epi], self._stream()))
        return y

    def layer_norm_f32(self, x, w, b, eps):
        from . import f32_ops  # noqa: F401
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        y = torch.empty(x2.shape, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.d3d_layer_norm_f32(_p(x2), None, _p(w), _p(b), _p(y), x2.shape[0], x2.shape[1], x2.stride(0), 0, x2.shape[1], eps, 0, self._stream()))
        return y.view(x.shape)

    def rms_norm_f32(self, x, w, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        y = torch.empty(x2.shape, dtype=torch.float32, device=x.device)
        _lib.check(self.lib.d3d_rms_norm_f32(_p(x2), _p(w), _p(y), x2.shape[0], x2.shape[1], x2.stride(0), x2.shape[1], eps, self._stream()))
        return y.view(x.shape)

    def swiglu_f32(self, gu):
        gu2 = gu.reshape(-1, gu.shape[-1]).contiguous()
        I = gu2.shape[1] // 2
        out = torch.empty((gu2.shape[0], I), dtype=torch.float32, device=gu.device)
        _lib.check(self.lib.d3d_swiglu_f32(_p(gu2), _p(out), gu2.shape[0], I, self._stream()))
        return out

    def patchify_f32(self, pixels, patch, Kp):
        px = pixels.contiguous()
        B, _, S, _ = px.shape
        G = S // patch
        out = torch.empty((B * G * G, Kp), dtype=torch.float32, device=px.device)
        _lib.check(self.lib.d3d_patchify_f32(_p(px), _p(out), B, S, patch, Kp, self._stream()))
        return out

    def vit_embed_ln_f32(self, patch_rows, cls, pos, ln_w, ln_b, B, eps):
        L, D = pos.shape
        out = torch.empty((B, L, D), dtype=torch.float32, device=patch_rows.device)
        _lib.check(self.lib.d3d_vit_embed_ln_f32(_p(patch_rows.contiguous()), _p(cls.contiguous()), _p(pos.contiguous()), _p(ln_w), _p(ln_b), _p(out), B, L, D,
                                                 eps, self._stream()))
        return out

    def assemble_prompt_f32(self, desc, embed, patch_feat, patch_pos, inst, zone, rows):
        D = embed.shape[1]
        for t in (embed, patch_feat, patch_pos, inst, zone):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == D, (t.dtype, t.shape)
        out = torch.empty((rows, D), dtype=torch.float32, device=embed.device)
        _lib.check(self.lib.d3d_assemble_prompt_f32(_p(desc), _p(embed), _p(patch_feat), _p(patch_pos), _p(inst), _p(zone), _p(out), rows, D, self._stream()))
        return out

    def attention_qkv_f32(self, qkv, n_heads, causal, window=0):
        B, S, Ht, hd = qkv.shape
        out = torch.empty((B, S, n_heads, hd), dtype=torch.float32, device=qkv.device)
        _lib.check(self.lib.d3d_attention_f32(_p(qkv), _p(out), B, S, n_heads, hd, Ht * hd, S * Ht * hd, 0, n_heads, 2 * n_heads, 1 if causal else 0, S, None,
                                              window, self._stream()))
        return out

    def attention_packed_f32(self, qkv, n_heads, causal, cu_seqlens, n_seq, max_len, n_valid=None, window=0, out=None):
        T, Ht, hd = qkv.shape
        if out is None:
            out = torch.zeros((T, n_heads, hd), dtype=torch.float32, device=qkv.device)
        assert out.shape == (T, n_heads, hd) and out.dtype == torch.float32 and out.is_contiguous()
        _lib.check(self.lib.d3d_attention_f32(_p(qkv), _p(out), n_seq, max_len, n_heads, hd, Ht * hd, 0, 0, n_heads, 2 * n_heads, 1 if causal else 0, max_len,
                                              _p(cu_seqlens), window, self._stream()))
        return out

    @staticmethod
    def attention_ok(qkv, hd):
        return qkv.dtype in (torch.bfloat16, torch.float16) and hd in (64, 96) and qkv.is_contiguous()

    def attention_qkv(self, qkv, n_heads, causal, seq_len=None, window=0):
        """qkv (B,S,3H,hd) contiguous fused projection -> (B,S,H,hd): flash attention straight off the projection buffer."""
        B, S, Ht, hd = qkv.shape
        out = torch.empty((B, S, n_heads, hd), dtype=qkv.dtype, device=qkv.device)
        _lib.check(self.lib.d3d_flash_attention_v3(_p(qkv), _p(out), B, S, n_heads, hd, Ht * hd, S * Ht * hd, 0, n_heads, 2 * n_heads, 1 if causal else 0,
                                                   S if seq_len is None else seq_len, None, window, 0 if qkv.dtype == torch.bfloat16 else 1, self._stream()))
        return out

    def can_fuse_rope_q(self):
        """The attention kernel can rotate the queries itself (d3d_flash_attention_v3_rope_q)."""
        return True

    @staticmethod
    def attention_schedule(lens, n_heads, block=128):
        """Workgroup table of `d3d_flash_attention_v3_sched` for a packed causal batch: one entry (sequence << 20 | head << 8 | query block)
        per (sequence, head, query block), heaviest first -- query block q of a sequence attends to 2 (q + 1) key tiles, so the table goes
        level by level from the highest block index down --, and inside a level the (sequence, head) pairs are dealt to positions whose
        index mod 8 is the pair's XCD (sequence * H + head) % 8 (an XCD's L2 then serves all query blocks of a head).  numpy int32."""
        import numpy as np
        nqb = [(int(n) + block - 1) // block for n in lens]
        if max(nqb, default=0) > 256 or n_heads > 4095 or len(lens) > 2047:      # entry = sequence << 20 | head << 8 | query block
            raise ValueError("attention_schedule: at most 256 query blocks (32 768 tokens) per sequence, 4095 heads, 2047 sequences")
        out = []
        for q in range(max(nqb) - 1, -1, -1):
            buckets = [[] for _ in range(8)]
            for b, n in enumerate(nqb):
                if q < n:
                    for h in range(n_heads):
                        buckets[(b * n_heads + h) % 8].append((b << 20) | (h << 8) | q)
            # positions advance through the whole table: entry p of this level sits at len(out) + p; deal so that (len(out) + p) % 8 = bucket
            depth = max(len(x) for x in buckets)
            start = len(out) % 8
            for r in range(depth):
                for k in range(8):
                    bk = buckets[(start + k) % 8]
                    if r < len(bk):
                        out.append(bk[r])
        return np.asarray(out, np.int32)

    def attention_packed(self, qkv, n_heads, causal, cu_seqlens, n_seq, max_len, n_valid=None, window=0, out=None, rope_q=None, sched=None):
        """PACKED variable-length batch: qkv (T, 3H, hd) rows of sequence b = [cu[b], cu[b+1]) -> (T, H, hd).  Rows beyond
        cu[-1] (padding of the packed buffer) come back zero; `n_valid` = cu[-1] as a host int saves zero-filling the rest.
        `out`: a caller-owned (T, H, hd) buffer whose padding rows are already zero (the kernel writes rows < cu[-1] only)."""
        T, Ht, hd = qkv.shape
        if out is not None:
            assert out.shape == (T, n_heads, hd) and out.dtype == qkv.dtype and out.is_contiguous()
        elif n_valid is None:
            out = torch.zeros((T, n_heads, hd), dtype=qkv.dtype, device=qkv.device)
        else:
            out = torch.empty((T, n_heads, hd), dtype=qkv.dtype, device=qkv.device)
            if n_valid < T:
                out[n_valid:].zero_()
        if sched is not None:                                     # (int32 device tensor from attention_schedule: causal packed batches)
            cos, sin = rope_q if rope_q is not None else (None, None)
            assert causal and sched.dtype == torch.int32 and sched.is_cuda
            _lib.check(self.lib.d3d_flash_attention_v3_sched(_p(qkv), _p(out), n_seq, max_len, n_heads, hd, Ht * hd, 0, 0, n_heads, 2 * n_heads, 1, max_len,
                                                             _p(cu_seqlens), window, _p(cos), _p(sin), _p(sched), int(sched.numel()),
                                                             0 if qkv.dtype == torch.bfloat16 else 1, self._stream()))
            return out
        if rope_q is not None:
            cos, sin = rope_q                                     # (positions >= max_len, hd / 2) float32, contiguous
            assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
            assert cos.shape[0] >= max_len and cos.shape[1] == hd // 2 and sin.shape == cos.shape
            _lib.check(self.lib.d3d_flash_attention_v3_rope_q(_p(qkv), _p(out), n_seq, max_len, n_heads, hd, Ht * hd, 0, 0, n_heads, 2 * n_heads,
                                                              1 if causal else 0, max_len, _p(cu_seqlens), window, _p(cos), _p(sin),
                                                              0 if qkv.dtype == torch.bfloat16 else 1, self._stream()))
            return out
        _lib.check(self.lib.d3d_flash_attention_v3(_p(qkv), _p(out), n_seq, max_len, n_heads, hd, Ht * hd, 0, 0, n_heads, 2 * n_heads, 1 if causal else 0,
                                                   max_len, _p(cu_seqlens), window, 0 if qkv.dtype == torch.bfloat16 else 1, self._stream()))
        return out
