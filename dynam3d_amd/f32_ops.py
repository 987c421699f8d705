"""ctypes wrappers of the float32 dense kernels of the 3D-token builder (csrc/f32_kernels.hip): fp32 MFMA GEMM with fused
bias / GELU / residual, the tiny-K and tiny-N linears, and LayerNorm with fused residual add / GELU.  `Mlp` and `linear`
pick the kernel by shape; weights whose K is not a multiple of 16 are zero-padded once (`prep_weight`)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import f32, i32, i64, vp

_lib.register("d3d_gemm_nt_f32", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp])
_lib.register("d3d_gemm_nt_f32x3", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp])
_lib.register("d3d_linear_smallk_f32", [vp, vp, vp, vp, i32, i32, i32, i64, i64, i32, vp])
_lib.register("d3d_linear_smalln_f32", [vp, vp, vp, vp, i32, i32, i32, i64, i64, vp])
_lib.register("d3d_layer_norm_f32", [vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, f32, i32, vp])

EPI = dict(none=0, bias=1, bias_gelu=2, bias_res=3)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return _lib.current_stream_ptr()


class F32Ops:
    # The float32 GEMMs run as split-precision fp16 MFMAs (hi hi + hi lo + lo hi, float32 accumulation: csrc/f32x3_kernels.hip) --
    # float32 accuracy at 16-bit matrix rate.  D3D_F32_SPLIT=0 (or F32Ops.SPLIT = False) selects the v_mfma_f32_16x16x4_f32 kernel.
    # RANGE CONTRACT of the split kernel: |operand| < 65504 (the hi half is an fp16), and an element below ~2^-13 keeps only its hi half
    # (its lo half falls into fp16's subnormals): absolute error <= 2^-25 per such element, invisible next to O(1) elements of the same
    # dot product -- which is what the token builder feeds it (LayerNorm'ed activations, unit-norm CLIP features, metres).  D3D_F32_CHECK=1
    # (or F32Ops.CHECK = True; on in the golden-trajectory GPU tests) verifies every output of the split kernel to be finite -- a host
    # synchronisation per GEMM, so a debug switch, not the default.
    SPLIT = os.environ.get("D3D_F32_SPLIT", "1") != "0"
    CHECK = os.environ.get("D3D_F32_CHECK", "0") == "1"
    KPAD = 32                                        # both kernels take K % 32 == 0 (the float32 one needs 16)

    def __init__(self):
        self.lib = _lib.load()
        self._padded: Dict[tuple, torch.Tensor] = {}

    def prep_weight(self, w: torch.Tensor) -> torch.Tensor:
        """(N, K) float32 contiguous; K zero-padded to a multiple of 32 for the MFMA kernels (cached per tensor)."""
        N, K = w.shape
        if K % self.KPAD == 0 or K <= 8 or N <= 8:
            return w
        # keyed on the storage AND its version counter: an in-place update (optimizer step, load_state_dict) or a new tensor allocated
        # at a freed address must not be served another tensor's / an older padded copy; one entry per address, so updates replace
        key = w.data_ptr()
        tag = (w._version, N, K)
        hit = self._padded.get(key)
        if hit is None or hit[0] != tag:
            wp = torch.zeros((N, (K + self.KPAD - 1) // self.KPAD * self.KPAD), dtype=torch.float32, device=w.device)
            wp[:, :K] = w.detach()
            hit = self._padded[key] = (tag, wp)
        return hit[1]

    def linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, act: Optional[str] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y = act(x w^T + b) [+ residual]; x (M, K) float32 (row stride % 4 == 0), w (N, K)."""
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        if M == 0:
            return y
        if x.stride(1) != 1:
            x = x.contiguous()
        if K <= 8:
            assert residual is None
            _lib.check(self.lib.d3d_linear_smallk_f32(_p(x), _p(w), _p(b), _p(y), M, N, K, x.stride(0), N, 1 if act == "gelu" else 0, _stream()))
            return y
        if N <= 8:
            assert act is None and residual is None
            if x.stride(0) % 4:
                x = x.contiguous()
            _lib.check(self.lib.d3d_linear_smalln_f32(_p(x), _p(w), _p(b), _p(y), M, N, K, x.stride(0), N, _stream()))
            return y
        wp = self.prep_weight(w)
        Kp = wp.shape[1]
        if Kp != K or x.stride(0) % 4:                      # activations with an odd K (the 1539-wide merge input): zero-padded copy
            xp = torch.zeros((M, Kp), dtype=torch.float32, device=x.device)
            xp[:, :K] = x
            x = xp
        if residual is not None:
            residual = residual.contiguous()
            epi = "bias_res"
        else:
            epi = "bias_gelu" if act == "gelu" else "bias"
        gemm = self.lib.d3d_gemm_nt_f32x3 if self.SPLIT else self.lib.d3d_gemm_nt_f32
        _lib.check(gemm(_p(x), _p(wp), _p(y), _p(b), _p(residual), M, N, Kp, x.stride(0), Kp, N, EPI[epi], _stream()))
        if self.CHECK and self.SPLIT and not bool(torch.isfinite(y).all()):
            raise FloatingPointError(f"d3d_gemm_nt_f32x3: non-finite output for x {tuple(x.shape)} (max |x| {float(x.abs().max()):.3g}), w {tuple(w.shape)} "
                                     f"(max |w| {float(w.abs().max()):.3g}): operands outside the split kernel's fp16 range; set D3D_F32_SPLIT=0")
        return y

    def layer_norm(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None, gelu: bool = False) -> torch.Tensor:
        """[gelu](LayerNorm(x [+ residual])), rows of D <= 3072."""
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or x2.stride(0) % 4:
            x2 = x2.contiguous()
        rows, D = x2.shape
        y = torch.empty((rows, D), dtype=torch.float32, device=x.device)
        if rows == 0:
            return y.view(x.shape)
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, D)
            if r2.stride(1) != 1 or r2.stride(0) % 4:
                r2 = r2.contiguous()
        _lib.check(self.lib.d3d_layer_norm_f32(_p(x2), _p(r2), _p(w), _p(b), _p(y), rows, D, x2.stride(0), 0 if r2 is None else r2.stride(0), D, eps,
                                               1 if gelu else 0, _stream()))
        return y.view(x.shape)

    def mlp(self, x: torch.Tensor, w: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
        """nn.Sequential(Linear, LayerNorm, GELU, Linear) (VLN-FF:139-161, VLN-POL:83-111): 3 launches."""
        h = self.linear(x, w[name + ".0.weight"], w[name + ".0.bias"])
        h = self.layer_norm(h, w[name + ".1.weight"], w[name + ".1.bias"], 1e-5, gelu=True)
        return self.linear(h, w[name + ".3.weight"], w[name + ".3.bias"])
