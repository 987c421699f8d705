"""ctypes wrappers of the float32 dense kernels of the 3D-token builder (csrc/f32_kernels.hip): fp32 MFMA GEMM with fused
bias / GELU / residual, the tiny-K and tiny-N linears, and LayerNorm with fused residual add / GELU.  `Mlp` and `linear`
pick the kernel by shape; weights whose K is not a multiple of 16 are zero-padded once (`prep_weight`)."""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import f32, i32, i64, vp

_lib.register("d3d_gemm_nt_f32", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp])
_lib.register("d3d_gemm_nt_f32x3", [vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i32, vp, vp, vp, vp])
_lib.register("d3d_row_exponents", [vp, i32, i32, i64, vp, vp, vp])
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
    # RANGE of the split kernel (round 5): every operand ROW is scaled by a power of two so that its largest element lies in [1, 2) before the
    # hi / lo split (`d3d_row_exponents`: once per weight -- cached next to the padded copy --, one small launch per call for the
    # activations) and the result is scaled back: no finite row overflows fp16, a row of uniformly tiny values keeps its 22 bits; inside
    # one row an element below 2^-14 of the row's maximum keeps only its hi half (absolute error <= 2^-25 x row max: invisible next to
    # the leading terms of the same dot product).  SCALE = False (D3D_F32_SCALE=0) is the unscaled round-3 kernel (|operand| < 65504).
    # A DEVICE status word collects "non-finite output" (bit 0) / "non-finite activation" (bit 1) from every launch with no host
    # synchronisation; `poll_status()` reads the copy the previous `snapshot_status()` started (the token builder: snapshot at the end
    # of an update, poll at the start of the next -- a bad GEMM is reported one step late, for free), `check_status()` synchronises.
    # D3D_F32_CHECK=1 (F32Ops.CHECK; on in the golden-trajectory GPU tests) checks every GEMM at once: a host synchronisation per call.
    SPLIT = os.environ.get("D3D_F32_SPLIT", "1") != "0"
    SCALE = os.environ.get("D3D_F32_SCALE", "1") != "0"
    CHECK = os.environ.get("D3D_F32_CHECK", "0") == "1"
    KPAD = 32                                        # both kernels take K % 32 == 0 (the float32 one needs 16)

    def __init__(self):
        self.lib = _lib.load()
        self._padded: Dict[tuple, torch.Tensor] = {}
        self._wexp: Dict[int, tuple] = {}
        self._status = None                          # device int32[1]
        self._status_host = None                     # pinned int32[1]: the last snapshot
        self._snap_event = None

    # ---- device status word ----------------------------------------------------------------------------------------------------------
    def _status_word(self, device) -> torch.Tensor:
        if self._status is None or self._status.device != device:
            self._status = torch.zeros(1, dtype=torch.int32, device=device)
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._snap_event = None
        return self._status

    def snapshot_status(self):
        """Start an asynchronous copy of the status word to pinned host memory (no synchronisation)."""
        if self._status is None:
            return
        self._status_host.copy_(self._status, non_blocking=True)
        self._snap_event = torch.cuda.Event()
        self._snap_event.record()

    def poll_status(self):
        """Raise if the last snapshot (when it has landed) recorded a non-finite float32 GEMM.  Never waits."""
        if self._snap_event is None or not self._snap_event.query():
            return
        self._snap_event = None
        self._raise_if(int(self._status_host[0]))

    def check_status(self):
        """Synchronising check (tests, end of an episode)."""
        if self._status is not None:
            self._raise_if(int(self._status.cpu()[0]))

    def _raise_if(self, bits: int):
        if bits:
            self._status.zero_()
            raise FloatingPointError("float32 token-builder GEMM (d3d_gemm_nt_f32x3): " + ", ".join(
                m for b, m in ((1, "a non-finite output element"), (2, "a non-finite activation element")) if bits & b)
                + " since the last check; D3D_F32_CHECK=1 locates the call, D3D_F32_SPLIT=0 selects the float32-MFMA kernel")

    def row_exponents(self, x: torch.Tensor, status=None) -> torch.Tensor:
        M, K = x.shape
        out = torch.empty((M,), dtype=torch.int32, device=x.device)
        _lib.check(self.lib.d3d_row_exponents(_p(x), M, K, x.stride(0), _p(out), _p(status), _stream()))
        return out

    def weight_exponents(self, wp: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """Row exponents of a (padded) weight, cached on the source tensor's storage + version like the padded copy."""
        key, tag = w.data_ptr(), (w._version, tuple(wp.shape))
        hit = self._wexp.get(key)
        if hit is None or hit[0] != tag or hit[2]() is not w:
            self._purge(self._wexp)
            hit = self._wexp[key] = (tag, self.row_exponents(wp), weakref.ref(w))
        return hit[1]

    @staticmethod
    def _purge(cache: dict):
        """Drop the entries whose source tensor is gone (their padded copies / exponent tensors would otherwise stay on the device across
        model reloads)."""
        for k in [k for k, v in cache.items() if v[2]() is None]:
            del cache[k]

    def prep_weight(self, w: torch.Tensor) -> torch.Tensor:
        """(N, K) float32 contiguous; K zero-padded to a multiple of 32 for the MFMA kernels (cached per tensor)."""
        N, K = w.shape
        if K % self.KPAD == 0 or K <= 8 or N <= 8:
            return w
        # keyed on the storage AND its version counter: an in-place update (optimizer step, load_state_dict) or a new tensor allocated
        # at a freed address must not be served another tensor's / an older padded copy; one entry per address, so updates replace
        # ... and on the tensor OBJECT (a weak reference): a new tensor of the same shape allocated at a freed address starts at version 0 too
        key = w.data_ptr()
        tag = (w._version, N, K)
        hit = self._padded.get(key)
        if hit is None or hit[0] != tag or hit[2]() is not w:
            self._purge(self._padded)
            wp = torch.zeros((N, (K + self.KPAD - 1) // self.KPAD * self.KPAD), dtype=torch.float32, device=w.device)
            wp[:, :K] = w.detach()
            hit = self._padded[key] = (tag, wp, weakref.ref(w))
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
        if not self.SPLIT:
            _lib.check(self.lib.d3d_gemm_nt_f32(_p(x), _p(wp), _p(y), _p(b), _p(residual), M, N, Kp, x.stride(0), Kp, N, EPI[epi], _stream()))
            return y
        st = self._status_word(x.device)
        ea = ew = None
        if self.SCALE:
            ew = self.weight_exponents(wp, w)
            ea = self.row_exponents(x, st)
        _lib.check(self.lib.d3d_gemm_nt_f32x3(_p(x), _p(wp), _p(y), _p(b), _p(residual), M, N, Kp, x.stride(0), Kp, N, EPI[epi], _p(ea), _p(ew), _p(st),
                                              _stream()))
        if self.CHECK and not bool(torch.isfinite(y).all()):
            raise FloatingPointError(f"d3d_gemm_nt_f32x3: non-finite output for x {tuple(x.shape)} (max |x| {float(x.abs().max()):.3g}), w {tuple(w.shape)} "
                                     f"(max |w| {float(w.abs().max()):.3g}); set D3D_F32_SPLIT=0 for the float32-MFMA kernel")
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
