"""Drop-in for the reference's un-vendored CUDA dependency `tinycudann` as it is used on this path:

    self.nerf_encoder = tcnn.Network(n_input_dims, n_output_dims, network_config={"otype": "CutlassMLP",
                 "activation": "LeakyReLU", "output_activation": "LeakyReLU" | "None", "n_neurons": 768,
                 "n_hidden_layers": 2})                                                                  (PRE-FF:221-243)
    y = self.nerf_encoder(x)                                                                             (PRE-FF:484, 488)

Like tinycudann's PyTorch binding, `Network` is an `nn.Module` with ONE flat trainable float32 parameter `params` (so
`state_dict()["nerf_encoder.params"]`, optimizers and DDP see what they see with tinycudann), evaluates in fp16 -- bias-free
layers y = act(x W^T), fp16 weights and activations, fp32 accumulation, an fp16 store per layer -- and is differentiable:

  forward   one `d3d_gemm_nt` launch per layer, LeakyReLU(0.01) fused into the epilogue (fp16 MFMA); inference = one C call
            (`d3d_mlp768_forward`)
  backward  per layer two more `d3d_gemm_nt` launches: the data gradient dz_{l-1} = (dz_l W_l) * act'(h_{l-1}) (epilogue 8) and
            the weight gradient dW_l = dz_l^T h_{l-1} (operands transposed once by `d3d_transpose_pad16`); gradients are fp16
            like tinycudann's and are accumulated into the float32 `params.grad`.

FLAT LAYOUT of `params` (documented assumption -- neither a tinycudann checkpoint nor its source is available offline):
layer after layer, each weight matrix row-major [out, in]; the LAST layer's rows are padded with zero rows to a multiple of
`OUTPUT_PAD` = 16 (tinycudann pads the output width to the tensor-core granularity: 769 -> 784), i.e.
n_params = sum_l out_l_padded * in_l.  `load_flat_params` / `flat_from_layers` convert both ways."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _lib
from .hip_dense import HipDense

_lib.register("d3d_mlp768_forward", [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_lib.register("d3d_mlp_fused", [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int32, C.c_void_p])
_lib.register("d3d_transpose_pad16", [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_void_p])
_lib.register("d3d_lrelu_bwd", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p])

OUTPUT_PAD = 16      # tinycudann: padded output width = next multiple of the tensor-core width
GEMM_PAD = 128       # d3d_gemm_nt: N % 128 == 0
LOSS_SCALE = 128.0   # tinycudann's PyTorch binding multiplies dL/dy by `loss_scale` (128 for fp16 networks) before the fp16 backward and
                     # divides the float32 gradients by it; the reference trains under autocast WITHOUT a GradScaler (PRE-TR:501-512) and
                     # relies on exactly that: the mean-reduced cosine / InfoNCE losses give per-element gradients of 1e-5 .. 1e-7, which
                     # are subnormal or zero in fp16


FUSED_MAX_WIDTH = 896       # csrc/mlp_kernels.hip: the 64-row activation slab lives in LDS
import os as _os
FUSED = _os.environ.get("D3D_MLP_FUSED", "0") == "1"     # True: one d3d_mlp_fused launch per network (forward and the backward's data-gradient
                                                         # chain); default: one d3d_gemm_nt launch per layer -- measured 2.5-3x FASTER (119 vs 37 us
                                                         # at 1 152 rows: csrc/mlp_kernels.hip header).  Both are bit-identical and both are tested.


def fused_ok(widths) -> bool:
    return FUSED and len(widths) - 1 <= 4 and all(w <= FUSED_MAX_WIDTH and w % 16 == 0 for w in widths) and all(w % 32 == 0 for w in widths[:-1])


def mlp_fused(lib, x, weights, modes, aux=None, save=None):
    """One launch of d3d_mlp_fused over the chain `weights` ([out_l, in_l] fp16 each).  modes[l]: 0 / 1 (LeakyReLU) / 2 (x LeakyReLU'(aux[l])).
    save[l] True -> layer l's output is also written to HBM and returned (the last layer's always is).  Returns the list of saved outputs."""
    n, L = x.shape[0], len(weights)
    widths = [x.shape[1]] + [int(w.shape[0]) for w in weights]
    save = [bool(save[l]) if save is not None else False for l in range(L)]
    save[-1] = True
    outs = [torch.empty((n, widths[l + 1]), dtype=x.dtype, device=x.device) if save[l] else None for l in range(L)]
    aux = aux or [None] * L
    i32a = lambda v: (C.c_int32 * len(v))(*v)
    i64a = lambda v: (C.c_int64 * len(v))(*v)
    pa = lambda ts: (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
    _lib.check(lib.d3d_mlp_fused(x.data_ptr(), x.stride(0), n, L, i32a(widths), pa(weights), i32a(modes), pa(aux), i64a([0 if t is None else t.stride(0) for t in aux]),
                                 pa(outs), i64a([0 if t is None else t.stride(0) for t in outs]), 0 if x.dtype == torch.bfloat16 else 1, _stream()))
    return outs


def _stream():
    return _lib.current_stream_ptr()


def _p(t):
    return C.c_void_p(t.data_ptr())


class _MlpFunction(torch.autograd.Function):
    """y = MLP(x; params) on the HIP kernels, forward and backward (see module docstring)."""

    @staticmethod
    def forward(ctx, x, params, net):
        hd = net.hd
        ws = net._layer_weights()                                    # fp16 [out_pad128, in] per layer (views of one cache)
        h = x.detach().to(net.device, torch.float16).contiguous()
        acts = [h]
        L = len(ws)
        if fused_ok([h.shape[1]] + [int(w.shape[0]) for w in ws]):
            # ONE launch: the slab of activations stays in LDS across the layers, every layer's output is also saved for the backward pass
            modes = [1 if (net.out_act if l == L - 1 else net.act) == "LeakyReLU" else 0 for l in range(L)]
            acts += mlp_fused(hd.lib, h, ws, modes, save=[True] * L)
            h = acts[-1]
        else:
            for l, w in enumerate(ws):
                act = net.out_act if l == L - 1 else net.act
                h = hd.gemm(h, w, None, None, "lrelu" if act == "LeakyReLU" else "none")
                acts.append(h)
        ctx.net = net
        ctx.save_for_backward(*acts)                                 # (saved tensors: autograd detects `params` / activations modified in between)
        ctx.x_dtype = x.dtype
        # `float32_grad_io`: hand the (fp16-valued) result out as float32, so that the gradient arrives here in float32 and is scaled by
        # LOSS_SCALE BEFORE its cast to fp16 -- through an fp16 output tensor autograd would deliver an fp16 gradient, in which the 1e-6-sized
        # entries of a mean-reduced loss are already flushed
        return h[:, : net.n_output_dims].float() if net.float32_grad_io else h[:, : net.n_output_dims]

    @staticmethod
    def backward(ctx, dy):
        net, acts, hd = ctx.net, list(ctx.saved_tensors), ctx.net.hd
        lib = hd.lib
        ws_t = net._layer_weights_t()                                # fp16 [in, out_pad128] per layer = W^T, the NT operand of dz W
        M = acts[0].shape[0]
        Mp = (M + 63) // 64 * 64
        L = len(ws_t)
        n_last = ws_t[-1].shape[1]
        dz = torch.zeros((M, n_last), dtype=torch.float16, device=net.device)
        dz[:, : net.n_output_dims] = (dy.float() * LOSS_SCALE).to(torch.float16)      # scaled into fp16's range (see LOSS_SCALE)
        if net.out_act == "LeakyReLU":
            _lib.check(lib.d3d_lrelu_bwd(_p(dz), _p(acts[-1]), _p(dz), dz.numel(), 1, _stream()))
        grads: List[torch.Tensor] = [None] * L
        # data gradients dz_l (the gradient at layer l's OUTPUT): dz_{l-1} = (dz_l W_l) * act'(h_{l-1}).  Fused: the whole chain in one
        # launch (weights = the transposed matrices, mode 2 takes the slope from the saved activation), every dz_l stored for dW_l below.
        dzs = {L - 1: dz}
        need_dx = bool(ctx.needs_input_grad[0])
        chain = list(range(L - 1, 0 if need_dx else 0, -1)) + ([0] if need_dx else [])
        fused_bwd = bool(chain) and fused_ok([dz.shape[1]] + [int(ws_t[l].shape[0]) for l in chain])
        if fused_bwd:
            modes = [(2 if net.act == "LeakyReLU" else 0) if l > 0 else 0 for l in chain]
            outs = mlp_fused(hd.lib, dz, [ws_t[l] for l in chain], modes, aux=[acts[l] if l > 0 else None for l in chain], save=[True] * len(chain))
            for l, o in zip(chain, outs):
                dzs[l - 1] = o                                            # (dzs[-1] = dx, padded to the tile)
        for l in range(L - 1, -1, -1):
            dz = dzs[l] if fused_bwd else dz
            h_prev = acts[l]                                          # input of layer l (= x for l = 0)
            N, K = dz.shape[1], h_prev.shape[1]
            Kp = (K + GEMM_PAD - 1) // GEMM_PAD * GEMM_PAD           # h^T is the GEMM's [N, K] operand: rows padded to the tile (a 64- or
            dz_t = torch.empty((N, Mp), dtype=torch.float16, device=net.device)      # 192-wide input layer would otherwise fail here)
            h_t = torch.empty((K, Mp), dtype=torch.float16, device=net.device) if Kp == K else torch.zeros((Kp, Mp), dtype=torch.float16, device=net.device)
            _lib.check(lib.d3d_transpose_pad16(_p(dz), _p(dz_t), M, N, dz.stride(0), Mp, _stream()))
            _lib.check(lib.d3d_transpose_pad16(_p(h_prev), _p(h_t), M, K, h_prev.stride(0), Mp, _stream()))
            grads[l] = hd.gemm(dz_t, h_t, None, None, "none")[:, :K]  # dW_l (N, K) = dz^T h_prev
            if fused_bwd:
                continue
            if l > 0:
                dz = hd.gemm(dz, ws_t[l], None, h_prev, "lrelu_bwd") if net.act == "LeakyReLU" else hd.gemm(dz, ws_t[l], None, None, "none")
            elif ctx.needs_input_grad[0]:
                dz = hd.gemm(dz, ws_t[0], None, None, "none")[:, : net.n_input_dims]        # dx
        if fused_bwd and need_dx:
            dz = dzs[-1][:, : net.n_input_dims]
        dx = (dz.float() / LOSS_SCALE).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        dparams = net._flat_grad(grads) if ctx.needs_input_grad[1] else None
        return dx, dparams, None


class Network(torch.nn.Module):
    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, weights: Sequence[torch.Tensor] | None = None,
                 device="cuda", seed: int = 1337):
        super().__init__()
        assert network_config.get("otype", "CutlassMLP") in ("CutlassMLP", "FullyFusedMLP")
        nh, nn_ = int(network_config["n_hidden_layers"]), int(network_config["n_neurons"])
        self.dims = [n_input_dims] + [nn_] * nh + [n_output_dims]
        self.act = network_config.get("activation", "None")
        self.out_act = network_config.get("output_activation", "None")
        for a in (self.act, self.out_act):
            if a not in ("LeakyReLU", "None", "none", None):
                raise NotImplementedError(f"activation {a!r} (the reference only uses LeakyReLU / None)")
        if n_input_dims % 64 or nn_ % 128:
            raise ValueError("the MFMA kernels need n_input_dims % 64 == 0 and n_neurons % 128 == 0 (the reference uses 768)")
        self.device = torch.device(device)
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.float32_grad_io = False             # see _MlpFunction.forward
        self.hd = HipDense()
        # rows of each layer in the flat vector / in the GEMM operand
        self._rows_flat = [self.dims[i + 1] for i in range(nh)] + [(n_output_dims + OUTPUT_PAD - 1) // OUTPUT_PAD * OUTPUT_PAD]
        self._rows_gemm = [self.dims[i + 1] for i in range(nh)] + [(n_output_dims + GEMM_PAD - 1) // GEMM_PAD * GEMM_PAD]
        n_params = sum(r * self.dims[i] for i, r in enumerate(self._rows_flat))
        self.params = torch.nn.Parameter(torch.zeros(n_params, dtype=torch.float32, device=self.device))
        self._cache_version = None
        if weights is None:
            g = torch.Generator().manual_seed(seed)
            weights = [torch.randn(self.dims[i + 1], self.dims[i], generator=g) * self.dims[i] ** -0.5 for i in range(len(self.dims) - 1)]
        self.load_layers(weights)

    # ---- flat parameter vector <-> per-layer matrices -------------------------------------------------------------------
    def flat_from_layers(self, weights: Sequence[torch.Tensor]) -> torch.Tensor:
        chunks = []
        for i, w in enumerate(weights):
            assert tuple(w.shape) == (self.dims[i + 1], self.dims[i]), (tuple(w.shape), self.dims)
            m = torch.zeros((self._rows_flat[i], self.dims[i]), dtype=torch.float32)
            m[: w.shape[0]] = w.detach().float().cpu()
            chunks.append(m.reshape(-1))
        return torch.cat(chunks)

    def layers_from_flat(self, flat: torch.Tensor) -> List[torch.Tensor]:
        out, o = [], 0
        for i, r in enumerate(self._rows_flat):
            n = r * self.dims[i]
            out.append(flat[o:o + n].view(r, self.dims[i])[: self.dims[i + 1]])
            o += n
        return out

    @torch.no_grad()
    def load_layers(self, weights: Sequence[torch.Tensor]):
        self.params.copy_(self.flat_from_layers(weights).to(self.device))
        self._cache_version = None

    @torch.no_grad()
    def load_flat_params(self, params: torch.Tensor):
        """A tinycudann `params` vector (see FLAT LAYOUT in the module docstring)."""
        if params.numel() != self.params.numel():
            raise ValueError(f"flat params: expected {self.params.numel()} values for dims {self.dims} (output rows padded to {OUTPUT_PAD}), got {params.numel()}")
        self.params.copy_(params.detach().reshape(-1).to(self.device, torch.float32))
        self._cache_version = None

    @classmethod
    def from_flat_params(cls, n_in, n_out, cfg, params: torch.Tensor, **kw) -> "Network":
        net = cls(n_in, n_out, cfg, weights=None, **kw)
        net.load_flat_params(params)
        return net

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._cache_version = None

    # ---- fp16 operand caches (rebuilt when `params` changed: optimizer step, load_state_dict) ----------------------------
    def _refresh(self):
        ver = (self.params._version, self.params.data_ptr())
        if self._cache_version == ver:
            return
        ws, ws_t = [], []
        for i, w in enumerate(self.layers_from_flat(self.params.detach())):
            w16 = torch.zeros((self._rows_gemm[i], self.dims[i]), dtype=torch.float16, device=self.device)
            w16[: w.shape[0]] = w.to(torch.float16)
            ws.append(w16)
            wt = w16.t().contiguous()                                # [in, out_pad]: the NT operand of dz W_l (N = in -> rows padded to the tile)
            if wt.shape[0] % GEMM_PAD:
                wt = torch.cat([wt, torch.zeros((GEMM_PAD - wt.shape[0] % GEMM_PAD, wt.shape[1]), dtype=wt.dtype, device=wt.device)])
            ws_t.append(wt)
        self._w, self._w_t, self._cache_version = ws, ws_t, ver

    def _layer_weights(self):
        self._refresh()
        return self._w

    def _layer_weights_t(self):
        self._refresh()
        return self._w_t

    @property
    def w(self):                       # the renderer's one-call forward reads the fp16 operands
        return self._layer_weights()

    def _flat_grad(self, grads: Sequence[torch.Tensor]) -> torch.Tensor:
        out = torch.zeros_like(self.params)
        o = 0
        for i, r in enumerate(self._rows_flat):
            n = r * self.dims[i]
            g = grads[i][: self.dims[i + 1]].float() / LOSS_SCALE
            out[o:o + g.numel()] = g.reshape(-1)
            o += n
        return out

    # ---- forward ------------------------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(n, n_input_dims) -> (n, n_output_dims) fp16.  With autograd on: per-layer launches with saved activations; without:
        one C call (`d3d_mlp768_forward`)."""
        if torch.is_grad_enabled() and (x.requires_grad or self.params.requires_grad):
            return _MlpFunction.apply(x, self.params, self)
        ws = self._layer_weights()
        h = x.detach().to(self.device, torch.float16).contiguous()
        if fused_ok([h.shape[1]] + [int(w.shape[0]) for w in ws]):
            modes = [1 if (self.out_act if l == len(ws) - 1 else self.act) == "LeakyReLU" else 0 for l in range(len(ws))]
            return mlp_fused(self.hd.lib, h, ws, modes)[-1][:, : self.n_output_dims]
        n, nn_, n_pad = h.shape[0], self.dims[1], ws[-1].shape[0]
        y = torch.empty((n, n_pad), dtype=torch.float16, device=self.device)
        sa, sb = (torch.empty((n, nn_), dtype=torch.float16, device=self.device) for _ in range(2))
        ptrs = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        lk = lambda a: 1 if a == "LeakyReLU" else 0
        _lib.check(self.hd.lib.d3d_mlp768_forward(h.data_ptr(), n, self.dims[0], ptrs, len(ws) - 1, nn_, n_pad, lk(self.act), lk(self.out_act),
                                                  sa.data_ptr(), sb.data_ptr(), y.data_ptr(), _stream()))
        return y[:, : self.n_output_dims]
