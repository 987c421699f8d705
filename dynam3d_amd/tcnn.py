"""Drop-in for the reference's un-vendored CUDA dependency `tinycudann` as it is used on this path:

    tcnn.Network(n_input_dims, n_output_dims, network_config={"otype": "CutlassMLP", "activation": "LeakyReLU",
                 "output_activation": "LeakyReLU" | "None", "n_neurons": 768, "n_hidden_layers": 2})       (PRE-FF:221-243)
    y = net(x)                                                                                              (PRE-FF:484, 488)

Bias-free layers, fp16 weights and activations, fp32 accumulation, fp16 result per layer; every layer is one
`d3d_gemm_nt` launch with the LeakyReLU(0.01) fused into the epilogue (fp16 MFMA).  Output widths that are not a
multiple of 128 (the 769-wide density head) are zero-padded at load and sliced on return."""
from __future__ import annotations

from typing import List, Sequence

import torch

import ctypes as C

from . import _lib
from .hip_dense import HipDense

_lib.register("d3d_mlp768_forward", [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])


class Network:
    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, weights: Sequence[torch.Tensor] | None = None,
                 device="cuda", seed: int = 1337):
        assert network_config.get("otype", "CutlassMLP") in ("CutlassMLP", "FullyFusedMLP")
        nh, nn_ = int(network_config["n_hidden_layers"]), int(network_config["n_neurons"])
        self.dims = [n_input_dims] + [nn_] * nh + [n_output_dims]
        self.act = network_config.get("activation", "None")
        self.out_act = network_config.get("output_activation", "None")
        for a in (self.act, self.out_act):
            if a not in ("LeakyReLU", "None", "none", None):
                raise NotImplementedError(f"activation {a!r} (the reference only uses LeakyReLU / None)")
        self.device = torch.device(device)
        self.n_output_dims = n_output_dims
        self.hd = HipDense()
        if weights is None:
            g = torch.Generator().manual_seed(seed)
            weights = [torch.randn(self.dims[i + 1], self.dims[i], generator=g) * self.dims[i] ** -0.5 for i in range(len(self.dims) - 1)]
        self.load_layers(weights)

    def load_layers(self, weights: Sequence[torch.Tensor]):
        self.w: List[torch.Tensor] = []
        for i, w in enumerate(weights):
            assert tuple(w.shape) == (self.dims[i + 1], self.dims[i]), (w.shape, self.dims)
            w16 = w.detach().to(self.device, torch.float16)
            pad = (-w16.shape[0]) % 128
            if pad:
                w16 = torch.cat([w16, torch.zeros((pad, w16.shape[1]), dtype=torch.float16, device=self.device)], 0)
            self.w.append(w16.contiguous())

    @classmethod
    def from_flat_params(cls, n_in, n_out, cfg, params: torch.Tensor, **kw) -> "Network":
        """tcnn keeps ONE flat `params` vector; assumed layout = row-major [out,in] per layer, back to back (unverified:
        no tinycudann checkpoint or source is available offline)."""
        net = cls(n_in, n_out, cfg, weights=None, **kw)
        ws, o = [], 0
        for i in range(len(net.dims) - 1):
            n = net.dims[i + 1] * net.dims[i]
            ws.append(params[o:o + n].view(net.dims[i + 1], net.dims[i]))
            o += n
        net.load_layers(ws)
        return net

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """One C call (`d3d_mlp768_forward`): n_hidden + 1 fused GEMM + activation launches."""
        h = x.to(self.device, torch.float16).contiguous()
        n, nn_, n_pad = h.shape[0], self.dims[1], self.w[-1].shape[0]
        y = torch.empty((n, n_pad), dtype=torch.float16, device=self.device)
        sa, sb = (torch.empty((n, nn_), dtype=torch.float16, device=self.device) for _ in range(2))
        ptrs = (C.c_void_p * len(self.w))(*[w.data_ptr() for w in self.w])
        lk = lambda a: 1 if a == "LeakyReLU" else 0
        _lib.check(self.hd.lib.d3d_mlp768_forward(h.data_ptr(), n, self.dims[0], ptrs, len(self.w) - 1, nn_, n_pad, lk(self.act), lk(self.out_act),
                                                  sa.data_ptr(), sb.data_ptr(), y.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return y[:, : self.n_output_dims]

    __call__ = forward
