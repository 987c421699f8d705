"""Dense half of the 3D-token builder: position-embedding MLPs, the two 2-layer post-LN set
encoders and the merge discriminator (VLN-FF:134-161), batched over ALL groups of ALL
environments instead of the reference's one-encoder-call-per-segment loop (VLN-FF:580-601).

Float32 end to end (the reference's Feature_Fields parameters are fp32; merge decisions are an
argmax, so the token builder is kept at full precision -- it is <1 % of the step's FLOPs)."""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F


class FFDense:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device, n_head: int = 12):
        self.device = torch.device(device)
        self.w = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in state_dict.items()}
        self.n_head = n_head
        self.width = self.w["aggregate_patch_to_instance_embedding"].shape[-1]
        self._hd = None
        self._f32ops = None

    def _f32(self):
        if self._f32ops is None:
            from .f32_ops import F32Ops
            self._f32ops = F32Ops()
        return self._f32ops

    # nn.Sequential(Linear, LayerNorm, GELU, Linear)
    def mlp(self, x: torch.Tensor, name: str) -> torch.Tensor:
        w = self.w
        if self.device.type == "cuda":                     # fp32 MFMA GEMM + fused LN/GELU (csrc/f32_kernels.hip): 3 launches
            return self._f32().mlp(x.reshape(-1, x.shape[-1]), w, name).view(*x.shape[:-1], -1)
        h = F.linear(x, w[name + ".0.weight"], w[name + ".0.bias"])
        h = F.layer_norm(h, (h.shape[-1],), w[name + ".1.weight"], w[name + ".1.bias"], 1e-5)
        return F.linear(F.gelu(h), w[name + ".3.weight"], w[name + ".3.bias"])

    def _encoder(self, x: torch.Tensor, key_mask: torch.Tensor, name: str) -> torch.Tensor:
        """x (G,L,768), key_mask (G,L) bool (True = real token).  Post-LN TransformerEncoder x2 + final
        LayerNorm(eps=1e-12); only row 0 (CLS) of the result is used by the caller."""
        w, H = self.w, self.n_head
        G, L, D = x.shape
        am = key_mask[:, None, None, :]
        for i in range(2):
            p = f"{name}.layers.{i}"
            qkv = F.linear(x, w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"])
            q, k, v = qkv.view(G, L, 3, H, D // H).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=am)
            a = a.transpose(1, 2).reshape(G, L, D)
            a = F.linear(a, w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
            x = F.layer_norm(x + a, (D,), w[p + ".norm1.weight"], w[p + ".norm1.bias"], 1e-5)
            h = F.linear(F.gelu(F.linear(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])), w[p + ".linear2.weight"], w[p + ".linear2.bias"])
            x = F.layer_norm(x + h, (D,), w[p + ".norm2.weight"], w[p + ".norm2.bias"], 1e-5)
        return F.layer_norm(x[:, 0], (D,), w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-12)

    def encode_sets(self, emb: torch.Tensor, lens: Sequence[int], which: str) -> torch.Tensor:
        """emb (T,768): member-token embeddings of G groups laid out back to back; lens[g] members.
        Returns (G,768) = encoder([CLS; members_g])[0].

        Device path: all sets are PACKED ([CLS; members] back to back, no padding); the linears / norms run once over
        the packed tokens and self-attention runs inside each set with the varlen HIP kernel (d3d_set_attention).
        The last layer only evaluates the CLS rows (the only rows the reference keeps, VLN-FF:595)."""
        G = len(lens)
        if G == 0:
            return torch.empty((0, self.width), dtype=torch.float32, device=self.device)
        if self.device.type != "cuda":
            return self._encode_sets_padded(emb, lens, which)
        if self._hd is None:
            from .hip_dense import HipDense
            self._hd = HipDense()
        w, H, D = self.w, self.n_head, self.width
        enc = f"aggregate_{which}_encoder"
        cls = w[f"aggregate_{which}_embedding"]
        lens = np.asarray(lens, np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)])
        T = int(offs[-1])
        poff = offs + np.arange(G + 1)                                    # packed offsets (one CLS per set)
        idx = np.empty(T + G, np.int64)
        cls_rows = poff[:-1]
        idx[cls_rows] = T                                                 # row T of `src` is the CLS embedding
        member = np.ones(T + G, bool)
        member[cls_rows] = False
        idx[member] = np.arange(T)
        dev = self.device
        src = torch.cat([emb, cls], 0)
        # one upload for the three index arrays (int32: index_select takes it; every small copy is host time on the update's latency chain)
        n_idx, n_off = T + G, G + 1
        o_off, o_cls = (n_idx + 3) // 4 * 4, (n_idx + 3) // 4 * 4 + (n_off + 3) // 4 * 4
        packed = np.zeros(o_cls + G, np.int32)
        packed[:n_idx], packed[o_off:o_off + n_off], packed[o_cls:] = idx, poff, cls_rows
        packed_d = torch.from_numpy(packed).to(dev, non_blocking=True)
        x = src.index_select(0, packed_d[:n_idx])
        set_off = packed_d[o_off:o_off + n_off]
        cls_t = packed_d[o_cls:]
        max_len = int(lens.max()) + 1
        f = self._f32()
        for i in range(2):                                                # post-LN nn.TransformerEncoderLayer: 7 launches per layer
            p = f"{enc}.layers.{i}"
            last = i == 1
            qkv = f.linear(x, w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"])
            a = self._hd.set_attention(qkv, set_off, G, H, max_len, q_rows=1 if last else 0)
            if last:                                                      # only the CLS rows feed the output
                a, x = a.index_select(0, cls_t), x.index_select(0, cls_t)
            a = f.linear(a, w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
            x = f.layer_norm(a, w[p + ".norm1.weight"], w[p + ".norm1.bias"], 1e-5, residual=x)              # LN(x + attn)
            h = f.linear(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"], act="gelu")
            h = f.linear(h, w[p + ".linear2.weight"], w[p + ".linear2.bias"])
            x = f.layer_norm(h, w[p + ".norm2.weight"], w[p + ".norm2.bias"], 1e-5, residual=x)              # LN(x + ffn)
        return f.layer_norm(x, w[enc + ".norm.weight"], w[enc + ".norm.bias"], 1e-12)

    def _encode_sets_padded(self, emb: torch.Tensor, lens: Sequence[int], which: str) -> torch.Tensor:
        """Host-logic path used by the CPU tests (tests/cpu_ops.py): same arithmetic on padded buckets."""
        enc = f"aggregate_{which}_encoder"
        cls = self.w[f"aggregate_{which}_embedding"]
        G = len(lens)
        out = torch.empty((G, self.width), dtype=torch.float32, device=self.device)
        lens = np.asarray(lens, np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)])
        T = int(offs[-1])
        src = torch.cat([emb, cls, torch.zeros_like(cls)], 0)          # row T = CLS, row T+1 = zero pad
        bucket = np.where(lens <= 0, 0, np.ceil(np.log2(np.maximum(lens, 1) + 1)).astype(np.int64))
        for bk in np.unique(bucket):
            gs = np.nonzero(bucket == bk)[0]
            L = int(lens[gs].max()) + 1
            idx = np.full((len(gs), L), T + 1, np.int64)
            idx[:, 0] = T
            msk = np.zeros((len(gs), L), bool)
            msk[:, 0] = True
            for r, g in enumerate(gs):
                n = int(lens[g])
                idx[r, 1:1 + n] = np.arange(offs[g], offs[g] + n)
                msk[r, 1:1 + n] = True
            idx_t = torch.from_numpy(idx).to(self.device)
            x = src.index_select(0, idx_t.view(-1)).view(len(gs), L, self.width)
            y = self._encoder(x, torch.from_numpy(msk).to(self.device), enc)
            out.index_copy_(0, torch.from_numpy(gs).to(self.device), y)
        return out

    def encode_patch_sets(self, tok_fts: torch.Tensor, geom7: torch.Tensor, lens) -> torch.Tensor:
        emb = tok_fts + self.mlp(geom7, "patch_to_instance_position_embedding")       # VLN-FF:592
        return self.encode_sets(emb, lens, "patch_to_instance")

    def encode_zone_sets(self, inst_fts: torch.Tensor, geom4: torch.Tensor, lens) -> torch.Tensor:
        emb = inst_fts + self.mlp(geom4, "instance_to_zone_position_embedding")        # VLN-FF:725
        return self.encode_sets(emb, lens, "instance_to_zone")

    def merge_logits(self, x: torch.Tensor) -> torch.Tensor:
        return self.mlp(x, "instance_merge_discriminator")                             # VLN-FF:618
