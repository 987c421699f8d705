"""Pre-training step of the 3D feature field: the `is_training=True` branch of the Pretrain class's
`update_feature_fields` (Dynam3D_Pretrain/src_3dff/models/feature_fields.py "PRE-FF":843-1345) and the trainer's optimisation step
(Dynam3D_Pretrain/src_3dff/ss_trainer_3DFF.py "PRE-TR":479-526), on the same device-resident memory as inference.

What trains: the patch->instance set encoder + its position embedding, the instance->zone set encoder + its position embedding and the
merge discriminator (PRE-FF:134-161).  Per view and environment the reference
  * encodes every 2D segment (PRE-FF:940-966) and aligns it with the mean CLIP feature of its patches, plainly and in the frame's
    mean-centred subspace (PRE-FF:969-974),
  * labels every segment with a ground-truth instance id: k = 1 nearest GT point per patch (`gt_pcd_tree.query`, PRE-FF:977-983; here
    `d3d_knn` over the 1e5-1e6 GT points), majority vote,
  * encodes the frame's instances as one zone and aligns it with the image-level CLIP feature (PRE-FF:989-1008),
  * scores every (segment, proposal) pair with the merge discriminator and trains it with a class-balanced cross-entropy against
    "same GT id" (PRE-FF:1029-1047); the memory update then MERGES BY GROUND TRUTH, not by the discriminator's argmax (PRE-FF:1035),
  * sums contrastive / 5 + cosine terms into `sim_loss` (PRE-FF:1302-1330).
`TrainableFF` evaluates exactly those expressions differentiably: every Linear runs forward AND backward on the float32 MFMA GEMM
(`d3d_gemm_nt_f32`: y = x W^T, dx = dy W, dW = dy^T x), LayerNorm / GELU / the small set attention are PyTorch autograd expressions.
`FFTrainer` plugs into `Feature_Fields.update_feature_fields(is_training=True, trainer=...)` at the two points where the reference's
loss terms are born and hands detached features / ground-truth merge decisions back to the (unchanged) memory state machine.
`pretrain_step` = PRE-TR:479-526: zero_grad, forward, NaN vote across ranks, backward, per-parameter NaN scrub, clip_grad_value_(10),
bucketed gradient all-reduce (dist.all_reduce_gradients replaces DDP), optimizer step, weights re-synced into the inference path.
Dropout (0.1 inside nn.TransformerEncoderLayer while the reference trains) is not applied: the step is deterministic."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from . import dist as DD
from . import losses as LS
from . import train_ops as TO


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return _lib.current_stream_ptr()


TRAIN_GEMM_SPLIT = __import__("os").environ.get("D3D_TRAIN_GEMM_SPLIT", "1") != "0"


def gemm_nt_f32(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a (M, K) @ w (N, K)^T -> (M, N) float32 (K / N zero-padded: transient copies, nothing cached -- the operands of a training step
    change every step).  Default: the split-precision kernel d3d_gemm_nt_f32x3 (fp16 hi + lo, three MFMAs, float32 accumulation: float32
    accuracy at 16-bit matrix rate) with BOTH operands row-scaled by d3d_row_exponents -- gradients are routinely 1e-6 .. 1e-9, far below
    fp16's normal range, and the power-of-two row scale makes the split exact for them too.  D3D_TRAIN_GEMM_SPLIT=0: d3d_gemm_nt_f32
    (v_mfma_f32_16x16x4_f32), the kernel of rounds 3-4."""
    from . import f32_ops  # noqa: F401  (registers the signatures)
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    if M == 0 or N == 0:
        return torch.zeros((M, N), dtype=torch.float32, device=a.device)
    kq = 32 if TRAIN_GEMM_SPLIT else 16
    Kp, Np = (K + kq - 1) // kq * kq, (N + 3) // 4 * 4
    if Kp != K or a.stride(1) != 1 or a.stride(0) % 4:
        ap = torch.zeros((M, Kp), dtype=torch.float32, device=a.device)
        ap[:, :K] = a
        a = ap
    if Kp != K or Np != N or not w.is_contiguous():
        wp = torch.zeros((Np, Kp), dtype=torch.float32, device=w.device)
        wp[:N, :K] = w
        w = wp
    y = torch.empty((M, Np), dtype=torch.float32, device=a.device)
    if TRAIN_GEMM_SPLIT:
        ea = torch.empty((M,), dtype=torch.int32, device=a.device)
        ew = torch.empty((Np,), dtype=torch.int32, device=a.device)
        _lib.check(lib.d3d_row_exponents(_p(a), M, Kp, a.stride(0), _p(ea), None, _stream()))
        _lib.check(lib.d3d_row_exponents(_p(w), Np, Kp, Kp, _p(ew), None, _stream()))
        _lib.check(lib.d3d_gemm_nt_f32x3(_p(a), _p(w), _p(y), None, None, M, Np, Kp, a.stride(0), Kp, Np, 0, _p(ea), _p(ew), None, _stream()))
    else:
        _lib.check(lib.d3d_gemm_nt_f32(_p(a), _p(w), _p(y), None, None, M, Np, Kp, a.stride(0), Kp, Np, 0, _stream()))
    return y[:, :N]


class _LinearF32(torch.autograd.Function):
    """y = x W^T + b with all three GEMMs (forward, dx, dW) on the float32 MFMA kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        y = gemm_nt_f32(x, w)
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm_nt_f32(dy, w.t().contiguous()) if ctx.needs_input_grad[0] else None        # dy (M,N) @ W (N,K)
        dw = gemm_nt_f32(dy.t().contiguous(), x.t().contiguous()) if ctx.needs_input_grad[1] else None   # dy^T (N,M) @ x (M,K)
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def lin(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    if not x.is_cuda:
        return F.linear(x, w, b)
    return _LinearF32.apply(x.reshape(-1, x.shape[-1]), w, b).view(*x.shape[:-1], w.shape[0])


class TrainableFF(torch.nn.Module):
    """The trainable parameters of `Feature_Fields` under the reference's state-dict keys, with differentiable evaluations."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", n_head: int = 12):
        super().__init__()
        self.names = [k for k in state_dict if not k.startswith(("FastSAM", "nerf_", "patch_to_nerf_", "aggregate_patch_to_nerf_"))]
        self.plist = torch.nn.ParameterList([torch.nn.Parameter(state_dict[k].detach().to(device, torch.float32).clone()) for k in self.names])
        self.n_head = n_head

    @property
    def w(self) -> Dict[str, torch.Tensor]:
        return dict(zip(self.names, self.plist))

    def named_state(self) -> Dict[str, torch.Tensor]:
        return {k: p.detach() for k, p in zip(self.names, self.plist)}

    def mlp(self, x, name):                                     # nn.Sequential(Linear, LayerNorm, GELU, Linear)
        w = self.w
        h = lin(x, w[name + ".0.weight"], w[name + ".0.bias"])
        h = TO.layer_norm(h, w[name + ".1.weight"], w[name + ".1.bias"], 1e-5, gelu=True)        # LayerNorm + GELU: one kernel each way
        return lin(h, w[name + ".3.weight"], w[name + ".3.bias"])

    def encoder_packed(self, x, set_off, lens, cls_rows, name):
        """The same post-LN encoder over PACKED sets (device path): x (T, D) = the sets' tokens back to back, CLS row first; `set_off`
        (G + 1,) int32 on the device, `lens` (G,) on the host, `cls_rows` (G,) int64 on the device.  Linears on the float32 MFMA GEMM,
        LayerNorm / GELU / the variable-length set attention on their own forward + backward kernels (train_ops)."""
        w, H = self.w, self.n_head
        for i in range(2):
            p = f"{name}.layers.{i}"
            last = i == 1
            qkv = lin(x, w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"])
            a = TO.set_attention(qkv, set_off, lens, H, q_rows=1 if last else 0)
            if last:                                            # only the CLS rows feed the output (PRE-FF:966 `[0]`)
                a, x = a.index_select(0, cls_rows), x.index_select(0, cls_rows)
            a = lin(a, w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
            x = TO.layer_norm(x + a, w[p + ".norm1.weight"], w[p + ".norm1.bias"], 1e-5)
            h = lin(TO.gelu(lin(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])), w[p + ".linear2.weight"], w[p + ".linear2.bias"])
            x = TO.layer_norm(x + h, w[p + ".norm2.weight"], w[p + ".norm2.bias"], 1e-5)
        return TO.layer_norm(x, w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-12)

    def encoder(self, x, key_mask, name):
        """x (G, L, D), key_mask (G, L) True = real token -> (G, D): post-LN TransformerEncoder x 2 + LayerNorm(1e-12), row 0 (PRE-FF:134-155)."""
        w, H = self.w, self.n_head
        G, L, D = x.shape
        am = key_mask[:, None, None, :]
        for i in range(2):
            p = f"{name}.layers.{i}"
            qkv = lin(x, w[p + ".self_attn.in_proj_weight"], w[p + ".self_attn.in_proj_bias"])
            q, k, v = qkv.view(G, L, 3, H, D // H).permute(2, 0, 3, 1, 4)
            a = F.scaled_dot_product_attention(q, k, v, attn_mask=am).transpose(1, 2).reshape(G, L, D)
            a = lin(a, w[p + ".self_attn.out_proj.weight"], w[p + ".self_attn.out_proj.bias"])
            x = F.layer_norm(x + a, (D,), w[p + ".norm1.weight"], w[p + ".norm1.bias"], 1e-5)
            h = lin(F.gelu(lin(x, w[p + ".linear1.weight"], w[p + ".linear1.bias"])), w[p + ".linear2.weight"], w[p + ".linear2.bias"])
            x = F.layer_norm(x + h, (D,), w[p + ".norm2.weight"], w[p + ".norm2.bias"], 1e-5)
        return F.layer_norm(x[:, 0], (D,), w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-12)

    def encode_sets(self, emb, lens, which):
        """emb (T, D) member tokens of G sets back to back -> (G, D) = encoder([CLS; members])[0]; sets padded per length bucket."""
        enc, cls = f"aggregate_{which}_encoder", self.w[f"aggregate_{which}_embedding"]
        lens = np.asarray(lens, np.int64)
        G, D = len(lens), emb.shape[-1]
        offs = np.concatenate([[0], np.cumsum(lens)])
        T = int(offs[-1])
        if emb.is_cuda and G > 0:
            # packed: [CLS; members] of every set back to back, no padding (the layout of the inference path, ff_dense.FFDense.encode_sets)
            poff = offs + np.arange(G + 1)
            idx = np.empty(T + G, np.int64)
            idx[poff[:-1]] = T                                              # row T of `src` is the CLS embedding
            member = np.ones(T + G, bool)
            member[poff[:-1]] = False
            idx[member] = np.arange(T)
            dev = emb.device
            x = torch.cat([emb, cls], 0).index_select(0, torch.from_numpy(idx).to(dev))
            return self.encoder_packed(x, torch.from_numpy(poff.astype(np.int32)).to(dev), lens + 1, torch.from_numpy(poff[:-1].astype(np.int64)).to(dev), enc)
        src = torch.cat([emb, cls, torch.zeros_like(cls)], 0)          # row T = CLS, row T + 1 = padding
        bucket = np.ceil(np.log2(np.maximum(lens, 1) + 1)).astype(np.int64)
        outs, order = [], []
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
            x = src.index_select(0, torch.from_numpy(idx).to(emb.device).view(-1)).view(len(gs), L, D)
            outs.append(self.encoder(x, torch.from_numpy(msk).to(emb.device), enc))
            order.append(gs)
        inv = np.argsort(np.concatenate(order))
        return torch.cat(outs, 0).index_select(0, torch.from_numpy(inv).to(emb.device))

    def encode_patch_sets(self, tok_fts, geom7, lens):
        return self.encode_sets(tok_fts + self.mlp(geom7, "patch_to_instance_position_embedding"), lens, "patch_to_instance")

    def encode_zone_sets(self, inst_fts, geom4, lens):
        return self.encode_sets(inst_fts + self.mlp(geom4, "instance_to_zone_position_embedding"), lens, "instance_to_zone")

    def merge_logits(self, x):
        return self.mlp(x, "instance_merge_discriminator")


class FFTrainer:
    """Collects the loss terms of one `update_feature_fields(is_training=True)` call (all views, all environments)."""

    def __init__(self, model: TrainableFF, gt_xyz: Optional[Sequence[np.ndarray]] = None, gt_label: Optional[Sequence[np.ndarray]] = None):
        self.model = model
        self.set_gt(gt_xyz, gt_label)
        self.gt_rows: Optional[torch.Tensor] = None               # (slots, m_cap) int64 ON THE DEVICE: GT instance id of every instance row
        self.begin()                                               # (PRE-FF global_gt_instance_ids), -1 = none yet

    @property
    def gt_ids_of_slot(self) -> Dict[int, torch.Tensor]:
        """memory slot -> (m_cap,) GT ids on the host (a view for tests / inspection; the working copy is `gt_rows`)."""
        return {} if self.gt_rows is None else {s: self.gt_rows[s].cpu() for s in range(self.gt_rows.shape[0])}

    def set_gt(self, gt_xyz, gt_label):
        """Ground-truth point clouds per environment: xyz (Ng, 3) float32 world coordinates, labels (Ng,) int (PRE-FF gt_pcd_tree / gt_pcd_label)."""
        self.gt_xyz, self.gt_label = gt_xyz, gt_label
        self._gt_dev = None
        self._gt_lab_dev = None

    def begin(self):
        self.pred_i, self.tgt_i, self.pred_is, self.tgt_is = [], [], [], []
        self.pred_z, self.tgt_z, self.pred_zs, self.tgt_zs = [], [], [], []
        self.segm_terms: List[torch.Tensor] = []
        self.debug: List[dict] = []

    # ---- hook 1: the 2D instances of one view of every environment -------------------------------------------------------------------------
    # (`Feature_Fields.update_feature_fields` runs under torch.no_grad(): the hooks re-enable autograd for the loss terms)
    @torch.enable_grad()
    def instances(self, ff, ops, pools, envs, slots_h, row_base, order, tok_fts, geom7, counts, valid_g, centroid, n_max, image_ft_ix, image_ft_mean):
        """tok_fts (B*P, 768) patch features ordered (env, segment label, patch), geom7 (B*P, 7), counts (B, n_max) patches per segment,
        valid_g: flat indices of the non-empty groups, centroid (B*n_max, 3).  image_ft_ix (B, 768) / image_ft_mean (B, 768): CLIP image
        feature of this view / mean over the views (or None).  Returns the detached (len(valid_g), 768) instance features."""
        m, dev = self.model, tok_fts.device
        B, P = len(envs), ff.P
        lens = counts.reshape(-1)[valid_g]
        tok = tok_fts.float()
        pred = m.encode_patch_sets(tok, geom7, lens)                                                   # PRE-FF:960-964
        grp_of_tok = torch.from_numpy(np.repeat(np.arange(len(valid_g)), lens)).to(dev)
        # The targets are means of float16 CLIP features and the reference evaluates them IN float16 (`patch_fts[...].mean(0)` on a half
        # tensor, half - half, PRE-FF:969-972; the losses normalise them in half too, losses.alignment_loss): same rounding points here.
        seg_mean = (torch.zeros((len(valid_g), tok.shape[1]), device=dev).index_add_(0, grp_of_tok, tok) / torch.from_numpy(lens).to(dev).float()[:, None]).half()
        frame_mean = tok.view(B, P, -1).mean(1).half()                                                 # patch_fts.mean(0) of every environment
        env_of_grp = torch.from_numpy(valid_g // n_max).to(dev)
        fm = frame_mean.index_select(0, env_of_grp)
        self.pred_i.append(pred); self.tgt_i.append(seg_mean)                                          # PRE-FF:969-974
        self.pred_is.append(pred - fm.float()); self.tgt_is.append(seg_mean - fm)
        cen = centroid.index_select(0, torch.from_numpy(valid_g).to(dev))
        if image_ft_ix is not None:                                                                    # PRE-FF:992-1008: the frame's instances as ONE zone
            n_inst = np.bincount(valid_g // n_max, minlength=B)
            cmean = torch.zeros((B, 3), device=dev).index_add_(0, env_of_grp, cen) / torch.from_numpy(np.maximum(n_inst, 1)).to(dev).float()[:, None]
            geom4 = torch.cat([cen - cmean.index_select(0, env_of_grp), torch.sqrt((cen * cen).sum(-1, keepdim=True))], -1)
            zone = m.encode_zone_sets(pred, geom4, n_inst)
            self.pred_z.append(zone); self.tgt_z.append(image_ft_ix.float())
            self.pred_zs.append(zone - image_ft_mean.float()); self.tgt_zs.append(image_ft_ix.float() - image_ft_mean.float())
        # ground-truth instance id of every segment: nearest GT point of each patch (k = 1), majority vote (PRE-FF:977-983)
        self.cur = dict(pred=pred, cen=cen, valid_g=valid_g, n_max=n_max, gt=None, inv=None)
        inv = np.full(B * n_max, -1, np.int64)
        inv[valid_g] = np.arange(len(valid_g))
        self.cur["inv"] = torch.from_numpy(inv).to(dev)
        if self.gt_xyz is not None:
            self.cur["gt"] = self._label_segments(ops, pools, envs, slots_h, row_base, order, grp_of_tok, len(valid_g))
        self.debug.append(dict(tok_fts=tok.detach(), geom7=geom7.detach(), lens=lens.copy(), cen=cen.detach(), env_of_group=(valid_g // n_max).copy(), B=B, P=P,
                               img_ix=None if image_ft_ix is None else image_ft_ix.detach().float(), img_mean=None if image_ft_mean is None else image_ft_mean.detach().float(),
                               gt=None if self.cur["gt"] is None else self.cur["gt"].clone(), pairs=None))
        return pred.detach()

    def _gt_device(self, dev):
        if self._gt_dev is None:
            cap = max(len(x) for x in self.gt_xyz)
            pts = torch.zeros((len(self.gt_xyz), cap, 3), dtype=torch.float32, device=dev)
            for e, x in enumerate(self.gt_xyz):
                pts[e, :len(x)] = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)
            self._gt_dev = (pts, cap, torch.tensor([len(x) for x in self.gt_xyz], dtype=torch.int32, device=dev))
        return self._gt_dev

    def _gt_labels_device(self, dev):
        if self._gt_lab_dev is None:
            cap = max(len(x) for x in self.gt_label)
            lo = min(int(np.min(x)) for x in self.gt_label)
            hi = max(int(np.max(x)) for x in self.gt_label)
            lab = torch.full((len(self.gt_label), cap), 0, dtype=torch.int64, device=dev)
            for e, x in enumerate(self.gt_label):
                lab[e, :len(x)] = torch.from_numpy(np.ascontiguousarray(x, np.int64)).to(dev) - lo
            self._gt_lab_dev = (lab, lo, hi - lo + 1)
        return self._gt_lab_dev

    def _label_segments(self, ops, pools, envs, slots_h, row_base, order, grp_of_tok, n_groups):
        """GT instance id of every 2D segment (PRE-FF:977-983): k = 1 nearest GT point of each patch (`d3d_knn[_chunked]` over the GT cloud),
        then the majority label of the segment's patches, ties to the SMALLEST label (`unique_vals[counts.argmax()]`: torch.unique is
        sorted, argmax takes the first maximum).  Everything stays on the device (round 5; rounds 3-4 read the indices back and voted
        with numpy: a synchronisation + 128 Python iterations per view): one sorted `unique` over (group, label) keys, the per-group
        maximum count, and the smallest key among the entries that reach it.  -> (n_groups,) int64 on the device."""
        dev = pools.rows_pos.device
        B, P = order.shape
        pts, cap, n_pts = self._gt_device(dev)
        lab, lo, L = self._gt_labels_device(dev)
        rows = torch.from_numpy((np.asarray(row_base, np.int64)[:, None] + order).astype(np.int64)).to(dev)       # (B, P) rows in (label, patch) order
        slots_d = torch.as_tensor(np.asarray(slots_h, np.int64)).to(dev)
        q = pools.rows_pos[slots_d[:, None], rows].contiguous()                                                     # (B, P, 3) world positions
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
        all_envs = len(envs) == pts.shape[0] and list(envs) == list(range(len(envs)))
        env_d = None if all_envs else torch.tensor(list(envs), device=dev)
        sel = pts if all_envs else pts.index_select(0, env_d)
        npts = n_pts if all_envs else n_pts.index_select(0, env_d)
        _d2, idx = ops.knn(sel.contiguous(), cap * 3, npts.contiguous(), q, P * 3, i32([P] * B), i32([1] * B), B, P, 1)
        lab_sel = lab if all_envs else lab.index_select(0, env_d)
        plab = torch.gather(lab_sel, 1, idx.view(B, P).long()).reshape(-1)                                          # label (minus lo) of every patch
        key = grp_of_tok.long() * L + plab
        uk, cnt = torch.unique(key, return_counts=True)                                                             # sorted: groups ascending, labels ascending
        g = uk // L
        best = torch.zeros(n_groups, dtype=cnt.dtype, device=dev).scatter_reduce_(0, g, cnt, "amax")
        big = torch.iinfo(torch.int64).max
        first = torch.full((n_groups,), big, dtype=torch.int64, device=dev).scatter_reduce_(0, g, torch.where(cnt == best[g], uk, torch.full_like(uk, big)), "amin")
        return first % L + lo

    # ---- hook 2: merge proposals ------------------------------------------------------------------------------------------------------------
    @torch.enable_grad()
    def merge(self, pools, slot_of_pair, pe, ps_, pj, pair_inst, n_envs):
        """(environment pe, segment ps_, proposal pj) pairs with instance row pair_inst.  Returns (pairs, 2) logits that make the memory merge
        by GROUND TRUTH when GT ids exist (PRE-FF:1031-1035); in training mode WITHOUT a GT point cloud the reference sets `merge_target`
        to zeros (PRE-FF:1061-1062): nothing merges, every segment becomes a new instance (the argmax is the `is_training=False`
        branch, PRE-FF:1064) -- the returned logits force exactly that."""
        m, c = self.model, self.cur
        dev = c["pred"].device
        g = c["inv"].index_select(0, (pe * c["n_max"] + ps_).long())
        f3 = pools.inst_fts[slot_of_pair.long(), pair_inst.long()]
        p3 = pools.inst_pos[slot_of_pair.long(), pair_inst.long()]
        x = torch.cat([f3, c["pred"].index_select(0, g), c["cen"].index_select(0, g) - p3], -1)       # [ft_3d, ft_2d, position offset] (PRE-FF:1023-1026)
        logits = m.merge_logits(x)
        if c["gt"] is None:
            no = torch.zeros((logits.shape[0],), device=logits.device)
            return torch.stack([1.0 - no, no], -1)                                                     # argmax = "do not merge" for every proposal
        gt3 = self._gt_rows(pools)[slot_of_pair.long(), pair_inst.long()]                              # (PRE-FF:1031: global_gt_instance_ids[gt_inds])
        target = (gt3 == c["gt"].index_select(0, g)).long()
        for j in range(n_envs):                                                                        # one cross-entropy per (environment, view)
            sel = pe == j
            if bool(sel.any()):
                t = LS.segmentation_loss(logits[sel], target[sel])
                if t is not None:
                    self.segm_terms.append(t)
        self.debug[-1]["pairs"] = dict(f3=f3.detach(), p3=p3.detach(), g=g.clone(), target=target.clone(), pe=pe.clone())
        return torch.stack([1.0 - target.float(), target.float()], -1)                                 # argmax = the ground-truth decision

    def _gt_rows(self, pools) -> torch.Tensor:
        """(slots, m_cap) GT ids next to the instance pool; grown with it."""
        S, cap = pools.inst_pos.shape[0], pools.inst_pos.shape[1]
        t = self.gt_rows
        if t is None or t.shape[0] < S or t.shape[1] < cap or t.device != pools.inst_pos.device:
            n = torch.full((S, cap), -1, dtype=torch.int64, device=pools.inst_pos.device)
            if t is not None:
                n[: min(S, t.shape[0]), : min(cap, t.shape[1])] = t[: min(S, t.shape[0]), : min(cap, t.shape[1])].to(n.device)
            self.gt_rows = t = n
        # A row whose instance died in delete_old_features_from_camera_frustum loses its GT id, like the reference's
        # `global_gt_instance_ids[b][instance_id] = -10000` (PRE-FF:728): dead rows are exactly the tomb-stoned ones (position -10000, both
        # planners), so the reset is taken from the pool -- otherwise KNN can propose a dead row (fewer than K live instances) whose STALE id
        # equals the segment's, and the memory would merge into a dead instance where the reference does not merge.  A recycled row gets its
        # new position and its new id (`new_instances`) before it is read again.
        t.masked_fill_(pools.inst_pos[: t.shape[0], : t.shape[1], 0] == -10000.0, -10000)
        return t

    def new_instances(self, pools, new_slots, new_rows, new_src):
        """Instance rows created by this view: remember their GT id (PRE-FF:1091-1097).  One scatter on the device."""
        if self.cur["gt"] is None or not len(new_rows):
            return
        dev = self.cur["gt"].device
        idx = torch.tensor([list(new_slots), list(new_rows), list(new_src)], dtype=torch.int64, device=dev)
        self._gt_rows(pools)[idx[0], idx[1]] = self.cur["gt"].index_select(0, self.cur["inv"].index_select(0, idx[2]))

    # ---- PRE-FF:1302-1345 ---------------------------------------------------------------------------------------------------------------------
    @torch.enable_grad()
    def losses(self):
        cat = lambda xs: torch.cat(xs, 0)
        zone = bool(self.pred_z)
        sim = LS.alignment_loss(cat(self.pred_i), cat(self.tgt_i), cat(self.pred_is), cat(self.tgt_is),
                                cat(self.pred_z) if zone else None, cat(self.tgt_z) if zone else None,
                                cat(self.pred_zs) if zone else None, cat(self.tgt_zs) if zone else None)
        # every trainable module takes part in every step (PRE-FF:1340 "Avoid DDP bug"): keeps the ranks' gradient sets identical
        dummy = self.model.merge_logits(torch.zeros((1, 2 * 768 + 3), device=sim.device)).sum() * 0.0
        segm = torch.stack(self.segm_terms).mean() if self.segm_terms else None
        return sim + dummy, segm


def render_target_from_grid(grid: torch.Tensor, hw: int = 24) -> torch.Tensor:
    """CLIP patch features of the novel view's own image (B, 576, 768) -> the (B, 144, 768) targets of the 12 x 12 ray grid: 2 x 2 average
    pooling over the 24 x 24 patch grid (PRE-TR:884-888)."""
    B, _, D = grid.shape
    g = grid.float().view(B, hw, hw, D).permute(0, 3, 1, 2)
    return F.avg_pool2d(g, kernel_size=2, stride=2).permute(0, 2, 3, 1).reshape(B, -1, D)


def pretrain_step(ff, trainer: FFTrainer, optimizer, update_kwargs: dict, clip_value: float = 10.0, render: Optional[dict] = None,
                  timing: Optional[dict] = None) -> dict:
    """One optimisation step (PRE-TR:479-526) on `ff` (a `Feature_Fields(variant="pretrain")`): forward with loss collection, NaN vote over
    the ranks, backward, gradient all-reduce (average), NaN scrub, value clipping, optimizer step; the updated weights are copied into the
    inference-path modules of `ff`.  `render` = dict(model=train_render.TrainableRenderer, views=[(positions, headings, target (B, 144, 768)),
    ...]): novel views rendered differentiably from the memory this step just updated and aligned with the CLIP patch features of their own
    images (PRE-TR:880-892, 1056-1075); the renderer's parameters must be in `optimizer` too.
    `timing`: a dict that receives synchronised wall milliseconds per phase (a measurement aid: it serialises host and device).
    Returns {'loss', 'sim_loss', 'segm_loss', 'render_loss', 'skipped', 'collectives'}."""
    import time as _time

    def lap(name, _t=[None]):
        if timing is None:
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        now = _time.perf_counter()
        if _t[0] is not None and name:
            timing[name] = timing.get(name, 0.0) + (now - _t[0]) * 1e3
        _t[0] = now

    lap("")
    optimizer.zero_grad(set_to_none=True)
    trainer.begin()
    ff.update_feature_fields(is_training=True, trainer=trainer, **update_kwargs)
    lap("update_forward")
    sim, segm = trainer.losses()
    lap("losses")
    loss = sim if segm is None else sim + segm
    rl = None
    if render is not None:
        preds, tgts = [], []
        for positions, headings, target in render["views"]:
            preds.append(render["model"].render(ff, positions, headings))
            tgts.append(torch.as_tensor(target).to(preds[-1].device, torch.float32))
        rl = LS.render_loss(torch.cat(preds, 0), torch.cat(tgts, 0))
        loss = loss + rl
    out = dict(loss=float(loss.detach()), sim_loss=float(sim.detach()), segm_loss=None if segm is None else float(segm.detach()),
               render_loss=None if rl is None else float(rl.detach()), skipped=False, collectives=0)
    if DD.any_nan_vote(loss.detach()):                                 # PRE-TR:503-509: a NaN on any rank skips the step on every rank
        out["skipped"] = True
        return out
    loss.backward()
    lap("backward")
    params = list(trainer.model.parameters()) + (list(render["model"].parameters()) if render is not None else [])
    zero = lambda p: torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()
    trainer.local_grads = {k: zero(p) for k, p in zip(trainer.model.names, trainer.model.plist)}      # this rank's own, raw (kept for the tests)
    # The reference runs backward under DDP (PRE-TR:356-360, 512): gradients are ALREADY averaged over the ranks when it scrubs NaNs
    # (PRE-TR:513-515) and clips (PRE-TR:517).  Same order here: average -> scrub -> clip.  A NaN on one rank therefore zeroes that element
    # on EVERY rank (NaN survives the sum), and clip(mean(g_r)), not mean(clip(g_r)), reaches the optimizer.
    out["collectives"] = DD.all_reduce_gradients(params, average=True, nan_to_zero=True)
    torch.nn.utils.clip_grad_value_(params, clip_value)                # PRE-TR:517
    trainer.last_grads = {k: zero(p) for k, p in zip(trainer.model.names, trainer.model.plist)}
    lap("allreduce_scrub_clip")
    ff.check_numerics()                                                # a non-finite float32 GEMM of THIS step's update is reported before its
    optimizer.step()                                                   # gradients reach the weights (one small device-to-host read)
    lap("optimizer")
    sync_weights(ff, trainer.model)
    lap("sync_weights")
    if render is not None:
        sync_render_weights(ff, render["model"])
    return out


def sync_render_weights(ff, model):
    """The inference renderer (`Feature_Fields.render_view_3d_patch`) reads the feature field's own `nerf_*` / `*_to_nerf_*` parameters:
    copy the trained tensors into them (in place) and let the renderer be rebuilt from them on its next use."""
    own = dict(ff.named_parameters())
    with torch.no_grad():
        for k, v in model.layer_weights().items():
            if k in own:
                own[k].copy_(v.to(own[k].device, own[k].dtype))
    ff._renderer = None


def sync_weights(ff, model: TrainableFF):
    """The no-grad memory path (`FFDense`: merged-instance re-encoding, zone update, inference) reads its own float32 copies."""
    for k, v in model.named_state().items():
        if k in ff.dense.w:
            ff.dense.w[k].copy_(v)
