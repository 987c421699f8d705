"""CPU restatement of the VLN `Feature_Fields` patch -> instance -> zone memory
(/root/reference/Dynam3D_VLN/vlnce_baselines/models/feature_fields.py, "VLN-FF").

TEST INFRASTRUCTURE (see oracle/geometry.py header).  Written as a plain sequential state
machine over python dicts + numpy stores so that it is easy to audit against the reference;
it is the CHECKER for the device-resident HIP implementation in dynam3d_amd/feature_fields.py.

Pinned: tests/test_oracle_vs_reference.py (container only) drives the reference's own class
and this one on the same seeded episodes; tests/golden/g4_*.npz are trajectories produced by
the reference and are replayed against this oracle on every CPU test run.

Reference quirks reproduced on purpose (compat='reference'):
  * F11  patch ids are the lowest unused dict keys, but rows are append-only and ids index rows
         (VLN-FF:433-445, 562-570, 662).
  * Z1   a NEW zone's position/feature are APPENDED even when its id is a recycled low id, so
         zone id != row after a zone deletion (VLN-FF:709-730).
  * Z2   an updated zone's position is the mean of member CELL CENTRES (VLN-FF:739-741).
  * Z3   a touched cell with no member instance yields a NaN zone position (mean of empty).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from . import geometry as G
from . import nnref as NN

F32 = np.float32


def lowest_unused(keys, n: int) -> np.ndarray:
    """VLN-FF:433-475: first n non-negative integers that are not keys (arange when empty)."""
    if len(keys) == 0:
        return np.arange(n, dtype=np.int64)
    out, i, used = [], 0, set(keys)
    limit = len(keys) + n
    while len(out) < n and i < limit:
        if i not in used:
            out.append(i)
        i += 1
    return np.array(out, dtype=np.int64)


class _Env:
    def __init__(self):
        self.pos = np.zeros((0, 3), F32)
        self.fts = np.zeros((0, 768), np.float16)
        self.dir = np.zeros((0,), F32)
        self.scale = np.zeros((0,), F32)
        self.owner: Dict[int, int] = {}
        self.members: Dict[int, np.ndarray] = {}
        self.ipos = np.zeros((0, 3), F32)
        self.ifts = np.zeros((0, 768), F32)
        self.zkey: Dict[tuple, int] = {}
        self.zmembers: Dict[int, np.ndarray] = {}
        self.zpos = np.zeros((0, 3), F32)
        self.zfts = np.zeros((0, 768), F32)
        self.tree = None  # snapshot of ipos (torch_kdtree copies on build)
        self.row_gt = np.zeros((0,), np.int64)   # training: GT instance id of every instance row (PRE-FF `global_gt_instance_ids`)


class FeatureFieldsOracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], batch_size: int = 1, hfov=90.0, vfov=90.0,
                 H=24, W=24, cell=(2.0, 2.0, 2.0), frustum_far=3.0, num_proposals=2):
        self.sd = {k: v.float() for k, v in state_dict.items()}
        self.H, self.W, self.hfov, self.vfov = H, W, hfov, vfov
        self.cell, self.far, self.K = cell, frustum_far, num_proposals
        self.reset(batch_size)

    # ---- lifecycle (VLN-FF:186-240) ---------------------------------------------------------
    def reset(self, batch_size: int):
        self.batch_size = batch_size
        self.env: List[_Env] = [_Env() for _ in range(batch_size)]

    def pop(self, index: int):
        self.batch_size -= 1
        self.env.pop(index)

    def initialize_camera_setting(self, hfov, vfov):
        self.hfov, self.vfov = hfov, vfov

    # ---- NN helpers ---------------------------------------------------------------------------
    @torch.no_grad()
    def _encode_patches(self, pos, direction, scale, fts16, centroid):
        geom = G.segment_geometry(pos, direction, scale, centroid)
        emb = torch.from_numpy(fts16.astype(np.float32)) + NN.mlp_ln_gelu(torch.from_numpy(geom), self.sd, "patch_to_instance_position_embedding")
        out = NN.encode_set(emb, self.sd["aggregate_patch_to_instance_embedding"], self.sd, "aggregate_patch_to_instance_encoder")
        return out.numpy()[0]

    @torch.no_grad()
    def _encode_zone(self, rel, dist, ifts):
        geom = np.concatenate([rel, dist[:, None]], axis=1).astype(F32).reshape(-1, 4)
        emb = torch.from_numpy(ifts.reshape(-1, 768)) + NN.mlp_ln_gelu(torch.from_numpy(geom), self.sd, "instance_to_zone_position_embedding")
        out = NN.encode_set(emb, self.sd["aggregate_instance_to_zone_embedding"], self.sd, "aggregate_instance_to_zone_encoder")
        return out.numpy()[0]

    @staticmethod
    def _norm3(p):
        p = p.astype(F32).reshape(-1, 3)
        with np.errstate(over="ignore", invalid="ignore"):
            n2 = (((p[:, 0] * p[:, 0]).astype(F32) + (p[:, 1] * p[:, 1]).astype(F32)).astype(F32) + (p[:, 2] * p[:, 2]).astype(F32)).astype(F32)
            return np.sqrt(n2).astype(F32)

    @staticmethod
    def _mean(x):
        if x.shape[0] == 0:
            return np.full(x.shape[1:], np.nan, F32)
        return G.mean_rows_f64(x)

    # ---- a4 + cascade: delete_old_features_from_camera_frustum (VLN-FF:329-396) ---------------
    def delete_old_features_from_camera_frustum(self, batch_depth, batch_position=None, batch_heading=None, num_of_views=1, view_ids=None,
                                                batch_camera_intrinsic=None, batch_extrinsic=None):
        """batch_depth (B,V,Hd,Wd) metres (already preprocess_depth'ed).  `view_ids` = the Pretrain variant's signature
        (PRE-FF:674-696): view ix looks along heading - view_ids[ix]*pi/6 in the cull as well."""
        batch_depth = np.asarray(batch_depth, F32)
        if view_ids is not None:
            num_of_views = len(view_ids)
        if batch_extrinsic is not None:                                  # PRE-FF:680-681: all views of the env
            num_of_views = batch_depth.shape[1]
        for b, e in enumerate(self.env):
            for ix in range(num_of_views):
                if e.pos.shape[0] == 0:
                    continue
                # NB: the VLN variant does not add the per-view heading offset here (VLN-FF:347); the Pretrain one does (PRE-FF:696)
                if batch_extrinsic is not None:                         # get_frustum_mask (PRE-FF:693)
                    mask = G.frustum_mask_pinhole(e.pos, batch_depth[b, ix], np.asarray(batch_camera_intrinsic[b][ix]),
                                                  np.asarray(batch_extrinsic[b][ix]), 0.0, self.far, 0.1)
                else:
                    off = 0.0 if view_ids is None else int(view_ids[ix]) * (-math.pi / 6)
                    mask = G.frustum_mask_habitat(e.pos, batch_depth[b, ix], batch_position[b], off + batch_heading[b],
                                                  self.hfov, self.vfov, 0.0, self.far, 0.1)
                e.pos[mask] = G.TOMBSTONE
                e.fts[mask] = 0
                e.dir[mask] = 0
                e.scale[mask] = 0
                for pid in np.nonzero(mask)[0].tolist():
                    if pid not in e.owner:
                        continue
                    inst = e.owner.pop(pid)
                    e.members[inst] = e.members[inst][e.members[inst] != pid]
                    if len(e.members[inst]) == 0:
                        e.members.pop(inst)
                        key = tuple(G.zone_cell_centre(e.ipos[inst:inst + 1], self.cell)[0].tolist())
                        e.ipos[inst] = G.TOMBSTONE
                        e.ifts[inst] = 0
                        if inst < e.row_gt.shape[0]:
                            e.row_gt[inst] = -10000                      # PRE-FF:728
                        if key in e.zkey:
                            zid = e.zkey[key]
                            e.zmembers[zid] = e.zmembers[zid][e.zmembers[zid] != inst]
                            if len(e.zmembers[zid]) == 0:
                                e.zkey.pop(key)
                                e.zmembers.pop(zid)
                                e.zpos[zid] = G.TOMBSTONE
                                e.zfts[zid] = 0
            e.tree = e.ipos.copy() if e.ipos.shape[0] > 0 else None

    # ---- update_feature_fields (VLN-FF:493-815) -----------------------------------------------
    @torch.no_grad()
    def update_feature_fields(self, batch_depth24, batch_grid_ft, patch_segm, batch_position=None, batch_heading=None, num_of_views=1,
                              view_ids=None, batch_camera_intrinsic=None, batch_rot=None, batch_trans=None, depth_scale=1000.0,
                              depth_trunc=1000.0, view_hw=(12, 12), train=None):
        """train = dict(gt_xyz=[(Ng,3) f32 per env], gt_label=[(Ng,) int per env], image_ft=(B,V,768) f32 or None): the Pretrain class's
        `is_training=True` branch (PRE-FF:843-1345) FROM RAW INPUTS -- every segment is labelled with the majority GT id of its patches'
        nearest GT points (PRE-FF:976-983), the memory merges by ground truth (PRE-FF:1029-1035), and the inputs of every loss term are
        recorded in `self.train_views` (one record per (environment, view), the format of oracle/train_ref.training_loss_and_grads)
        together with the integer by-products the reference returns (`self.train_ints`).  The module arithmetic is the inference one
        (dropout off: the goldens' p = 0 convention, oracle/ref_harness.RefTrainingRun).

        batch_depth24 (B,V,P) metres; batch_grid_ft (B,V,P,768); patch_segm (B,V,H,W) or (B*V,1,H,W)
        dense labels; positions habitat xyz; headings rad.  `view_ids` (Pretrain signature, PRE-FF:843,920): view ix looks
        along heading - view_ids[ix]*pi/6 instead of heading - ix*pi/6 (VLN-FF:550); with is_training=False and no GT point
        cloud the Pretrain update is otherwise the same state machine (diffed, SURVEY.md 8 note under a23)."""
        P = self.H * self.W
        if view_ids is not None:
            num_of_views = len(view_ids)
        if batch_camera_intrinsic is not None:
            # intrinsics mode (PRE-FF:849-856, 886-916): `batch_depth24` holds the RAW depth stacks (B,V,Hd,Wd)
            num_of_views = np.asarray(batch_depth24[0]).shape[0]
            scale_tan = G.view_scale_tan(np.asarray(batch_camera_intrinsic[0][0]), np.asarray(batch_depth24[0]).shape[-2:], view_hw)
        vid = list(range(num_of_views)) if view_ids is None else [int(v) for v in view_ids]
        segm_all = np.asarray(patch_segm).reshape(self.batch_size, num_of_views, P)
        self.last_debug = []
        self.train_views = []
        self.train_ints = dict(gt_nn=[[] for _ in self.env], gt3d=[[] for _ in self.env], gt_in_zone=[[] for _ in self.env])
        for b, e in enumerate(self.env):
            for ix in range(num_of_views):
                dbg = {}
                proposal_num = min(len(e.members), self.K)
                if batch_camera_intrinsic is not None:
                    pos, direction, scale = G.unproject_pinhole(np.asarray(batch_depth24[b][ix]), np.asarray(batch_camera_intrinsic[b][ix]),
                                                                np.asarray(batch_rot[b][ix]), np.asarray(batch_trans[b][ix]), scale_tan,
                                                                self.W, depth_scale, depth_trunc, (self.H, self.W))
                else:
                    pos, direction, scale = G.unproject_habitat(np.asarray(batch_depth24[b][ix], F32), batch_position[b],
                                                                vid[ix] * (-math.pi / 6) + batch_heading[b], self.H, self.W, self.hfov, self.vfov)
                fts16 = np.asarray(batch_grid_ft[b][ix]).astype(np.float16)
                e.pos = np.concatenate([e.pos, pos], 0)
                e.dir = np.concatenate([e.dir, direction], 0)
                e.scale = np.concatenate([e.scale, scale], 0)
                e.fts = np.concatenate([e.fts, fts16], 0)
                segm = segm_all[b, ix]
                labels = np.unique(segm).tolist()
                n = len(labels)
                new_pos = np.zeros((n, 3), F32)
                new_fts = np.zeros((n, 768), F32)
                for i, s in enumerate(labels):
                    sel = segm == s
                    new_pos[i] = G.mean_rows_f64(pos[sel])
                    new_fts[i] = self._encode_patches(pos[sel], direction[sel], scale[sel], fts16[sel], new_pos[i])
                dbg["new_pos"], dbg["new_fts"] = new_pos.copy(), new_fts.copy()
                gt_seg = rec = None
                if train is not None:
                    gxyz, glab = np.asarray(train["gt_xyz"][b], F32), np.asarray(train["gt_label"][b], np.int64)
                    gt_seg = np.zeros((n,), np.int64)
                    toks, geoms, lens = [], [], []
                    for i, s in enumerate(labels):
                        sel = segm == s
                        _, nn = G.knn_bruteforce(gxyz, pos[sel], 1)                    # k = 1 nearest GT point of every patch (PRE-FF:978)
                        self.train_ints["gt_nn"][b].append(nn[:, 0].astype(np.int64))
                        vals, cnt = np.unique(glab[nn[:, 0]], return_counts=True)
                        gt_seg[i] = vals[cnt.argmax()]                                 # unique_vals[counts.argmax()] (PRE-FF:982-983)
                        toks.append(fts16[sel]); lens.append(int(sel.sum()))
                        geoms.append(G.segment_geometry(pos[sel], direction[sel], scale[sel], new_pos[i]))
                    img = train.get("image_ft")
                    rec = dict(tok_fts=np.concatenate(toks, 0), geom7=np.concatenate(geoms, 0).astype(F32), lens=np.asarray(lens, np.int64),
                               cen=new_pos.copy(), env_of_group=np.zeros((n,), np.int64), B=1, P=P, frame_fts=fts16.copy(),
                               img_ix=None if img is None else np.asarray(img[b][ix:ix + 1], F32),
                               img_mean=None if img is None else np.asarray(img[b], F32).mean(0, keepdims=True), gt=gt_seg.copy(), pairs=None)
                    self.train_views.append(rec)

                if e.tree is not None:
                    d2, idx = G.knn_bruteforce(e.tree, new_pos, proposal_num)
                    if float(d2.astype(np.float64).sum()) > 1e6:          # VLN-FF:607-610
                        col = d2.sum(0)
                        proposal_num = int((col < 1e6).sum())
                        d2, idx = G.knn_bruteforce(e.tree, new_pos, proposal_num)
                    dbg["knn_idx"], dbg["knn_d2"] = idx.copy(), d2.copy()
                    if proposal_num > 0:
                        delta = (new_pos[:, None, :] - e.ipos[idx]).astype(F32)
                        x = np.concatenate([e.ifts[idx], np.repeat(new_fts[:, None, :], proposal_num, 1), delta], -1)
                        logits = NN.mlp_ln_gelu(torch.from_numpy(x), self.sd, "instance_merge_discriminator").numpy()
                        target = np.argmax(logits, -1)                  # == argmax(softmax)
                        if train is not None:                           # merge by GROUND TRUTH (PRE-FF:1031-1035)
                            target = (e.row_gt[idx] == gt_seg[:, None]).astype(np.int64)
                            rec["pairs"] = dict(f3=e.ifts[idx].reshape(-1, 768).copy(), p3=e.ipos[idx].reshape(-1, 3).copy(),
                                                g=np.repeat(np.arange(n), proposal_num), target=target.reshape(-1).copy(),
                                                pe=np.zeros((n * proposal_num,), np.int64))
                    else:
                        logits = np.zeros((n, 0, 2), F32)
                        target = np.zeros((n, 0), np.int64)
                    dbg["merge_logits"], dbg["merge_target"] = logits, target.copy()
                    is_new = target.sum(-1) == 0
                    new_inst_ids = lowest_unused(e.members.keys(), int(is_new.sum())) if is_new.any() else None
                    new_patch_ids = lowest_unused(e.owner.keys(), P)
                    dbg["new_patch_ids"] = new_patch_ids.copy()
                    nxt = 0
                    for s in range(n):
                        pids = new_patch_ids[segm == s]
                        if is_new[s]:
                            inst = int(new_inst_ids[nxt]); nxt += 1
                            e.members[inst] = pids
                            for p_ in pids.tolist():
                                e.owner[p_] = inst
                            if inst < e.ipos.shape[0]:
                                e.ipos[inst], e.ifts[inst] = new_pos[s], new_fts[s]
                                if train is not None:
                                    e.row_gt[inst] = gt_seg[s]
                            else:
                                e.ipos = np.concatenate([e.ipos, new_pos[s:s + 1]], 0)
                                e.ifts = np.concatenate([e.ifts, new_fts[s:s + 1]], 0)
                                if train is not None:
                                    e.row_gt = np.concatenate([e.row_gt, gt_seg[s:s + 1]], 0)
                            if train is not None:
                                self.train_ints["gt3d"][b].append(int(gt_seg[s]))
                        else:
                            j = int(np.nonzero(target[s])[0][0])       # first positive proposal only
                            inst = int(idx[s, j])
                            e.members[inst] = np.concatenate([e.members[inst], pids], 0)
                            for p_ in pids.tolist():
                                e.owner[p_] = inst
                            rows = e.members[inst]                     # ids used as ROW indices (F11)
                            cen = G.mean_rows_f64(e.pos[rows])
                            e.ipos[inst] = cen
                            e.ifts[inst] = self._encode_patches(e.pos[rows], e.dir[rows], e.scale[rows], e.fts[rows], cen)
                            if train is not None:
                                self.train_ints["gt3d"][b].append(int(e.row_gt[inst]))
                    # zones (VLN-FF:694-756)
                    gz = G.zone_cell_centre(e.ipos, self.cell)
                    uz = G.zone_cell_centre(new_pos, self.cell)
                    zones = np.unique(uz, axis=0)
                    zone_ids = lowest_unused(e.zmembers.keys(), len(zones))
                    nz = 0
                    for zi in range(len(zones)):
                        key = tuple(zones[zi].tolist())
                        mask = (gz[:, 0] == key[0]) & (gz[:, 1] == key[1]) & (gz[:, 2] == key[2])
                        if key not in e.zkey:
                            zid = int(zone_ids[nz]); nz += 1
                            e.zkey[key] = zid
                            e.zmembers[zid] = np.nonzero(mask)[0]
                            self.train_ints["gt_in_zone"][b].append(e.zmembers[zid].astype(np.int64))
                            pset = e.ipos[mask]
                            cen = self._mean(pset)
                            e.zpos = np.concatenate([e.zpos, cen[None]], 0)            # quirk Z1: append
                            ft = self._encode_zone((pset - cen[None]).astype(F32), self._norm3(pset), e.ifts[mask])
                            e.zfts = np.concatenate([e.zfts, ft[None]], 0)
                        else:
                            zid = e.zkey[key]
                            e.zmembers[zid] = np.nonzero(mask)[0]
                            self.train_ints["gt_in_zone"][b].append(e.zmembers[zid].astype(np.int64))
                            pset = gz[mask]                                            # quirk Z2: cell centres
                            cen = self._mean(pset)
                            e.zpos[zid] = cen
                            e.zfts[zid] = self._encode_zone((pset - cen[None]).astype(F32), self._norm3(pset), e.ifts[mask])
                else:
                    # first frame of the episode (VLN-FF:759-812)
                    e.ipos, e.ifts = new_pos.copy(), new_fts.copy()
                    if train is not None:
                        e.row_gt = gt_seg.copy()
                        self.train_ints["gt3d"][b].extend(int(g) for g in gt_seg)
                    inst_ids = lowest_unused(e.members.keys(), n)
                    new_patch_ids = lowest_unused(e.owner.keys(), P)
                    dbg["new_patch_ids"] = new_patch_ids.copy()
                    for s in labels:
                        pids = new_patch_ids[segm == s]
                        inst = int(inst_ids[s])
                        e.members[inst] = pids
                        for p_ in pids.tolist():
                            e.owner[p_] = inst
                    uz = G.zone_cell_centre(new_pos, self.cell)
                    zones = np.unique(uz, axis=0)
                    zone_ids = lowest_unused(e.zmembers.keys(), len(zones))
                    for zi in range(len(zones)):
                        key = tuple(zones[zi].tolist())
                        mask = (uz[:, 0] == key[0]) & (uz[:, 1] == key[1]) & (uz[:, 2] == key[2])
                        zid = int(zone_ids[zi])
                        e.zkey[key] = zid
                        e.zmembers[zid] = np.nonzero(mask)[0]
                        self.train_ints["gt_in_zone"][b].append(e.zmembers[zid].astype(np.int64))
                        pset = new_pos[mask]
                        cen = self._mean(pset)
                        e.zpos = np.concatenate([e.zpos, cen[None]], 0)
                        ft = self._encode_zone((pset - cen[None]).astype(F32), self._norm3(pset), new_fts[mask])
                        e.zfts = np.concatenate([e.zfts, ft[None]], 0)
                e.tree = e.ipos.copy() if e.ipos.shape[0] > 0 else None
                self.last_debug.append(dbg)

    # ---- a12 get_environment_features (VLN-FF:818-862) ------------------------------------------
    def get_environment_features(self, agent_position, agent_heading, instance_distance=5.0, zone_distance=100.0):
        out = {"batch_instance_fts": [], "batch_instance_relative_position": [], "batch_zone_fts": [],
               "batch_zone_relative_position": [], "batch_instance_ids": [], "batch_zone_ids": []}
        for b, e in enumerate(self.env):
            ids = np.array(list(e.members.keys()), np.int64)
            rel, keep = G.agent_frame(e.ipos[ids], agent_position[b], agent_heading[b], instance_distance)
            out["batch_instance_relative_position"].append(rel[keep])
            out["batch_instance_fts"].append(e.ifts[ids][keep])
            out["batch_instance_ids"].append(ids[keep])
            zids = np.array(list(e.zmembers.keys()), np.int64)
            rel, keep = G.agent_frame(e.zpos[zids], agent_position[b], agent_heading[b], zone_distance)
            out["batch_zone_relative_position"].append(rel[keep])
            out["batch_zone_fts"].append(e.zfts[zids][keep])
            out["batch_zone_ids"].append(zids[keep])
        return out

    def get_patch_3d_info(self, depth24):
        return G.patch_3d_info(np.asarray(depth24, F32), self.H, self.W, self.hfov, self.vfov)
