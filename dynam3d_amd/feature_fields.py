"""`Feature_Fields` -- MI355X-native drop-in for the reference's online patch -> instance -> zone
memory (Dynam3D_VLN/vlnce_baselines/models/feature_fields.py:119-862, "VLN-FF").

Same public surface (reset / pop / initialize_camera_setting / delete_old_features_from_camera_frustum
/ update_feature_fields / get_environment_features / get_patch_3d_info / delete_feature_fields,
attributes batch_size, args, keep_target_waypoint, history_actions) but a different architecture:

  * all patch / instance / zone stores are DEVICE-RESIDENT SoA pools (ops.Pools) sized for the
    episode; nothing is copied to the host except a few hundred bytes of indices per step;
  * every environment of the batch is processed by the same kernel launches (the reference loops
    over environments, views and segments in Python);
  * float work = HIP kernels (ops.HipOps -> libdynam3d_hip.so) + batched dense encoders
    (ff_dense.FFDense); the dict / id bookkeeping is the C++ state machine behind d3d_ff_*
    (csrc/ff_state.cpp), which only ever sees integers.

`compat='reference'` (default) reproduces the reference's id/row quirks so golden
trajectories match; `compat='fixed'` keeps id == row.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch

from .ops import pinhole_unproject_rows, pinhole_views, FTS, CameraTables, Pools, make_pose
from ._ffstate import FFState
from .ff_plan import REPORT_WORDS, DevicePlanner
from .ff_dense import FFDense
from .modules import RefreshOnChange, install_param, own_copy

RENDER_PREFIXES = ("nerf_", "patch_to_nerf_", "aggregate_patch_to_nerf_")          # Pretrain-only renderer parameters (PRE-FF:221-254)
IGNORED_PREFIXES = ("FastSAM", "freezed_", "clip_")                                 # sub-networks / frozen copies the memory update never reads

K_MAX_CHOICES = (1, 2, 4, 8)


def _args_namespace():
    # VLN-FF:22-46 defaults (fts_dim is declared float there; kept numeric-compatible)
    return SimpleNamespace(input_hfov=90.0, input_vfov=90.0, input_height=24, input_width=24, fts_dim=768,
                           zone_x_length=2.0, zone_y_length=2.0, zone_z_length=2.0, deleted_frustum_distance=3.0,
                           num_proposal_instances=2,
                           # novel-view rays of the Pretrain class (PRE-FF:47-74); used by the renderer and, in the intrinsics
                           # mode, for the patch scale (PRE-FF:849-856, 909)
                           near=0.0, far=10.0, view_height=12, view_width=12, N_samples=501)


def _host(x):
    """tensor / array / nested list -> numpy on the host (camera matrices are a few numbers per view)."""
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _ray0_tan(fx_view: float, view_width: int, near: float, far: float, n_samples: int) -> float:
    """|tan(rel_direction[0][-1])| of `get_rays(init_camera_intrinsic)` (PRE-FF:390-405, 909): ray 0 = pixel (0,0), last depth
    sample, principal point view_width/2; Open3D evaluates x = (0 - cx) * z / fx in double on the float32 depth value."""
    z = float(np.float32(near + (far - near) / n_samples * n_samples))
    x = (0.0 - view_width / 2) * z / fx_view
    return math.fabs(math.tan(-math.atan(x / z)))


class Feature_Fields(RefreshOnChange):
    """A `torch.nn.Module` like the reference's (VLN-FF:119): its parameters sit under the reference's own state-dict keys
    (`weights.ff_param_spec`, float32), so `parameters()`, `state_dict()`, `load_state_dict(torch.load("dynam3d.pth"), strict=True)`
    (VLN-POL:77-80), `to()`, `eval()` behave as the trainer expects; the kernels' view of them is `self.dense` (ff_dense.FFDense,
    aliasing the parameter storage), rebuilt by `refresh()` when the parameters were replaced."""

    def __init__(self, batch_size: int = 1, device="cuda", state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 compat: str = "reference", max_steps: int = 64, max_views: int = 1, m_cap: int = 4096, z_cap: int = 2048,
                 ops=None, segmenter=None, variant: str = "vln", seed: int = 0, planner: Optional[str] = None):
        """variant: "vln" = the VLN class's argument defaults (VLN-FF:22-46, 2 merge proposals); "pretrain" = the Pretrain
        class's (PRE-FF:29-45: `num_proposal_instances` 4) -- the memory update itself is the same state machine.
        `state_dict=None`: seeded synthetic parameters (the reference constructs with random initial weights and loads
        `dynam3d.pth` afterwards, VLN-POL:77-80).
        planner: "device" = the id / dict bookkeeping runs on the GPU next to the float kernels (csrc/ff_plan.h: one small report read
        back per view); "host" = the C++ state machine (csrc/ff_state.cpp: the update waits for the host before every decision; the
        training branch needs it).  Default: $D3D_FF_PLANNER, else "device" for the VLN variant (inference only, VLN-POL:150-151) and
        "host" for the Pretrain variant (whose update also runs with is_training=True)."""
        super().__init__()
        self.device = torch.device(device)
        self.args = _args_namespace()
        if variant not in ("vln", "pretrain"):
            raise ValueError("variant must be 'vln' or 'pretrain'")
        if variant == "pretrain":
            self.args.num_proposal_instances = 4
        if ops is None:
            from .ops import HipOps
            ops = HipOps()                      # raises if libdynam3d_hip.so is missing: no CPU fallback
        self.ops = ops
        self.compat = compat
        self.max_steps, self.max_views = max_steps, max_views
        self._m_cap, self._z_cap = m_cap, z_cap
        self.segmenter = segmenter              # callable(batch_image) -> (N,1,24,24) dense int labels (a6)
        self.dense: Optional[FFDense] = None
        self._renderer = None
        self._sig = None
        if state_dict is None:
            from .weights import ff_param_spec, synth_state_dict
            state_dict = synth_state_dict(ff_param_spec(int(self.args.fts_dim)), seed)
        for k, v in state_dict.items():
            if not k.startswith(IGNORED_PREFIXES):
                install_param(self, k, own_copy(v, self.device, torch.float32))
        self._init_refresh_hooks()
        self.refresh()
        self.planner = planner or os.environ.get("D3D_FF_PLANNER") or ("device" if variant == "vln" else "host")
        if self.planner not in ("host", "device"):
            raise ValueError("planner must be 'host' or 'device'")
        if self.planner == "device":
            self.state = DevicePlanner(compat, self.P, self.args.num_proposal_instances, self.device)
        else:
            self.state = FFState(ops.lib, compat, self.P, self.args.num_proposal_instances)
        self._cam: Optional[CameraTables] = None
        self.pools: Optional[Pools] = None
        self.reset(batch_size)

    # ------------------------------------------------------------------------------------------
    @property
    def P(self):
        return self.args.input_height * self.args.input_width

    @property
    def cell_len(self):
        return (float(self.args.zone_x_length), float(self.args.zone_y_length), float(self.args.zone_z_length))

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """`nn.Module.load_state_dict` over the reference's keys (`dynam3d.pth` after convert_ckpt.py).  Keys of sub-networks the
        memory update never reads (`FastSAM.*`, `freezed_*`, `clip_*`) are dropped; Pretrain-only renderer keys (`nerf_*`,
        `patch_to_nerf_*`, `aggregate_patch_to_nerf_*`) are registered on first sight and enable `render_view_3d_patch`."""
        sd = {k: v for k, v in state_dict.items() if not k.startswith(IGNORED_PREFIXES)}
        own = set(dict(self.named_parameters()))
        for k, v in sd.items():
            if k.startswith(RENDER_PREFIXES) and k not in own:
                install_param(self, k, own_copy(v, self.device, torch.float32))
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def refresh(self):
        """Re-derive the kernels' view of the parameters (called by the nn.Module hooks, modules.RefreshOnChange; a no-op while
        the parameters' storages are the ones the current view aliases)."""
        params = dict(self.named_parameters())
        sig = tuple((k, p.data_ptr(), p._version, p.dtype, p.device) for k, p in params.items())
        if sig == self._sig:
            return
        self._sig = sig
        if params:
            dev = next(iter(params.values())).device
            if dev != self.device:
                # `.to(other_device)` (the trainer's `policy.to(device)`, VLN-TR:183-186): the kernels take raw pointers, so everything
                # they read must move with the parameters -- pools, planner state, camera tables and the renderer are re-allocated on
                # the new device.  The 3D memory does not travel (the reference moves the module once, before the first episode).
                self.device = dev
                self._cam = None
                self.pools = None
                if getattr(self, "state", None) is not None:
                    if self.planner == "device":
                        self.state = DevicePlanner(self.compat, self.P, self.args.num_proposal_instances, self.device)
                    self.reset(self.batch_size)
        flat = {k: p.detach() for k, p in params.items()}
        self._render_sd = {k: v for k, v in flat.items() if k.startswith(RENDER_PREFIXES)}
        self._renderer = None
        self.dense = FFDense({k: v for k, v in flat.items() if k not in self._render_sd}, self.device, n_head=int(self.args.fts_dim) // 64)

    # ---- a20-a23: PRE-FF:494-625 ----------------------------------------------------------------------
    @torch.no_grad()
    def render_view_3d_patch(self, batch_position=None, batch_heading=None, batch_camera_intrinsic=None, batch_rot=None,
                             batch_trans=None, visualization=False, debug=False, **render_kw):
        """Novel-view 12x12 feature map rendered from the stored patches (Pretrain `Feature_Fields` surface).
        Returns (features (B,12,12,768), positions (B,12,12,3), gt_labels=[]) like the reference (habitat mode)."""
        pin = {}
        if batch_rot is not None:                                       # intrinsics mode (PRE-FF:505-515, 532-536)
            if getattr(self, "view_intrinsic", None) is None:
                raise RuntimeError("intrinsics mode renders with the rays of the last update_feature_fields(batch_camera_intrinsic=...)")
            pin = dict(batch_rot=[_host(r) for r in batch_rot], batch_trans=[_host(t) for t in batch_trans], view_intrinsic=self.view_intrinsic)
        if not getattr(self, "_render_sd", None):
            raise RuntimeError("no renderer weights loaded (nerf_encoder/nerf_decoder/... keys of the Pretrain checkpoint)")
        if self._renderer is None:
            from .render import FieldRenderer
            self._renderer = FieldRenderer(self._render_sd, self.device, **render_kw)
        st = self.state
        n_rows = [st.count(e, st.ROWS) for e in range(self.batch_size)]
        out = self._renderer.render(self.pools, self.slots, n_rows, batch_position, batch_heading, self.ops, debug=debug, **pin)
        return (out[0], out[1], []) + tuple(out[2:])

    # ---- lifecycle (VLN-FF:186-240) ------------------------------------------------------------
    def reset(self, batch_size: int = 1):
        self.batch_size = batch_size
        tomb = [int(math.floor(-10000.0 / L)) for L in self.cell_len]
        n_cap = self.P * self.max_views * self.max_steps
        if self.pools is None or self.pools.rows_pos.shape[0] < batch_size or self.pools.n_cap != n_cap:
            self.pools = Pools.allocate(batch_size, n_cap, self._m_cap, self._z_cap, self.device)
        if self.planner == "device":
            self.state.reset(batch_size, self.pools.n_cap, self.pools.m_cap, self.pools.z_cap, tomb)
        else:
            self.state.reset(batch_size)
            self.state.set_tomb_cell(tomb)
        self.slots: List[int] = list(range(batch_size))
        if self.planner == "device":
            self.state.env_slots = self.slots
        self.keep_target_waypoint = [None for _ in range(batch_size)]
        self.history_actions = [["none\n"] * 4 for _ in range(batch_size)]      # per-row lists
        self._tree_slots = [0] * batch_size

    def pop(self, index: int):
        self.batch_size -= 1
        self.state.pop(index)
        self.slots.pop(index)
        self.keep_target_waypoint.pop(index)
        self.history_actions.pop(index)
        self._tree_slots.pop(index)

    def initialize_camera_setting(self, hfov, vfov):
        self.args.input_hfov, self.args.input_vfov = hfov, vfov
        self._cam = None

    def delete_feature_fields(self):
        self.pools = None
        if self.planner == "device":
            self.state.reset(0, 0, 0, 0)
        else:
            self.state.reset(0)
        self.slots, self.keep_target_waypoint, self.history_actions, self._tree_slots = [], [], [], []
        self.batch_size = 0

    # ---- helpers ---------------------------------------------------------------------------------
    def _camera(self) -> CameraTables:
        if self._cam is None:
            self._cam = CameraTables.build(self.args.input_height, self.args.input_width, self.args.input_hfov,
                                           self.args.input_vfov, self.device)
        return self._cam

    def _i32(self, a) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(self.device, non_blocking=True)

    def _i32_many(self, *arrays):
        """Several int32 index arrays in ONE host-to-device copy (every small upload is ~15 us of host time on the update's latency
        chain): returns device views, each starting on a 16-byte boundary."""
        arrs = [np.ascontiguousarray(a, np.int32).reshape(-1) for a in arrays]
        offs, n = [], 0
        for a in arrs:
            offs.append(n)
            n += (a.size + 3) // 4 * 4
        buf = np.zeros(max(n, 4), np.int32)
        for a, o in zip(arrs, offs):
            buf[o:o + a.size] = a
        dev = torch.from_numpy(buf).to(self.device, non_blocking=True)
        return [dev[o:o + a.size] for a, o in zip(arrs, offs)]

    def _f32(self, a) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device, non_blocking=True)

    def _dev(self, x, dtype=torch.float32) -> torch.Tensor:
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        return x.to(self.device, dtype=dtype, non_blocking=True).contiguous()

    @staticmethod
    def _view_ids(view_ids):
        if view_ids is None:
            return None
        if isinstance(view_ids, torch.Tensor):
            view_ids = view_ids.cpu().numpy()
        return [int(v) for v in np.asarray(view_ids).reshape(-1)]

    def _stack_views(self, batch_depth) -> torch.Tensor:
        """list over envs of (V,H,W) images (or one (B,V,H,W) array) -> (B,V,H,W) float32 on the device; the batched kernels need
        the same number and size of views for every environment of the batch."""
        if isinstance(batch_depth, (list, tuple)):
            items = [torch.as_tensor(np.asarray(_host(d))) for d in batch_depth]
            if len({tuple(t.shape) for t in items}) != 1:
                raise ValueError("intrinsics mode: every environment of the batch must bring the same (V,H,W) depth stack")
            batch_depth = torch.stack(items)
        d = self._dev(batch_depth)
        if d.dim() == 5 and d.shape[-1] == 1:
            d = d[..., 0]
        if d.dim() != 4:
            raise ValueError("depth stack must be (B,V,H,W)")
        return d.contiguous()

    def _poses(self, positions, headings, envs, view_offset=0.0) -> torch.Tensor:
        return self._f32(np.stack([make_pose(positions[e], view_offset + float(headings[e])) for e in envs]))

    def _grow_rows(self, need: int):
        if need <= self.pools.n_cap:
            return
        new_cap = max(need, self.pools.n_cap * 2)
        old = self.pools
        S = old.rows_pos.shape[0]
        new = Pools.allocate(S, new_cap, old.m_cap, old.z_cap, self.device)
        n = old.n_cap
        new.rows_pos[:, :n], new.rows_fts[:, :n], new.rows_dir[:, :n], new.rows_scale[:, :n] = old.rows_pos, old.rows_fts, old.rows_dir, old.rows_scale
        new.inst_pos, new.inst_fts, new.tree_pos, new.zone_pos, new.zone_fts = old.inst_pos, old.inst_fts, old.tree_pos, old.zone_pos, old.zone_fts
        self.pools = new
        if self.planner == "device":
            self.state.ensure(R=new_cap)

    def _grow_slots(self, which: str, need: int):
        """The reference's instance / zone stores grow without bound (torch.cat, VLN-FF:645-648, 737); these pools double."""
        p = self.pools
        names = ("inst_pos", "inst_fts", "tree_pos") if which == "inst" else ("zone_pos", "zone_fts")
        cap = getattr(p, names[0]).shape[1]
        if need <= cap:
            return
        new_cap = max(need, 2 * cap)
        for n in names:
            old = getattr(p, n)
            new = torch.zeros((old.shape[0], new_cap) + tuple(old.shape[2:]), dtype=old.dtype, device=old.device)
            new[:, :cap] = old
            setattr(p, n, new)
        if self.planner == "device":
            self.state.ensure(**({"M": new_cap} if which == "inst" else {"Z": new_cap}))

    def _snapshot_tree(self):
        # kd-tree rebuild (VLN-FF:396, 815): the tree owns a COPY of the instance centres
        self.pools.tree_pos.copy_(self.pools.inst_pos)
        if self.planner == "device":                                   # (the device planner marks the rebuild itself: apply_hits / plan_zones)
            return
        for e in range(self.batch_size):
            self._tree_slots[e] = self.state.end_view(e)

    # ---- a4 + cascade (VLN-FF:329-396) ---------------------------------------------------------------
    @torch.no_grad()
    def delete_old_features_from_camera_frustum(self, batch_depth, batch_position=None, batch_heading=None,
                                                batch_camera_intrinsic=None, batch_extrinsic=None, num_of_views=1, view_ids=None):
        """`view_ids` is the Pretrain variant's keyword (PRE-FF:674): V = len(view_ids) and view ix is culled along
        heading - view_ids[ix]*pi/6 (PRE-FF:696); `num_of_views` is the VLN variant's (no per-view offset, VLN-FF:347)."""
        pinhole = batch_extrinsic is not None
        if pinhole:                                                     # PRE-FF:680-681: every view of the env, in order
            batch_depth = self._stack_views(batch_depth)
            num_of_views, view_ids = batch_depth.shape[1], None
        view_ids = self._view_ids(view_ids)
        if view_ids is not None:
            num_of_views = len(view_ids)
        depth = self._dev(batch_depth)                                  # (B,V,Hd,Wd) metres
        B, st, pools = self.batch_size, self.state, self.pools
        Hd, Wd = depth.shape[-2], depth.shape[-1]
        a = self.args
        intr = (float(np.float32(Wd / np.tan(np.deg2rad(a.input_hfov) / 2.0) / 2.0)),
                float(np.float32(Hd / np.tan(np.deg2rad(a.input_vfov) / 2.0) / 2.0)), Wd / 2.0, Hd / 2.0)
        if self.planner == "device":
            return self._delete_device(depth, batch_position, batch_heading, batch_camera_intrinsic, batch_extrinsic, num_of_views, view_ids, pinhole, intr)
        for ix in range(num_of_views):
            envs = [e for e in range(B) if st.count(e, st.ROWS) > 0]
            if not envs:
                continue
            n_rows = [st.count(e, st.ROWS) for e in envs]
            slot = self._i32([self.slots[e] for e in envs])
            hits = torch.empty((len(envs), max(n_rows)), dtype=torch.int32, device=self.device)
            n_hits = torch.zeros((len(envs),), dtype=torch.int32, device=self.device)
            d_ix = depth[envs, ix].contiguous() if len(envs) != B else depth[:, ix].contiguous()
            if pinhole:                                                 # get_frustum_mask (PRE-FF:98-118, 693)
                views = torch.from_numpy(pinhole_views([_host(batch_camera_intrinsic[e][ix]) for e in envs],
                                                       [_host(batch_extrinsic[e][ix]) for e in envs])).to(self.device)
                self.ops.frustum_cull_pinhole(pools, slot, self._i32(n_rows), max(n_rows), d_ix, views, 0.0,
                                              float(a.deleted_frustum_distance), 0.1, hits, n_hits)
            else:
                pose = self._poses(batch_position, batch_heading, envs,    # VLN: no per-view offset (VLN-FF:347)
                                   view_offset=0.0 if view_ids is None else view_ids[ix] * (-math.pi / 6))
                self.ops.frustum_cull(pools, slot, self._i32(n_rows), max(n_rows), d_ix, pose, intr, 0.0,
                                      float(a.deleted_frustum_distance), 0.1, hits, n_hits)
            n_hits_h = n_hits.cpu().numpy()                                  # sync #1: a few ints
            mx = int(n_hits_h.max())
            if mx == 0:
                continue
            hits_h = hits[:, :mx].cpu().numpy()
            di_s, di_r, dz_s, dz_r = [], [], [], []
            for j, e in enumerate(envs):
                dead_i, dead_z = st.apply_hits(e, hits_h[j, :n_hits_h[j]])
                di_s += [self.slots[e]] * len(dead_i); di_r += dead_i.tolist()
                dz_s += [self.slots[e]] * len(dead_z); dz_r += dead_z.tolist()
            if di_r:                                                          # VLN-FF:378-379
                s, r = self._i32(di_s), self._i32(di_r)
                self.ops.fill_rows(pools.inst_pos, s, r, -10000.0)
                self.ops.fill_rows(pools.inst_fts, s, r, 0.0)
            if dz_r:                                                          # VLN-FF:392-393
                s, r = self._i32(dz_s), self._i32(dz_r)
                self.ops.fill_rows(pools.zone_pos, s, r, -10000.0)
                self.ops.fill_rows(pools.zone_fts, s, r, 0.0)
        self._snapshot_tree()

    def _delete_device(self, depth, batch_position, batch_heading, batch_camera_intrinsic, batch_extrinsic, num_of_views, view_ids, pinhole, intr):
        """The same pass with the cascade on the device (d3d_ffdev_apply_hits consumes the frustum kernel's hit list in place): no
        device-to-host read."""
        B, pl, pools, a = self.batch_size, self.state, self.pools, self.args
        envs = list(range(B))
        n_rows = [pl.n_rows[e] for e in envs]
        mx = max(n_rows, default=0)
        if mx == 0:
            return
        slot, n_rows_d = self._i32_many([self.slots[e] for e in envs], n_rows)
        for ix in range(num_of_views):
            hits = torch.empty((B, mx), dtype=torch.int32, device=self.device)
            n_hits = torch.zeros((B,), dtype=torch.int32, device=self.device)
            d_ix = depth[:, ix].contiguous()
            if pinhole:
                views = torch.from_numpy(pinhole_views([_host(batch_camera_intrinsic[e][ix]) for e in envs],
                                                       [_host(batch_extrinsic[e][ix]) for e in envs])).to(self.device)
                self.ops.frustum_cull_pinhole(pools, slot, n_rows_d, mx, d_ix, views, 0.0, float(a.deleted_frustum_distance), 0.1, hits, n_hits)
            else:
                pose = self._poses(batch_position, batch_heading, envs, view_offset=0.0 if view_ids is None else view_ids[ix] * (-math.pi / 6))
                self.ops.frustum_cull(pools, slot, n_rows_d, mx, d_ix, pose, intr, 0.0, float(a.deleted_frustum_distance), 0.1, hits, n_hits)
            self.ops.ffdev_apply_hits(pl, slot, hits, n_hits, pools)         # VLN-FF:362-393 + the rebuild mark of VLN-FF:396
            pl.stale = True                                                   # LIVE / ZLIVE / OWNED changed on the device; no report is read here
        self._snapshot_tree()

    # ---- a6 contract ------------------------------------------------------------------------------------
    def get_patch_segm(self, batch_image, **kw):
        if self.segmenter is None:
            raise NotImplementedError("no segmenter configured: pass patch_segm=... (dense int labels, (B*V,1,24,24)); "
                                      "FastSAM itself is outside the hot path (SURVEY.md 8f-3)")
        return self.segmenter(batch_image, **kw)

    # ---- update_feature_fields (VLN-FF:493-815) ------------------------------------------------------------
    @torch.no_grad()
    def update_feature_fields(self, *args, **kw):
        """`_update_feature_fields` between two looks at the float32 GEMMs' DEVICE status word (f32_ops.F32Ops): the copy started at the end
        of the previous update is examined (it has landed long ago: no wait), a new one is started behind this update's launches -- a
        non-finite token-builder GEMM raises FloatingPointError one update late, with no host synchronisation on the step's path."""
        f32 = self.dense._f32ops if (self.dense is not None and self.device.type == "cuda") else None
        if f32 is not None:
            f32.poll_status()
        out = self._update_feature_fields(*args, **kw)
        f32 = self.dense._f32ops if (self.dense is not None and self.device.type == "cuda") else None
        if f32 is not None:
            f32.snapshot_status()
        return out

    def check_numerics(self):
        """Synchronising form of the same check (end of an episode, tests)."""
        if self.dense is not None and self.dense._f32ops is not None:
            self.dense._f32ops.check_status()

    def _update_feature_fields(self, batch_depth, batch_grid_ft, batch_image=None, batch_position=None, batch_heading=None,
                               batch_camera_intrinsic=None, batch_rot=None, batch_trans=None, depth_scale=1000.0,
                               depth_trunc=1000.0, num_of_views=1, patch_segm=None, view_ids=None, batch_image_ft=None,
                               is_training=False, trainer=None):
        """`num_of_views` (VLN-FF:493: view ix at heading - ix*pi/6) or `view_ids` (PRE-FF:843,920: view ix at
        heading - view_ids[ix]*pi/6, e.g. [0,3,6,9] = the four 90-degree views of `Net_3DFF.forward`, PRE-POL:160)."""
        if is_training and trainer is None:
            raise ValueError("is_training=True needs trainer=train_ff.FFTrainer(...): it collects the loss terms (PRE-FF:969-1047, 1302-1345) "
                             "and hands ground-truth merge decisions back to the memory update (dynam3d_amd/train_ff.py)")
        if not is_training:
            trainer = None
        pinhole = batch_camera_intrinsic is not None
        a = self.args
        if pinhole:
            # "Most 3D datasets" (PRE-FF:849-856, 886-916): raw depth images + pinhole intrinsics + camera->world (R, T) per view.
            depth_raw = self._stack_views(batch_depth)                       # (B,V,Hd,Wd) raw sensor units
            num_of_views, view_ids = depth_raw.shape[1], None
            K0 = _host(batch_camera_intrinsic[0][0])
            self.view_intrinsic = (float(K0[0][0]) * (a.view_width / depth_raw.shape[-1]),    # init_camera_intrinsic (PRE-FF:851-855)
                                   float(K0[1][1]) * (a.view_height / depth_raw.shape[-2]))
            scale_tan = _ray0_tan(self.view_intrinsic[0], a.view_width, a.near, a.far, a.N_samples)
        view_ids = self._view_ids(view_ids)
        if view_ids is not None:
            num_of_views = len(view_ids)
        if self.dense is None:
            raise RuntimeError("Feature_Fields has no weights: call load_state_dict first")
        B, V, P, st, pools, ops = self.batch_size, num_of_views, self.P, self.state, self.pools, self.ops
        K = self.args.num_proposal_instances
        k_max = next(k for k in K_MAX_CHOICES if k >= K)
        if patch_segm is None:
            patch_segm = self.get_patch_segm(batch_image)
        if isinstance(patch_segm, torch.Tensor):
            patch_segm = patch_segm.cpu().numpy()
        segm_all = np.asarray(patch_segm).reshape(B, V, P).astype(np.int32)
        depth24 = None if pinhole else self._dev(batch_depth).view(B, V, P)
        if isinstance(batch_grid_ft, (list, tuple)):
            batch_grid_ft = np.stack([np.asarray(g) for g in batch_grid_ft])
        grid = batch_grid_ft if isinstance(batch_grid_ft, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(batch_grid_ft))
        grid = grid.to(self.device)
        if grid.dtype not in (torch.float16, torch.float32):
            grid = grid.float()
        grid = grid.view(B, V, P, FTS)
        cam = self._camera()
        envs = list(range(B))
        slots_h = np.array([self.slots[e] for e in envs], np.int32)
        slot = self._i32(slots_h)
        self.last_debug = []
        dev_plan = self.planner == "device"
        if dev_plan and trainer is not None:
            raise RuntimeError("is_training=True needs Feature_Fields(planner='host'): the loss terms read the dictionaries on the host")
        for ix in range(V):
            self._grow_rows(max(st.count(e, st.ROWS) for e in envs) + P)
            pools = self.pools
            if dev_plan:
                rb = [st.n_rows[e] for e in envs]
                k0_d, tree_slots_d = ops.ffdev_begin_view(st, slot)            # VLN-FF:532, on the device
            else:
                rb, k0, has_tree = zip(*[st.begin_view(e) for e in envs])
                k0 = [k if t else 0 for k, t in zip(k0, has_tree)]
            row_base = self._i32(rb)        # (re-bound below to a slice of this view's one index upload)
            if pinhole:                                                                                   # PRE-FF:905-916
                cams = pinhole_unproject_rows([_host(batch_camera_intrinsic[e][ix]) for e in envs], [_host(batch_rot[e][ix]) for e in envs],
                                              [_host(batch_trans[e][ix]) for e in envs], scale_tan, depth_scale, depth_trunc)
                ops.unproject_pinhole_append(depth_raw[:, ix].contiguous(), torch.from_numpy(cams).to(self.device), slot, row_base,
                                             a.input_height, a.input_width, a.input_width, pools)
            else:
                pose = self._poses(batch_position, batch_heading, envs,                                 # VLN-FF:550 / PRE-FF:920
                                   view_offset=(ix if view_ids is None else view_ids[ix]) * (-math.pi / 6))
                ops.unproject_append(depth24[:, ix].contiguous(), pose, slot, row_base, cam, pools)
            ops.append_fts(grid[:, ix].contiguous(), slot, row_base, pools)

            # ---- 2D instances of this frame: groups padded to n_max per env -------------------------
            segm = segm_all[:, ix]
            n_seg = segm.max(axis=1) + 1
            n_max = int(n_seg.max())
            order = np.argsort(segm, axis=1, kind="stable")                   # patches by (label, p)
            counts = np.stack([np.bincount(segm[e], minlength=n_max) for e in envs])
            if np.any((counts[np.arange(n_max)[None] < n_seg[:, None]]) == 0):
                raise ValueError("patch_segm labels must be dense 0..n-1 (VLN-FF:416-420)")
            tok_slot = np.repeat(slots_h, P)
            tok_row = (np.asarray(rb, np.int32)[:, None] + order).reshape(-1).astype(np.int32)
            grp_off = np.concatenate([[0], np.cumsum(counts.reshape(-1))]).astype(np.int32)
            G = len(envs) * n_max
            valid_g = np.nonzero(counts.reshape(-1) > 0)[0]
            if dev_plan:
                seg_off = np.concatenate([np.zeros((len(envs), 1), np.int64), np.cumsum(counts, 1)], 1)
                tok_slot_d, tok_row_d, grp_off_d, valid_g_d, n_seg_d, order_d, tok_seg_d, seg_off_d = self._i32_many(
                    tok_slot, tok_row, grp_off, valid_g, n_seg, order, np.take_along_axis(segm, order, 1), seg_off)
                self._grow_slots("inst", max(st.n_slots) + n_max)               # the planner may open n_seg new slots / zones per environment
                self._grow_slots("zone", max(max(st.n_zrows), max(st.n_zids)) + n_max)
                st.ensure(E=max(st.n_edges) + max(st.n_slots) + n_max)
                pools = self.pools
            else:
                tok_slot_d, tok_row_d, grp_off_d, valid_g_d, n_seg_d, k0_d, tree_slots_d = self._i32_many(tok_slot, tok_row, grp_off, valid_g, n_seg, k0,
                                                                                                        self._tree_slots)
            centroid, cell, geom7 = ops.group_stats7(pools, tok_slot_d, tok_row_d, grp_off_d, G, self.cell_len)
            tok_fts = ops.gather_fts(pools, tok_slot_d, tok_row_d)
            if trainer is not None:                                           # differentiable encoding + loss terms of this view (PRE-FF:940-1008)
                img_ix = img_mean = None
                if batch_image_ft is not None:
                    bif = torch.stack([torch.as_tensor(np.asarray(_host(f)) if not isinstance(f, torch.Tensor) else f).to(self.device) for f in batch_image_ft]).float()
                    img_ix, img_mean = bif[:, ix], bif.mean(1)
                new_fts_valid = trainer.instances(self, ops, pools, envs, slots_h, rb, order, tok_fts, geom7, counts, valid_g, centroid, n_max, img_ix, img_mean)
            else:
                new_fts_valid = self.dense.encode_patch_sets(tok_fts, geom7, counts.reshape(-1)[valid_g])
            new_fts = torch.zeros((G, FTS), dtype=torch.float32, device=self.device)
            new_fts.index_copy_(0, valid_g_d.long(), new_fts_valid)

            # ---- KNN proposals + merge discriminator (VLN-FF:604-621) ------------------------------
            ident = all(self.slots[e] == e for e in envs)
            tree_pts = pools.tree_pos if ident else pools.tree_pos.index_select(0, slot.long()).contiguous()
            d2, idx = ops.knn(tree_pts, pools.m_cap * 3, tree_slots_d, centroid, n_max * 3, n_seg_d, k0_d, len(envs), n_max, k_max)
            pair_e, pair_s, pair_j = [], [], []
            for j_, e in enumerate(envs):
                kk = K if dev_plan else k0[j_]       # device planner: every (segment, proposal slot); the planner reads the first k of them
                if kk > 0:
                    ss, jj = np.meshgrid(np.arange(n_seg[j_]), np.arange(kk), indexing="ij")
                    pair_e.append(np.full(ss.size, j_)); pair_s.append(ss.reshape(-1)); pair_j.append(jj.reshape(-1))
            logits_full = torch.zeros((len(envs), n_max, k_max, 2), dtype=torch.float32, device=self.device)
            if pair_e:
                pe_h, ps_h, pj_h = (np.concatenate(x) for x in (pair_e, pair_s, pair_j))
                pe, ps_, pj, pair_new, pair_slot = self._i32_many(pe_h, ps_h, pj_h, pe_h * n_max + ps_h, slots_h[pe_h])   # (int32 index tensors)
                pair_inst = idx[pe, ps_, pj].contiguous()
                if dev_plan:
                    pair_inst.clamp_(min=0)                                   # proposal slots past k0 hold -1: any row will do, their logits are never read
                if trainer is not None:                                       # discriminator loss; the memory merges by ground truth (PRE-FF:1029-1047)
                    logits_full[pe, ps_, pj] = trainer.merge(pools, pair_slot, pe, ps_, pj, pair_inst, len(envs))
                else:
                    x = ops.merge_input(pools, new_fts, centroid, pair_slot, pair_inst, pair_new)
                    logits_full[pe, ps_, pj] = self.dense.merge_logits(x)
            if dev_plan:
                self._finish_view_device(envs, slot, slots_h, order_d, tok_seg_d, seg_off_d, n_seg_d, n_max, k_max, k0_d, d2, idx, logits_full, cell, centroid,
                                         new_fts, rb)
                continue
            d2_h, idx_h, logits_h, cell_h = d2.cpu().numpy(), idx.cpu().numpy(), logits_full.cpu().numpy(), cell.cpu().numpy()  # sync #2

            # ---- bookkeeping: new ids / merges ---------------------------------------------------------
            new_s, new_r, new_src = [], [], []
            m_tok_slot, m_tok_row, m_lens, m_slot, m_inst, m_env = [], [], [], [], [], []
            dbg = []
            for j_, e in enumerate(envs):
                n = int(n_seg[j_])
                cells_e = cell_h[j_ * n_max: j_ * n_max + n]
                k_eff, seg_slot, dirty, doff, drows = st.plan_merge(e, segm[j_], n, k0[j_], k_max, d2_h[j_, :n], idx_h[j_, :n],
                                                                    logits_h[j_, :n], cells_e)
                dbg.append(dict(k_eff=k_eff, seg_slot=seg_slot.copy(), dirty=dirty.copy(), idx=idx_h[j_, :n].copy(),
                                d2=d2_h[j_, :n].copy(), logits=logits_h[j_, :n].copy()))
                for s in np.nonzero(seg_slot >= 0)[0]:
                    new_s.append(self.slots[e]); new_r.append(int(seg_slot[s])); new_src.append(j_ * n_max + int(s))
                for i, inst in enumerate(dirty):
                    rows = drows[doff[i]:doff[i + 1]]
                    m_tok_slot.append(np.full(len(rows), self.slots[e], np.int32)); m_tok_row.append(rows)
                    m_lens.append(len(rows)); m_slot.append(self.slots[e]); m_inst.append(int(inst)); m_env.append(j_)
            self.last_debug.append(dbg)
            self._grow_slots("inst", max(st.count(e, st.SLOTS) for e in envs))
            if trainer is not None:
                trainer.new_instances(pools, new_s, new_r, new_src)
            if new_r:                                                        # VLN-FF:643-648
                s, r, src = self._i32_many(new_s, new_r, new_src)
                ops.scatter_rows(pools.inst_pos, s, r, centroid, src)
                ops.scatter_rows(pools.inst_fts, s, r, new_fts, src)
            dirty_cells = np.zeros((0, 3), np.int32)
            if m_lens:                                                       # VLN-FF:662-688
                ts, tr, goff, gs, gi = self._i32_many(np.concatenate(m_tok_slot), np.concatenate(m_tok_row), np.concatenate([[0], np.cumsum(m_lens)]),
                                                      m_slot, m_inst)
                _, mcell, mgeom = ops.group_stats7(pools, ts, tr, goff, len(m_lens), self.cell_len, pools.inst_pos, gs, gi)
                mfts = ops.gather_fts(pools, ts, tr)
                merged = self.dense.encode_patch_sets(mfts, mgeom, m_lens)
                ops.scatter_rows(pools.inst_fts, gs, gi, merged)
                dirty_cells = mcell.cpu().numpy()                           # sync #3
            # ---- zones (VLN-FF:694-756 / 777-812) -------------------------------------------------------
            z_tok_slot, z_tok_inst, z_lens, z_mode, z_slot, z_row = [], [], [], [], [], []
            m_env_a = np.asarray(m_env, np.int64)
            for j_, e in enumerate(envs):
                zrow, zmode, zoff, zmem = st.plan_zones(e, dirty_cells[m_env_a == j_] if len(m_env_a) else dirty_cells, int(n_seg[j_]))
                for t in range(len(zrow)):
                    mem = zmem[zoff[t]:zoff[t + 1]]
                    z_tok_slot.append(np.full(len(mem), self.slots[e], np.int32)); z_tok_inst.append(mem)
                    z_lens.append(len(mem)); z_mode.append(int(zmode[t])); z_slot.append(self.slots[e]); z_row.append(int(zrow[t]))
            self._grow_slots("zone", max(st.count(e, st.ZROWS) for e in envs))
            if z_lens:
                ts, ti, goff, gs, gr, zm = self._i32_many(np.concatenate(z_tok_slot) if sum(z_lens) else np.zeros(0, np.int32),
                                                          np.concatenate(z_tok_inst) if sum(z_lens) else np.zeros(0, np.int32),
                                                          np.concatenate([[0], np.cumsum(z_lens)]), z_slot, z_row, z_mode)
                geom4 = ops.group_stats4(pools, ts, ti, goff, zm, gs, gr, len(z_lens), self.cell_len)
                ifts = ops.gather_rows(pools.inst_fts, ts, ti) if sum(z_lens) else torch.zeros((0, FTS), device=self.device)
                zfts = self.dense.encode_zone_sets(ifts, geom4, z_lens)
                ops.scatter_rows(pools.zone_fts, gs, gr, zfts)
            self._snapshot_tree()

    def _finish_view_device(self, envs, slot, slots_h, order_d, tok_seg_d, seg_off_d, n_seg_d, n_max, k_max, k0_d, d2, idx, logits, cell, centroid, new_fts, rb):
        """New / merge bookkeeping and the zone update of one view with the planner on the device (VLN-FF:623-756): every decision is a
        kernel (csrc/ff_plan.h) fed by the kernels before it; the host reads ONE report (sizes of the merged and the zone sets) and then
        launches the two set encoders."""
        st, ops, pools, B, P = self.state, self.ops, self.pools, len(envs), self.P
        G_ub = B * n_max
        rows_stride = max(rb) + P
        # one int32 block for everything the host reads back: report words | totals (merge) | totals (zones) | group offsets x 2
        nt_ = 2 + 2 * B
        rep = torch.zeros((B * REPORT_WORDS + 2 * nt_ + 2 * (G_ub + 1),), dtype=torch.int32, device=self.device)
        o = B * REPORT_WORDS
        report, tot_m, tot_z = rep[:o], rep[o:o + nt_], rep[o + nt_:o + 2 * nt_]
        o += 2 * nt_
        goff, zgoff = rep[o:o + G_ub + 1], rep[o + G_ub + 1:]
        seg_slot, dirty_inst, dirty_off, dirty_rows = ops.ffdev_plan_merge(st, slot, order_d, tok_seg_d, seg_off_d, n_seg_d, n_max, k_max, k0_d, d2, idx, logits,
                                                                           cell, rows_stride, report)
        grp_slot_new = self._i32(np.repeat(slots_h, n_max))
        ops.scatter_rows(pools.inst_pos, grp_slot_new, seg_slot.view(-1), centroid)                 # VLN-FF:643-648 (rows -1 = merged segments: skipped)
        ops.scatter_rows(pools.inst_fts, grp_slot_new, seg_slot.view(-1), new_fts)
        ts, tr, gs, gi = ops.ffdev_flatten_merge(slot, n_max, dirty_inst, dirty_off, dirty_rows, report, goff, tot_m)
        _, mcell, mgeom = ops.group_stats7(pools, ts, tr, goff, G_ub, self.cell_len, pools.inst_pos, gs, gi)    # merged centroids -> inst_pos, their cells
        zone_row, zone_mode, zone_off, zone_mem = ops.ffdev_plan_zones(st, slot, dirty_inst, mcell, cell, n_seg_d, n_max, pools.m_cap, report)
        zts, zti, zmode, zgs, zgr = ops.ffdev_flatten_zones(slot, n_max, zone_row, zone_mode, zone_off, zone_mem, report, zgoff, tot_z)
        rep_h = rep.cpu().numpy()                                                                   # the view's ONE device-to-host read
        for j_, e in enumerate(envs):
            st.n_rows[e] += P
        o = B * REPORT_WORDS
        st.take_report_envs(envs, rep_h[:o].reshape(B, REPORT_WORDS))
        n_m, T_m = int(rep_h[o]), int(rep_h[o + 1])
        n_z, T_z = int(rep_h[o + nt_]), int(rep_h[o + nt_ + 1])
        o += 2 * nt_
        goff_h, zgoff_h = rep_h[o:o + G_ub + 1], rep_h[o + G_ub + 1:]
        if n_m:                                                                                      # VLN-FF:662-688
            mfts = ops.gather_fts(pools, ts[:T_m], tr[:T_m])
            merged = self.dense.encode_patch_sets(mfts, mgeom[:T_m], np.diff(goff_h[:n_m + 1]).tolist())
            ops.scatter_rows(pools.inst_fts, gs[:n_m], gi[:n_m], merged)
        if n_z:                                                                                      # VLN-FF:694-756 / 777-812
            z_lens = np.diff(zgoff_h[:n_z + 1]).tolist()
            geom4 = ops.group_stats4(pools, zts[:T_z], zti[:T_z], zgoff[:n_z + 1], zmode[:n_z], zgs[:n_z], zgr[:n_z], n_z, self.cell_len)
            ifts = ops.gather_rows(pools.inst_fts, zts[:T_z], zti[:T_z]) if T_z else torch.zeros((0, FTS), device=self.device)
            zfts = self.dense.encode_zone_sets(ifts, geom4, z_lens)
            ops.scatter_rows(pools.zone_fts, zgs[:n_z], zgr[:n_z], zfts)
        self._snapshot_tree()

    # ---- a12 (VLN-FF:818-862) -----------------------------------------------------------------------------------
    @torch.no_grad()
    def get_environment_features(self, agent_position, agent_heading_angle, instance_distance=5.0, zone_distance=100.0):
        B, st, pools = self.batch_size, self.state, self.pools
        envs = list(range(B))
        pose = self._poses(agent_position, agent_heading_angle, envs)
        slot = self._i32([self.slots[e] for e in envs])
        out = {}
        if self.planner == "device":                                       # dict order (VLN-FF:825, 844) ranked on the device
            mx = max(1, max(st.n_slots, default=0), max(st.n_zids, default=0))
            inst_ids, n_inst, zone_ids, n_zone = self.ops.ffdev_live_ids(st, slot, mx)
            out["instance"] = self.ops.agent_frame_compact(pools.inst_pos, pools.inst_fts, slot, inst_ids, n_inst, pose, float(instance_distance))
            out["zone"] = self.ops.agent_frame_compact(pools.zone_pos, pools.zone_fts, slot, zone_ids, n_zone, pose, float(zone_distance))
        else:
            ids = [st.live_ids(e) for e in envs]
            for which, (pp, pf, radius, key) in enumerate(((pools.inst_pos, pools.inst_fts, instance_distance, "instance"),
                                                           (pools.zone_pos, pools.zone_fts, zone_distance, "zone"))):
                n_ids = np.array([len(i[which]) for i in ids], np.int32)
                mx = max(1, int(n_ids.max()))
                pad = np.zeros((B, mx), np.int32)
                for e in envs:
                    pad[e, :n_ids[e]] = ids[e][which]
                rel, fts, kept, count = self.ops.agent_frame_compact(pp, pf, slot, self._i32(pad), self._i32(n_ids), pose, float(radius))
                out[key] = (rel, fts, kept, count)
        cc = torch.stack([out["instance"][3], out["zone"][3]]).cpu().numpy()          # sync #4: Ni, Nz (one read)
        ci, cz = cc[0], cc[1]
        return {
            "batch_instance_fts": [out["instance"][1][e, :ci[e]] for e in envs],
            "batch_instance_relative_position": [out["instance"][0][e, :ci[e]] for e in envs],
            "batch_zone_fts": [out["zone"][1][e, :cz[e]] for e in envs],
            "batch_zone_relative_position": [out["zone"][0][e, :cz[e]] for e in envs],
            "batch_instance_ids": [out["instance"][2][e, :ci[e]] for e in envs],
            "batch_zone_ids": [out["zone"][2][e, :cz[e]] for e in envs],
        }

    # ---- a13 (VLN-FF:296-326) -------------------------------------------------------------------------------------
    @torch.no_grad()
    def get_patch_3d_info(self, batch_depth_map):
        d = self._dev(batch_depth_map).view(-1, self.P)
        outs = self.ops.patch_3d_info(d, self._camera())
        return tuple(o.unsqueeze(-1) for o in outs)

    # ---- test / debug export ------------------------------------------------------------------------------------------
    def export_env(self, e: int):
        st, pools, s = self.state, self.pools, self.slots[e]
        ex = st.export(s if self.planner == "device" else e)
        nr, ns, nz = st.count(e, st.ROWS), st.count(e, st.SLOTS), st.count(e, st.ZROWS)
        ex.update(rows_pos=pools.rows_pos[s, :nr].cpu().numpy(), ipos=pools.inst_pos[s, :ns].cpu().numpy(),
                  ifts=pools.inst_fts[s, :ns].cpu().numpy(), zpos=pools.zone_pos[s, :nz].cpu().numpy(),
                  zfts=pools.zone_fts[s, :nz].cpu().numpy())
        L = np.array(self.cell_len, np.float32)
        ex["zkey"] = {tuple((np.array(c, np.float32) * L + L / 2).tolist()): z for c, z in ex["zkey_cells"].items()}
        return ex
