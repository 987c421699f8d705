"""Container-only harness that imports the REFERENCE's own Feature_Fields / VisionTransformer.

TEST INFRASTRUCTURE -- never imported by the product (`dynam3d_amd/`), by `bench.py`'s GPU leg
or by any `-m gpu` test.  It exists to (a) generate the golden vectors committed under
`tests/golden/` (see `tests/golden/gen_golden.py`) and (b) validate `oracle/ff_oracle.py`
against the real reference while `/root/reference` is mounted (this container only; the GPU box
has no `/root/reference`).

Nothing from the reference is copied: the files are loaded in place with
`importlib.util.spec_from_file_location` after pre-seeding `sys.modules` with stubs for the
un-installed third-party packages (recipe: SURVEY.md section 10).

Stub semantics that DEFINE parity for un-vendored dependencies ("parity unpinned" upstream):
  * torch_kdtree.build_kd_tree(points).query(q, nr_nns_searches=k)
        -> brute force, ascending (dist^2, index); d2 = (dx*dx + dy*dy) + dz*dz in float32, stable
           sort (torch_kdtree itself evaluates exact squared differences; SURVEY's cdist probe did not)
  * tinycudann.Network -> bias-free MLP, LeakyReLU(0.01), see `TcnnStub`.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF_ROOT = os.environ.get("DYNAM3D_REFERENCE", "/root/reference")
VLN_FF = os.path.join(REF_ROOT, "Dynam3D_VLN/vlnce_baselines/models/feature_fields.py")
PRE_FF = os.path.join(REF_ROOT, "Dynam3D_Pretrain/src_3dff/models/feature_fields.py")
CLIP_MODEL = os.path.join(REF_ROOT, "Dynam3D_VLN/vlnce_baselines/models/encoders/clip/model.py")


def reference_available() -> bool:
    return os.path.isfile(VLN_FF)


# --------------------------------------------------------------------------------------------
# stubs
# --------------------------------------------------------------------------------------------
class _BruteTree:
    """Defines KNN parity: ascending (dist^2, idx), ties -> lowest index."""

    def __init__(self, pts):
        if isinstance(pts, np.ndarray):
            pts = torch.from_numpy(pts)
        # torch_kdtree copies the points when it builds the tree -> snapshot semantics.
        self.pts = pts.detach().clone().float()

    def query(self, q, nr_nns_searches=1):
        if isinstance(q, np.ndarray):
            q = torch.from_numpy(q)
        q = q.float()
        p = self.pts.to(q.device)
        k = int(nr_nns_searches)
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d = (dx * dx + dy * dy) + dz * dz          # one rounding per op, fixed order
        ds, idx = torch.sort(d, dim=-1, stable=True)  # ascending (dist^2, index)
        return ds[:, :k].contiguous(), idx[:, :k].contiguous()


class TcnnStub(torch.nn.Module):
    """tinycudann.Network('CutlassMLP') stand-in DEFINING the parity target for a22: bias-free layers,
    y = act(x W^T) with fp16 weights and activations, fp32 accumulation, fp16 store after every layer."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        nh = int(network_config["n_hidden_layers"])
        nn_ = int(network_config["n_neurons"])
        dims = [n_input_dims] + [nn_] * nh + [n_output_dims]
        self.act = network_config.get("activation", "None")
        self.out_act = network_config.get("output_activation", "None")
        self.layers = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)])

    @staticmethod
    def _apply(name, x):
        if name in ("LeakyReLU", "leakyrelu"):
            return torch.nn.functional.leaky_relu(x, 0.01)
        if name in ("ReLU", "relu"):
            return torch.relu(x)
        if name in ("None", "none", None):
            return x
        raise NotImplementedError(name)

    def forward(self, x):
        h = x.to(torch.float16)
        for i, l in enumerate(self.layers):
            y = h.float() @ l.weight.to(torch.float16).float().t()
            y = self._apply(self.act if i < len(self.layers) - 1 else self.out_act, y)
            h = y.to(torch.float16)
        return h


class _NdArrayEq(np.ndarray):
    """numpy>=2 raises on `arr == []`; the reference tests emptiness that way (VLN-FF:557,567)."""

    def __eq__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return False
        return np.ndarray.__eq__(self, other)

    __hash__ = None


class _ViewIds:
    """`view_ids` stand-in for the Pretrain class under numpy >= 2: the reference computes
    `view_ids[ix].cpu().numpy() * (-math.pi / 6) + heading` (PRE-FF:696, 920), a numpy float64 SCALAR, and adds it to float32
    arrays.  With the numpy 1.x the reference was written for, value-based casting keeps those arrays float32; numpy >= 2
    (NEP 50) promotes them to float64 and the float32 `nn.Linear`s then reject them.  Handing the index back as a Python int
    makes the offset a Python float -- 'weak' under NEP 50 -- which reproduces the numpy 1.x arithmetic exactly."""

    class _Item:
        def __init__(self, v):
            self.v = int(v)

        def cpu(self):
            return self

        def numpy(self):
            return self.v

    def __init__(self, ids):
        self.ids = [int(i) for i in ids]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, i):
        return self._Item(self.ids[i])

    def __eq__(self, other):
        return False if other is None else NotImplemented

    __hash__ = None


class _StoreList(list):
    def __setitem__(self, k, v):
        if isinstance(v, np.ndarray) and not isinstance(v, _NdArrayEq):
            v = v.view(_NdArrayEq)
        list.__setitem__(self, k, v)


def _install_stubs(pkg: str):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("torch_kdtree", build_kd_tree=lambda pts, *a, **k: _BruteTree(pts))
    mod("open3d")
    import argparse

    mod("configargparse", ArgumentParser=argparse.ArgumentParser)
    p = mod(pkg)
    p.__path__ = []
    pm = mod(pkg + ".models")
    pm.__path__ = []

    class FastSAM:  # noqa: D401 - dummy
        def __init__(self, *a, **k):
            pass

    class FastSAMPrompt:
        def __init__(self, *a, **k):
            pass

    mod(pkg + ".models.fastsam", FastSAM=FastSAM, FastSAMPrompt=FastSAMPrompt)
    mod("tinycudann", Network=TcnnStub)
    torch.cuda.get_device_properties = lambda *a, **k: SimpleNamespace(total_memory=64 * 1024 ** 3)
    torch.cuda.memory_allocated = lambda *a, **k: 0


def load_ref_module(which: str = "vln"):
    path, pkg = (VLN_FF, "vlnce_baselines") if which == "vln" else (PRE_FF, "src_3dff")
    _install_stubs(pkg)
    argv = sys.argv
    sys.argv = ["x"]
    try:
        spec = importlib.util.spec_from_file_location("ref_ff_" + which, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def load_ref_clip():
    spec = importlib.util.spec_from_file_location("ref_clip_model", CLIP_MODEL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class RefFeatureFields:
    """Drives the reference's VLN `Feature_Fields` on CPU with a synthetic segmentation."""

    def __init__(self, batch_size: int, state_dict=None, which="vln"):
        self.mod = load_ref_module(which)
        argv = sys.argv
        sys.argv = ["x"]
        try:
            self.F = self.mod.Feature_Fields(batch_size=batch_size, device="cpu").eval()
        finally:
            sys.argv = argv
        if state_dict is not None:
            missing, unexpected = self.F.load_state_dict(state_dict, strict=False)
            assert not [k for k in missing if "FastSAM" not in k], missing
        self._segm = None
        self.F.get_patch_segm = self._get_patch_segm
        self.reset(batch_size)

    def _get_patch_segm(self, imgs, **kw):
        s = self._segm
        assert s is not None and s.shape[0] == len(imgs)
        return s.clone()

    def _wrap_stores(self):
        for name in ("global_patch_position", "global_patch_fts", "global_patch_scales", "global_patch_directions"):
            setattr(self.F, name, _StoreList(getattr(self.F, name)))

    def reset(self, batch_size):
        self.F.reset(batch_size)
        self.F.initialize_camera_setting(90.0, 90.0)
        self._wrap_stores()

    @torch.no_grad()
    def step_pretrain(self, depth_full, depth24, grid_fts, patch_segm, positions, headings, view_ids):
        """The Pretrain class's per-step pair (PRE-POL:188-189) in inference mode: delete + update with `view_ids`
        (is_training=False, no GT point cloud).  Same tensor conventions as `step`."""
        F = self.F
        B, V = F.batch_size, len(view_ids)
        vid = _ViewIds(view_ids)
        F.delete_old_features_from_camera_frustum(depth_full, positions, headings, view_ids=vid)
        self._wrap_stores()
        self._segm = patch_segm
        img = np.zeros((B, V, 8, 8, 3), np.uint8)
        F.update_feature_fields(depth24, grid_fts, batch_image=img, batch_position=positions, batch_heading=headings,
                                view_ids=vid, is_training=False)

    @torch.no_grad()
    def step(self, depth_full, depth24, grid_fts, patch_segm, positions, headings, num_of_views=1, delete=True):
        """depth_full (B,V,Hd,Wd) f32 metres torch; depth24 (B,V,576) np f32 metres;
        grid_fts (B,V,576,768) np f32; patch_segm (B*V,1,24,24) int64 torch; positions list of np(3,) habitat."""
        F = self.F
        B = F.batch_size
        if delete:
            F.delete_old_features_from_camera_frustum(depth_full, positions, headings, num_of_views=num_of_views)
        self._segm = patch_segm
        img = np.zeros((B, num_of_views, 8, 8, 3), np.uint8)
        F.update_feature_fields(depth24, grid_fts, batch_image=img, batch_position=positions,
                                batch_heading=headings, num_of_views=num_of_views)
        return F.get_environment_features(positions, headings)


# --------------------------------------------------------------------------------------------
# PRE-FF:843-1345 with is_training=True, executed on the CPU
# --------------------------------------------------------------------------------------------
class _RecordingTree:
    """Wraps a GT point-cloud tree: every k = 1 query's nearest-point indices are logged (PRE-FF:978: one call per 2D segment)."""

    def __init__(self, tree, log):
        self.tree, self.log = tree, log

    def query(self, q, nr_nns_searches=1):
        d, idx = self.tree.query(q, nr_nns_searches=nr_nns_searches)
        self.log.append(idx[:, 0].numpy().astype(np.int64).copy())
        return d, idx


class _PromotingMatmul:
    """The training branch multiplies float32 predictions with the float16 CLIP targets (`contrastive_loss`, PRE-FF:836); under the CUDA
    autocast the reference trains in (PRE-TR:501) that runs, on the CPU `torch.matmul` rejects mixed dtypes.  This context promotes mixed
    operands to their common dtype (float32) -- the ONE shim the CPU execution needs; everything else is the reference's own code."""

    def __enter__(self):
        self.orig = torch.matmul

        def mm(a, b, *args, **kw):
            if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.dtype != b.dtype:
                dt = torch.promote_types(a.dtype, b.dtype)
                a, b = a.to(dt), b.to(dt)
            return self.orig(a, b, *args, **kw)

        torch.matmul = mm
        return self

    def __exit__(self, *exc):
        torch.matmul = self.orig
        return False


class RefTrainingRun(RefFeatureFields):
    """Drives the reference's Pretrain `Feature_Fields.update_feature_fields(is_training=True)` -- GT labelling (PRE-FF:976-986), GT merges
    (PRE-FF:1029-1047), loss assembly (PRE-FF:1302-1345) -- on the CPU: habitat mode, view_ids (0, 3, 6, 9), a synthetic GT instance point
    cloud per environment, CLIP image features per view.  The module is in eval() mode: dropout (0.1 inside the encoder layers while
    the reference trains) is OFF, i.e. the goldens made with this class encode the p = 0 arithmetic."""

    def __init__(self, batch_size, state_dict, gt_xyz, gt_label):
        super().__init__(batch_size, state_dict, which="pre")
        self.gt_log = [[] for _ in range(batch_size)]
        # mode "scannet": the GT cloud is used as given ("habitat" subtracts the 1.25 m agent height from z, PRE-FF:296-297)
        self.F.reset(batch_size, mode="scannet", batch_gt_pcd_xyz=[torch.from_numpy(np.asarray(x, np.float32)) for x in gt_xyz],
                     batch_gt_pcd_label=[torch.from_numpy(np.asarray(l, np.int64)) for l in gt_label])
        self.F.initialize_camera_setting(90.0, 90.0)
        self._wrap_stores()
        self.F.gt_pcd_tree = [_RecordingTree(t, self.gt_log[b]) for b, t in enumerate(self.F.gt_pcd_tree)]

    def train_step(self, depth_full, depth24, grid_fts, patch_segm, positions, headings, view_ids, image_ft, backward=True):
        """One delete + update(is_training=True).  image_ft (B, V, 768) float32 torch.  Returns a dict with the two losses, the returned
        GT-id / feature lists, the cross-entropy calls' (score, target) pairs and -- with `backward` -- every parameter's gradient of
        sim_loss + segm_loss."""
        import torch.nn.functional as TF
        F = self.F
        B, V = F.batch_size, len(view_ids)
        vid = _ViewIds(view_ids)
        for l in self.gt_log:
            l.clear()
        with torch.no_grad():
            F.delete_old_features_from_camera_frustum(depth_full, positions, headings, view_ids=vid)
        self._wrap_stores()
        self._segm = patch_segm
        img = np.zeros((B, V, 8, 8, 3), np.uint8)
        ce_calls = []
        orig_ce = TF.cross_entropy

        def ce(inp, tgt, *a, **k):
            if inp.dim() == 2 and inp.shape[-1] == 2:
                ce_calls.append((inp.detach().numpy().copy(), tgt.detach().numpy().copy()))
            return orig_ce(inp, tgt, *a, **k)

        F.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy = ce
        try:
            with _PromotingMatmul():
                out = F.update_feature_fields(depth24, grid_fts, batch_image=img, batch_image_ft=image_ft, batch_position=positions,
                                              batch_heading=headings, view_ids=vid, is_training=True)
        finally:
            torch.nn.functional.cross_entropy = orig_ce
        sim, segm, gt3d, pred3d, gt_in_zone, pred_zone = out
        res = dict(sim_loss=float(sim.detach()), segm_loss=float(segm.detach()) if isinstance(segm, torch.Tensor) else float(segm),
                   has_segm=isinstance(segm, torch.Tensor), gt3d=[g.numpy().astype(np.int64).copy() for g in gt3d],
                   pred3d=[p.detach().numpy().copy() for p in pred3d],
                   gt_in_zone=[[z.numpy().astype(np.int64).copy() for z in zs] for zs in gt_in_zone],
                   pred_zone=[[z.detach().numpy()[0].copy() for z in zs] for zs in pred_zone],
                   ce_calls=ce_calls, gt_nn=[[a.copy() for a in l] for l in self.gt_log],
                   row_gt=[np.asarray(g.numpy() if isinstance(g, torch.Tensor) else g, np.int64).copy() for g in F.global_gt_instance_ids])
        if backward:
            loss = sim + segm if isinstance(segm, torch.Tensor) else sim
            loss.backward()
            res["grads"] = {k: (p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
                            for k, p in F.named_parameters() if not k.startswith(("FastSAM", "nerf_", "patch_to_nerf_", "aggregate_patch_to_nerf_"))}
        return res
