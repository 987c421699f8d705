"""Multi-view front-end of the Pretrain policy net: `Net_3DFF.forward` up to the memory update.

Reference: Dynam3D_Pretrain/src_3dff/models/Policy_3DFF.py:136-189 (PRE-POL).  Per step the reference
  * places the 12 panorama sensors clockwise (`ra = (12 - a) % 12` for the a-th depth key, PRE-POL:156-163),
  * keeps views `view_ids = [0, 3, 6, 9]` (PRE-POL:165; four 90-degree views, view ix looks along heading - view_ids[ix]*pi/6),
  * CLIP-encodes those B*4 images (PRE-POL:176), nearest-resizes each depth to 24x24 and `preprocess_depth`s it
    (PRE-POL:179-185), and calls `delete_old_features_from_camera_frustum` + `update_feature_fields` with `view_ids`
    (PRE-POL:188-189).
What follows in the reference's forward (depth ResNet encoder, waypoint predictor, NMS) is Habitat control plane and out of
scope (not on the hot path).  Only the four kept views are ever touched here: the 12-image batch the reference
builds first exists only for that depth encoder.

Same kernels as the VLN step: `d3d_resize_normalize` + the ViT tower, `d3d_resize_nearest_preprocess`,
`d3d_preprocess_depth`, `d3d_frustum_cull`, `d3d_unproject_append`, ... -- through `Feature_Fields(variant="pretrain")`."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .feature_fields import Feature_Fields
from .modules import RefreshOnChange
from .towers import ClipEncoder, VitConfig, preprocess_rgb
from .weights import ff_param_spec

NUM_IMGS = 12
VIEW_IDS = (0, 3, 6, 9)


def clockwise_sources(observations: Dict[str, torch.Tensor], view_ids: Sequence[int] = VIEW_IDS) -> List[str]:
    """Depth key whose image lands in clockwise slot `v` for every kept view (PRE-POL:156-163): the a-th depth key of the
    dict (insertion order, 'You might need to double check the keys order') goes to slot (12 - a) % 12."""
    depth_keys = [k for k in observations if "depth" in k]
    if len(depth_keys) != NUM_IMGS:
        raise ValueError(f"expected {NUM_IMGS} depth sensors in the observation dict, got {len(depth_keys)}")
    slot_to_key = {(NUM_IMGS - a) % NUM_IMGS: k for a, k in enumerate(depth_keys)}
    return [slot_to_key[int(v)] for v in view_ids]


class Net_3DFF(RefreshOnChange):
    """`torch.nn.Module` like the reference's `Net_3DFF` (PRE-POL:66-115): `feature_fields.*` and `rgb_encoder.model.visual.*`
    parameters under the reference's keys (modules.py)."""

    def __init__(self, vit: VitConfig, weights: Dict[str, torch.Tensor], device="cuda", batch_size: int = 1, ops=None,
                 clip_dtype=torch.float16, segmenter=None, max_steps: int = 16, depth_scale=(0.0, 10.0)):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type == "cuda":
            from . import dense_ops as D
            D.enable_hip_kernels(["all"])                                # the CLIP tower on the hand-written HIP kernels
        ff_sd = {k: weights[k] for k, _ in ff_param_spec(768)}
        self.feature_fields = Feature_Fields(batch_size, device, ff_sd, ops=ops, segmenter=segmenter, max_steps=max_steps,
                                             max_views=len(VIEW_IDS), variant="pretrain")
        self.ops = self.feature_fields.ops
        self.rgb_encoder = ClipEncoder(weights, vit, clip_dtype, device)
        self._init_refresh_hooks()
        self.depth_scale = depth_scale                                  # PRE-POL:121-122 (R2R: 0 .. 10 m)
        self.positions: List = [0 for _ in range(batch_size)]          # set by the caller before forward (PRE-POL:104-105)
        self.headings: List = [0 for _ in range(batch_size)]

    def refresh(self):
        self.feature_fields.refresh()
        self.rgb_encoder.refresh()

    def preprocess_depth(self, depth):
        """PRE-POL:118-133; (N,H,W,1) in [0,1] -> metres, zero pixels <- column max."""
        d = depth.to(self.device, torch.float32)
        return self.ops.preprocess_depth(d.reshape(d.shape[0], d.shape[1], d.shape[2]), *self.depth_scale).view(d.shape)

    @torch.no_grad()
    def forward(self, observations: Dict[str, torch.Tensor], waypoint_predictor=None, patch_segm=None, in_train: bool = False,
                view_ids: Sequence[int] = VIEW_IDS, **unused) -> Dict[str, torch.Tensor]:
        """observations: the 12 'depth*' (B,H,W,1) float32 in [0,1] and 'rgb*' (B,h,w,3) uint8 panorama sensors.
        Updates `self.feature_fields` exactly as PRE-POL:188-189 does in inference mode and returns what the reference keeps
        on the way: `rgb_embedding` (B,V,768) CLIP class tokens, `grid_fts` (B,V,576,768), `depth24` (B,V,576) metres."""
        if waypoint_predictor is not None:
            raise NotImplementedError("waypoint prediction (PRE-POL:192-260) is outside the hot path: SURVEY.md 8")
        if in_train:
            raise NotImplementedError("pre-training losses: SURVEY.md 8f-1")
        ff = self.feature_fields
        B, V = ff.batch_size, len(view_ids)
        keys = clockwise_sources(observations, view_ids)
        depth = torch.stack([observations[k] for k in keys], 1).to(self.device, torch.float32)                  # (B,V,H,W,1)
        rgb = torch.stack([observations[k.replace("depth", "rgb")] for k in keys], 1).to(self.device)           # (B,V,h,w,3)
        rgb = rgb.view(B * V, *rgb.shape[2:])
        depth = depth.view(B * V, *depth.shape[2:])
        cls, grid = self.rgb_encoder.tower.forward(preprocess_rgb(rgb))                                                 # PRE-POL:176
        a = ff.args
        depth24 = self.ops.resize_nearest_preprocess(depth[..., 0], a.input_height, a.input_width, *self.depth_scale).view(B, V, -1)
        origin_depth = self.ops.preprocess_depth(depth[..., 0], *self.depth_scale).view(B, V, depth.shape[1], depth.shape[2])
        # "Do not change the order of the following two lines" (PRE-POL:187)
        ff.delete_old_features_from_camera_frustum(origin_depth, self.positions, self.headings, view_ids=view_ids)
        ff.update_feature_fields(depth24, grid.view(B, V, ff.P, -1), rgb, batch_position=self.positions, batch_heading=self.headings,
                                 view_ids=view_ids, patch_segm=patch_segm, is_training=False)
        return dict(rgb_embedding=cls.view(B, V, -1), grid_fts=grid.view(B, V, ff.P, -1), depth24=depth24)
