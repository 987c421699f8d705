"""a6: `Feature_Fields.get_patch_segm` (VLN-FF:399-430) around a pluggable mask generator.

The reference runs FastSAM (YOLOv8-seg, `everything_prompt`) per image and then turns its masks into the 24x24 dense label
map the memory update consumes.  The network itself is outside the hot path; everything AFTER it is here,
on the device, in one launch per batch (`d3d_patch_segm_from_masks`): 'last mask wins' label image, nearest resize to the
patch grid, dense relabel in `torch.unique` order; a failing / empty segmentation gives an all-zero map exactly like the
reference's `except` branch (VLN-FF:424-426)."""
from __future__ import annotations

import ctypes as _C
from typing import Callable, Optional, Sequence

import torch

from . import _lib

_lib.register("d3d_segment_slic", [_C.c_void_p, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_int32, _C.c_float, _C.c_void_p,
                                   _C.c_void_p, _C.c_void_p])


class MaskSegmenter:
    """segmenter(batch_image) -> (N,1,h,w) int64 dense labels.  `mask_fn(image) -> (n,H,W)` tensor of {0,1} masks (any dtype,
    any device) or None; exceptions raised by it are treated like the reference treats FastSAM errors."""

    def __init__(self, mask_fn: Callable, ops, grid_hw=(24, 24), device="cuda", verbose: bool = False):
        self.mask_fn, self.ops, self.grid_hw, self.device, self.verbose = mask_fn, ops, grid_hw, torch.device(device), verbose

    def __call__(self, batch_image: Sequence, **kw) -> torch.Tensor:
        per_image, shape = [], None
        for img in batch_image:
            try:
                m = self.mask_fn(img, **kw)
                m = None if m is None or len(m) == 0 else (torch.as_tensor(m).to(self.device) == 1).to(torch.uint8)
            except Exception as e:                                       # VLN-FF:424: "FastSAM error, skip..."
                if self.verbose:
                    print("segmenter error, skip...", e)
                m = None
            if m is not None:
                if shape is not None and tuple(m.shape[1:]) != shape:
                    raise ValueError("all images of a batch must share one mask resolution")
                shape = tuple(m.shape[1:])
            per_image.append(m)
        h, w = self.grid_hw
        if shape is None:                                                # nothing segmented at all
            return torch.zeros((len(per_image), 1, h, w), dtype=torch.int64, device=self.device)
        off = [0]
        for m in per_image:
            off.append(off[-1] + (0 if m is None else m.shape[0]))
        masks = torch.cat([m for m in per_image if m is not None]).contiguous()
        segm, _ = self.ops.patch_segm_from_masks(masks, off, h, w)
        return segm


class SlicSegmenter:
    """segmenter(batch_image) -> (N,1,24,24) int64 dense labels with the masks GENERATED ON THE DEVICE: `d3d_segment_slic`
    (grid-seeded colour + position k-means, csrc/segment_kernels.hip) produces gx * gy disjoint masks per frame in one launch for the whole
    batch, `d3d_patch_segm_from_masks` turns them into the label map exactly as it does for FastSAM's masks (VLN-FF:411-420).  A stand-in for
    the reference's FastSAM network (whose weights are not available offline), behind the same callable: pass it as
    `Feature_Fields(segmenter=SlicSegmenter(ops))` / `Dynam3D_VLN(segmenter=...)` and the step needs no `patch_segm` input."""

    def __init__(self, ops, seeds=(4, 4), iters: int = 5, compactness: float = 20.0, grid_hw=(24, 24), device="cuda"):
        self.ops, self.seeds, self.iters, self.compactness, self.grid_hw, self.device = ops, seeds, iters, compactness, grid_hw, torch.device(device)
        self._C = _C

    def masks(self, batch_image, return_labels: bool = False):
        """batch_image: (N,H,W,3) uint8 tensor / array (or a sequence of (H,W,3) images of one size) -> masks (N, K, H, W) uint8 on the device."""
        C = self._C
        if isinstance(batch_image, (list, tuple)):
            batch_image = torch.stack([torch.as_tensor(i) for i in batch_image])
        img = torch.as_tensor(batch_image).to(self.device, torch.uint8).contiguous()
        if img.dim() == 5:                                              # (B, V, H, W, 3) -> frames, environment-major like the reference's batch_image
            img = img.reshape(-1, *img.shape[2:])
        N, H, W, _ = img.shape
        gy, gx = self.seeds
        masks = torch.empty((N, gx * gy, H, W), dtype=torch.uint8, device=self.device)
        labels = torch.empty((N, H, W), dtype=torch.int32, device=self.device) if return_labels else None
        _lib.check(self.ops.lib.d3d_segment_slic(C.c_void_p(img.data_ptr()), N, H, W, gx, gy, self.iters, float(self.compactness), C.c_void_p(masks.data_ptr()),
                                                 None if labels is None else C.c_void_p(labels.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return (masks, labels) if return_labels else masks

    def __call__(self, batch_image, **kw) -> torch.Tensor:
        masks = self.masks(batch_image)
        N, K = masks.shape[:2]
        segm, _ = self.ops.patch_segm_from_masks(masks.view(N * K, *masks.shape[2:]), [i * K for i in range(N + 1)], *self.grid_hw)
        return segm
