"""a6: `Feature_Fields.get_patch_segm` (VLN-FF:399-430) around a pluggable mask generator.

The reference runs FastSAM (YOLOv8-seg, `everything_prompt`) per image and then turns its masks into the 24x24 dense label
map the memory update consumes.  The network itself is outside the hot path; everything AFTER it is here,
on the device, in one launch per batch (`d3d_patch_segm_from_masks`): 'last mask wins' label image, nearest resize to the
patch grid, dense relabel in `torch.unique` order; a failing / empty segmentation gives an all-zero map exactly like the
reference's `except` branch (VLN-FF:424-426)."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch


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
