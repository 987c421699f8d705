"""Drop-in for the reference's un-vendored CUDA dependency `torch_kdtree` (environment.yml:293):

    from torch_kdtree import build_kd_tree                      (VLN-FF:7, PRE-FF)
    tree = build_kd_tree(points)                                 (VLN-FF:246; PRE-FF:303, 364)
    dist2, idx = tree.query(queries, nr_nns_searches=k)          (VLN-FF:606-610; PRE-FF:540, 562, 584, 978)

Same call surface, backed by `d3d_knn` (brute force on the GPU: the point sets on this path are 10^2..10^4, where a
flat LDS-tiled scan beats building a tree every frame).  Returns squared distances ascending with a DEFINED tie
order (lowest index first); torch_kdtree's own tie order is traversal-dependent and unpinned."""
from __future__ import annotations

import numpy as np
import torch

from .ops import HipOps

_K_CHOICES = (1, 2, 4, 8)


class KDTree:
    def __init__(self, points, ops: HipOps | None = None, device="cuda"):
        if isinstance(points, np.ndarray):
            points = torch.from_numpy(points)
        # the tree owns a COPY of the points (later in-place edits of the caller's tensor are not seen)
        self.points = points.detach().to(device, torch.float32).reshape(-1, 3).contiguous().clone()
        self.ops = ops or HipOps()

    def query(self, queries, nr_nns_searches: int = 1):
        if isinstance(queries, np.ndarray):
            queries = torch.from_numpy(queries)
        q = queries.detach().to(self.points.device, torch.float32).reshape(-1, 3).contiguous()
        k = int(nr_nns_searches)
        if k > 8:
            raise ValueError("d3d_knn supports k <= 8 (the reference uses k <= 4)")
        m, n = q.shape[0], self.points.shape[0]
        if k == 0 or m == 0:
            return (torch.zeros((m, 0), dtype=torch.float32, device=q.device), torch.zeros((m, 0), dtype=torch.int64, device=q.device))
        kmax = next(x for x in _K_CHOICES if x >= k)
        cnt = lambda v: torch.tensor([v], dtype=torch.int32, device=q.device)
        d2, idx = self.ops.knn(self.points, 0, cnt(n), q, 0, cnt(m), cnt(min(k, n)), 1, m, kmax)
        return d2[0, :, :k].contiguous(), idx[0, :, :k].long().contiguous()


def build_kd_tree(points, device=None, **_ignored) -> KDTree:
    dev = device or (points.device if isinstance(points, torch.Tensor) and points.is_cuda else "cuda")
    return KDTree(points, device=dev)
