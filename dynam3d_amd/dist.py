"""Multi-GPU mode: independent episodes shard one process per GPU (the reference shards by rank too: seed+rank
VLN-TR:141, scene splits env_utils.py:87-107); nothing is exchanged on the data path.  The only collective is the
end-of-evaluation metric gather, issued as ONE all_gather of a float32[10] vector (9 metric sums + episode count)
instead of the reference's barrier + reduce + 9 scalar all_gathers (VLN-TR:389-408, 735-746).  Backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

METRIC_KEYS = ("steps_taken", "distance_to_goal", "success", "oracle_success", "path_length", "collisions", "spl", "ndtw", "sdtw")


def init_from_env(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run.  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("D3D_SHARE_DEVICE0") == "1":
        local = 0          # test hook: several ranks on ONE GPU (with D3D_DIST_BACKEND=gloo; RCCL refuses duplicate devices)
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("D3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, local, world


def gather_metrics(sums: Dict[str, float], n_episodes: int, device="cpu") -> Dict[str, float]:
    """One collective: all_gather of [9 metric sums, count] -> global means (every rank gets the result)."""
    v = torch.tensor([float(sums.get(k, 0.0)) for k in METRIC_KEYS] + [float(n_episodes)], dtype=torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(out, v)
        tot = torch.stack(out).sum(0)
    else:
        tot = v
    n = max(float(tot[-1]), 1.0)
    res = {k: float(tot[i]) / n for i, k in enumerate(METRIC_KEYS)}
    res["episodes"] = float(tot[-1])
    return res


def max_over_ranks(x: float, device="cpu") -> float:
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])
    return x


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])      # pin the collective to this rank's GPU
        else:
            dist.barrier()


def shutdown():
    """Tear the process group down before interpreter exit (RCCL communicators are otherwise destroyed by atexit handlers in an
    unspecified order, which newer PyTorch versions warn about)."""
    if dist.is_initialized():
        dist.destroy_process_group()
