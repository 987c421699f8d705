"""Multi-GPU mode: independent episodes shard one process per GPU (the reference shards by rank too: seed+rank
VLN-TR:141, scene splits env_utils.py:87-107); nothing is exchanged on the data path.  The only collective is the
end-of-evaluation metric gather, issued as ONE all_gather of a float32[10] vector (9 metric sums + episode count)
instead of the reference's barrier + reduce + 9 scalar all_gathers (VLN-TR:389-408, 735-746).  Backend "nccl" is
RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.

Pre-training step (the reference wraps the net in DDP, PRE-TR:356-360, 479-526, 2237-2271): the same three
collectives, explicit -- `broadcast_int` (rank 0 picks the dataset loop of the iteration), `any_nan_vote` (the loss is all-reduced
and the step skipped everywhere if it is NaN anywhere) and `all_reduce_gradients`: gradients are flattened into a few large
buckets (one collective each; per-message latency dominates a ~160 MB gradient set over xGMI's point-to-point links, so few
large messages, and the reduction is averaged like DDP's)."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

def _active() -> bool:
    """Collectives are issued when a process group exists and it has more than one rank -- or, with `D3D_DIST_FORCE=1`, even at
    world size 1: the single-rank collective is an identity, but it runs the whole backend path (communicator creation, device
    buffers, stream ordering), which is how the one-GPU test box exercises RCCL (tests/test_gpu_rccl.py)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("D3D_DIST_FORCE") == "1")


METRIC_KEYS = ("steps_taken", "distance_to_goal", "success", "oracle_success", "path_length", "collisions", "spl", "ndtw", "sdtw")


def init_from_env(backend: Optional[str] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run.  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("D3D_SHARE_DEVICE0") == "1":
        local = 0          # test hook: several ranks on ONE GPU (with D3D_DIST_BACKEND=gloo; RCCL refuses duplicate devices)
    if (world > 1 or os.environ.get("D3D_DIST_FORCE") == "1") and not dist.is_initialized():
        backend = backend or os.environ.get("D3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://")
    return rank, local, world


def free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_guard(gpus: int, script: str, argv):
    """`--gpus N` must mean N ranks on N distinct GPUs, or no number at all (shared by bench.py and rollout.py).
      * N > 1 without a launcher (WORLD_SIZE unset): re-launch `script argv` under `torch.distributed.run` with N ranks -- if the
        box has N GPUs; otherwise exit non-zero (a plain `--gpus 8` on one GPU must not print 8x one GPU's rate);
      * under a launcher whose WORLD_SIZE differs from --gpus: exit non-zero.
    Runs before anything is allocated."""
    import subprocess
    import sys
    name = os.path.basename(script)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != gpus:
            sys.exit(f"{name}: --gpus {gpus} but the launcher started WORLD_SIZE={world_env} ranks; refusing to report a number for a job that is not the one named")
        return
    if gpus == 1:
        return
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("D3D_SHARE_DEVICE0") == "1":
        n_dev = max(n_dev, gpus if n_dev >= 1 else 0)          # test hook: several ranks on cuda:0 (see init_from_env)
    if n_dev < gpus:
        sys.exit(f"{name}: --gpus {gpus} needs {gpus} GPUs, this machine shows {n_dev}; refusing to report a {gpus}-GPU number "
                 f"(launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node {gpus} --master-addr 127.0.0.1 --master-port P {name} --gpus {gpus} ...)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + ([script] if not script.startswith("-m ") else ["-m", script[3:]]) + list(argv)
    print("%s: --gpus %d without a launcher -> %s" % (name, gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd))


def gather_metrics(sums: Dict[str, float], n_episodes: int, device="cpu") -> Dict[str, float]:
    """One collective: all_gather of [9 metric sums, count] -> global means (every rank gets the result)."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"                      # gloo gathers host tensors (RCCL / "nccl" takes the device tensor)
    v = torch.tensor([float(sums.get(k, 0.0)) for k in METRIC_KEYS] + [float(n_episodes)], dtype=torch.float32, device=device)
    if _active():
        out = [torch.empty_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(out, v)
        tot = torch.stack(out).sum(0)
    else:
        tot = v
    n = max(float(tot[-1]), 1.0)
    res = {k: float(tot[i]) / n for i, k in enumerate(METRIC_KEYS)}
    res["episodes"] = float(tot[-1])
    return res


def gather_objects(obj):
    """Small per-rank records (device identity, own timing) of every rank, on every rank, in rank order."""
    if _active():
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def max_over_ranks(x: float, device="cpu") -> float:
    if _active():
        if dist.get_backend() == "gloo":
            device = "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])
    return x


def barrier():
    if _active():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])      # pin the collective to this rank's GPU
        else:
            dist.barrier()


def broadcast_int(value: int, src: int = 0, device="cpu") -> int:
    """PRE-TR:2239-2244: every rank runs the dataset loop rank `src` drew."""
    if _active():
        if dist.get_backend() == "gloo":
            device = "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=device)
        dist.broadcast(t, src=src)
        return int(t[0])
    return int(value)


def any_nan_vote(loss: torch.Tensor) -> bool:
    """PRE-TR:505-509: SUM all-reduce of the loss value; True = some rank produced NaN -> every rank skips backward / the step."""
    v = loss.detach().clone().float().reshape(1)
    if _active():
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return bool(torch.isnan(v).any())


def all_reduce_gradients(params, bucket_bytes: int = 64 << 20, average: bool = True, nan_to_zero: bool = True, assume_uniform: bool = False):
    """Data-parallel gradient reduction (DDP's role, PRE-TR:356-360, 512): the gradients of `params` (those that have one) are
    packed into float32 buckets of ~`bucket_bytes`, each bucket is ONE all_reduce (SUM), divided by the world size and scattered
    back into `.grad` in place.  `nan_to_zero` applies the reference's per-parameter NaN scrub (PRE-TR:513-515) after the
    reduction.  Parameters whose grad is None on this rank contribute zeros so that every rank issues the same collectives
    (the reference's `* 0.` terms exist for the same reason, PRE-FF:1340) -- but a parameter that has NO gradient on ANY rank
    keeps `.grad = None` (one extra all_reduce of a has-grad mask): the optimizer then skips it, as it does under the
    reference's DDP; materialising zeros would let AdamW decay and update moments of parameters the step did not touch.
    `assume_uniform`: the caller guarantees that every rank has gradients for the same parameters (e.g. `pretrain_step`'s dummy
    term); the mask exchange and its host synchronisation are skipped.
    Returns the number of collectives issued (the mask exchange included)."""
    params = [p for p in params if p.requires_grad]
    world = dist.get_world_size() if dist.is_initialized() else 1
    has = None
    n_coll = 0
    active = _active()
    if params and active and not assume_uniform:
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=params[0].device)
        dist.all_reduce(has, op=dist.ReduceOp.SUM)
        n_coll += 1
        has = (has > 0).tolist()
    elif params:
        has = [p.grad is not None for p in params]
    if has is not None:
        params = [p for p, h in zip(params, has) if h]
    if not params:
        return n_coll
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, n_coll
        if not bucket:
            return
        dev = bucket[0].device
        flat = torch.empty(sum(p.numel() for p in bucket), dtype=torch.float32, device=dev)      # ONE float32 staging buffer per bucket
        o = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                flat[o:o + n].zero_()
            else:
                flat[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        if active:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            n_coll += 1
            if average:
                flat /= world
        if nan_to_zero:
            flat = torch.nan_to_num(flat, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        o = 0
        for p in bucket:
            n = p.numel()
            g = flat[o:o + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
    return n_coll


def shutdown():
    """Tear the process group down before interpreter exit (RCCL communicators are otherwise destroyed by atexit handlers in an
    unspecified order, which newer PyTorch versions warn about)."""
    if dist.is_initialized():
        dist.destroy_process_group()
