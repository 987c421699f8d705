"""GPU: the multi-rank code of dynam3d_amd/dist.py on the REAL backend ("nccl" = RCCL on ROCm) at world size 1.

The box has one GPU, so no scaling is measured here; what is checked is that every collective the episode-parallel mode and the
pre-training step issue (VLN-TR:389-408, 735-746; PRE-TR:479-526, 2237-2271) runs through RCCL on device tensors: communicator
creation from the launcher's environment, `all_gather` of the float32[10] metric vector, `all_gather_object`, the float64 MAX
all-reduce of the timing, `barrier(device_ids=...)`, `broadcast`, the NaN vote and the bucketed gradient all-reduce on real
parameters.  `D3D_DIST_FORCE=1` makes dist.py issue the collectives although world == 1 (each is then an identity -- the values
are checked against that)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from dynam3d_amd import dist as D
rank, local, world = D.init_from_env()                     # backend chosen by dist.py: "nccl" on a GPU box
dev = f"cuda:{local}"
out = dict(backend=dist.get_backend(), world=dist.get_world_size(), initialized=dist.is_initialized(), active=D._active())
sums = {k: float(i + 1) for i, k in enumerate(D.METRIC_KEYS)}
out["metrics"] = D.gather_metrics(sums, n_episodes=4, device=dev)                     # all_gather of a DEVICE float32[10]
out["objects"] = D.gather_objects(dict(rank=rank, dev=dev))                           # all_gather_object
out["mx"] = D.max_over_ranks(12.5, device=dev)                                        # float64 MAX all-reduce on the device
D.barrier()                                                                           # barrier(device_ids=[...])
out["bcast"] = D.broadcast_int(7, src=0, device=dev)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3)).to(dev)
extra = torch.nn.Parameter(torch.ones(5, device=dev))                                 # no gradient: must keep grad None
x, y = torch.randn(8, 6, device=dev), torch.randn(8, 3, device=dev)
loss = ((net(x) - y) ** 2).mean()
out["nan_vote"] = D.any_nan_vote(loss)
out["nan_vote_nan"] = D.any_nan_vote(loss * float("nan"))
loss.backward()
before = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
out["n_coll"] = D.all_reduce_gradients(list(net.parameters()) + [extra], bucket_bytes=256)       # mask exchange + several buckets
after = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
out["grad_equal"] = bool(torch.equal(before, after))
out["extra_grad_none"] = extra.grad is None
big = [torch.nn.Parameter(torch.zeros(4 << 20, device=dev)) for _ in range(5)]        # 80 MB of gradients: 64 MB bucket + remainder
for i, p in enumerate(big):
    p.grad = torch.full_like(p, float(i + 1))
out["n_coll_big"] = D.all_reduce_gradients(big, assume_uniform=True)
out["big_ok"] = all(bool((p.grad == float(i + 1)).all()) for i, p in enumerate(big))
torch.cuda.synchronize()
D.shutdown()
print("RESULT", json.dumps(out))
'''


@pytest.mark.gpu
def test_every_collective_runs_on_rccl_at_world_1(tmp_path):
    from dynam3d_amd.dist import free_port
    w = tmp_path / "rccl_worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               D3D_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("D3D_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, str(w), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    o = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0][7:])
    assert o["backend"] == "nccl" and o["world"] == 1 and o["initialized"] and o["active"]
    assert o["metrics"]["episodes"] == 4.0 and abs(o["metrics"]["steps_taken"] - 0.25) < 1e-7 and abs(o["metrics"]["sdtw"] - 9 / 4) < 1e-6
    assert o["objects"] == [dict(rank=0, dev="cuda:0")]
    assert o["mx"] == 12.5 and o["bcast"] == 7
    assert o["nan_vote"] is False and o["nan_vote_nan"] is True
    assert o["n_coll"] >= 3 and o["grad_equal"] and o["extra_grad_none"]
    assert o["n_coll_big"] == 2 and o["big_ok"]


@pytest.mark.gpu
def test_rollout_self_spawns_under_the_launcher_and_refuses_a_world_it_does_not_have():
    """`python -m dynam3d_amd.rollout --gpus N`: N = 2 on a one-GPU box exits non-zero before allocating anything (dist.launch_guard,
    the same guard as bench.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "D3D_SHARE_DEVICE0")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has 2+ GPUs: the refusal branch cannot be reached")
    p = subprocess.run([sys.executable, "-m", "dynam3d_amd.rollout", "--gpus", "2", "--max-steps", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "needs 2 GPUs" in (p.stdout + p.stderr), p.stdout + p.stderr
