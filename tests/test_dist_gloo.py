"""CPU: the N>1 path (episode sharding + single metric all_gather) with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
from dynam3d_amd import dist as D
rank, local, world = D.init_from_env("gloo")
sums = {k: float((rank + 1) * (i + 1)) for i, k in enumerate(D.METRIC_KEYS)}
res = D.gather_metrics(sums, n_episodes=rank + 2)
mx = D.max_over_ranks(10.0 + rank)
D.barrier()
print("RESULT", json.dumps(dict(rank=rank, world=world, res=res, mx=mx)))
'''


def test_two_rank_metric_gather(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "w.py"
    w.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    import json
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=120)
        assert p.returncode == 0, o
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT")][0][7:]))
    for o in outs:
        assert o["world"] == 2 and o["mx"] == 11.0
        assert o["res"]["episodes"] == 5.0
        # sums: rank0 -> (i+1), rank1 -> 2(i+1); mean over 5 episodes
        assert abs(o["res"]["steps_taken"] - 3.0 / 5.0) < 1e-6 and abs(o["res"]["sdtw"] - 27.0 / 5.0) < 1e-6


TRAIN_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
from dynam3d_amd import dist as D
rank, local, world = D.init_from_env("gloo")
torch.manual_seed(0)                                                # identical initial parameters on every rank (DDP's broadcast)
net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
extra = torch.nn.Parameter(torch.ones(5))                           # never used on rank 1: its grad is None there
ds = D.broadcast_int(3 if rank == 0 else 99, src=0)                 # PRE-TR:2239-2244
g = torch.Generator().manual_seed(100 + rank)                       # different data per rank
x, y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
loss = ((net(x) - y) ** 2).mean() + (extra.sum() * 0.5 if rank == 0 else 0.0)
skip = D.any_nan_vote(loss)
loss.backward()
n = D.all_reduce_gradients(list(net.parameters()) + [extra], bucket_bytes=256)     # tiny buckets -> several collectives
nan_loss = loss * float("nan") if rank == 1 else loss
skip_nan = D.any_nan_vote(nan_loss)                                 # one rank NaN -> everyone skips (PRE-TR:505-509)
D.barrier()
flat = torch.cat([p.grad.reshape(-1) for p in list(net.parameters()) + [extra]])
print("RESULT", json.dumps(dict(rank=rank, ds=ds, skip=skip, skip_nan=skip_nan, n=n, grad=flat.tolist(), x=x.tolist(), y=y.tolist())))
"""


def test_two_rank_gradient_all_reduce_broadcast_and_nan_vote(tmp_path):
    """The pre-training step's collectives (PRE-TR:479-526, 2237-2271) over gloo, world 2: the averaged bucketed gradient equals the
    gradient of the mean loss over both ranks' batches computed in one process."""
    import json
    import torch
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "tw.py"
    w.write_text(TRAIN_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=180)
        assert p.returncode == 0, o
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT")][0][7:]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["ds"] for o in outs] == [3, 3] and not any(o["skip"] for o in outs) and all(o["skip_nan"] for o in outs)
    assert outs[0]["n"] == outs[1]["n"] and outs[0]["n"] >= 2                      # same collectives on both ranks, bucketed
    assert outs[0]["grad"] == outs[1]["grad"]                                        # identical after the reduction
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.GELU(), torch.nn.Linear(16, 3))
    extra = torch.nn.Parameter(torch.ones(5))
    total = 0.0
    for o in outs:
        x, y = torch.tensor(o["x"]), torch.tensor(o["y"])
        total = total + ((net(x) - y) ** 2).mean() + (extra.sum() * 0.5 if o["rank"] == 0 else 0.0)
    (total / 2).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in list(net.parameters()) + [extra]])
    assert torch.allclose(torch.tensor(outs[0]["grad"]), ref, atol=1e-6, rtol=1e-5)


def test_bench_refuses_a_multi_gpu_number_it_cannot_measure():
    """`python bench.py --gpus N` (N > 1) with no launcher on a machine without N GPUs, or under a launcher with a different
    WORLD_SIZE, exits non-zero BEFORE allocating anything and prints no result line (round 3 printed N x one GPU's rate)."""
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "D3D_SHARE_DEVICE0")}
    out = subprocess.run([sys.executable, bench, "--gpus", "8"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "refusing" in out.stderr and "{" not in out.stdout
    out = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr and "{" not in out.stdout


FORCED_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
from dynam3d_amd import dist as D
rank, local, world = D.init_from_env("gloo")
import torch.distributed as dist
sums = {k: float(i + 1) for i, k in enumerate(D.METRIC_KEYS)}
res = D.gather_metrics(sums, n_episodes=2)
p = torch.nn.Parameter(torch.ones(3)); p.grad = torch.full((3,), 2.0)
n = D.all_reduce_gradients([p])
print("RESULT", json.dumps(dict(init=dist.is_initialized(), active=D._active(), world=world, res=res, mx=D.max_over_ranks(3.0), b=D.broadcast_int(5),
                                objs=D.gather_objects(dict(r=rank)), n=n, g=p.grad.tolist(), vote=D.any_nan_vote(torch.tensor(float("nan"))))))
D.barrier(); D.shutdown()
"""


def test_forced_collectives_at_world_1(tmp_path):
    """`D3D_DIST_FORCE=1`: the process group is created and every collective is ISSUED at world size 1 (identity results) -- the
    hook tests/test_gpu_rccl.py uses to run the same code on RCCL on the one-GPU box; here over gloo."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "fw.py"
    w.write_text(FORCED_WORKER)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), D3D_DIST_FORCE="1")
    p = subprocess.run([sys.executable, str(w), ROOT], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    o = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0][7:])
    assert o["init"] and o["active"] and o["world"] == 1
    assert o["res"]["episodes"] == 2.0 and o["mx"] == 3.0 and o["b"] == 5 and o["objs"] == [dict(r=0)]
    assert o["n"] == 2 and o["g"] == [2.0, 2.0, 2.0] and o["vote"] is True
