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
