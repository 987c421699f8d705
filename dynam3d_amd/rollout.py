"""Synthetic-episode rollout driver (BASELINE config 4): the shape of `RLTrainer.rollout` / `_eval_checkpoint`
(VLN-TR:332-431, 564-822) without Habitat -- every rank advances its own batch of independent episodes with the
drop-in policy, finished episodes are popped from the 3D memory exactly like the trainer does (VLN-TR:778-784),
and the per-rank metric sums are merged with ONE all_gather (dist.gather_metrics).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m dynam3d_amd.rollout --episodes-per-rank 8 --max-steps 50
"""
from __future__ import annotations

import argparse
import json
import time

import numpy as np
import torch

from . import dense_ops as D
from . import dist as DD
from .policy import Dynam3D_VLN, PolicyConfig, synth_policy_weights
from .synthetic import INSTRUCTION_64, SyntheticEpisodes


def run_rollout(net: Dynam3D_VLN, episodes: int, max_steps: int, seed: int, stop_token_mod: int = 7):
    """Runs `episodes` concurrent synthetic episodes to completion.  An episode 'stops' when the argmax token of its
    action logits is divisible by `stop_token_mod` (stand-in for the text 'stop.'), or at max_steps.  Returns metric sums."""
    ff = net.feature_fields
    ff.reset(episodes)
    ff.initialize_camera_setting(90.0, 90.0)
    ep = SyntheticEpisodes(episodes, seed=seed)
    alive = list(range(episodes))
    steps_taken = np.zeros(episodes)
    path_len = np.zeros(episodes)
    sums = {k: 0.0 for k in DD.METRIC_KEYS}
    done = 0
    for t in range(max_steps):
        fr = ep.next()
        idx = np.asarray(alive)
        obs = {"rgb": torch.from_numpy(fr.rgb[idx]).to(net.device), "depth": torch.from_numpy(fr.depth[idx]).to(net.device)}
        pos = [fr.positions[i].tolist() for i in alive]
        hd = [fr.headings[i] for i in alive]
        logits = net.forward_logits(obs, [INSTRUCTION_64] * len(alive), pos, hd, patch_segm=fr.patch_segm[idx])
        tok = logits.argmax(-1).cpu().numpy()
        steps_taken[idx] += 1
        path_len[idx] += 0.25
        stop = [(int(tok[j]) % stop_token_mod == 0) or t == max_steps - 1 for j in range(len(alive))]
        for j in reversed(range(len(alive))):                      # pop finished episodes (VLN-TR:778-784)
            if stop[j]:
                e = alive.pop(j)
                ff.pop(j)
                done += 1
                sums["steps_taken"] += steps_taken[e]
                sums["path_length"] += path_len[e]
                sums["success"] += float(steps_taken[e] < max_steps)
                sums["spl"] += float(steps_taken[e] < max_steps) / max(path_len[e], 0.25)
        if not alive:
            break
    return sums, done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes-per-rank", type=int, default=8)
    ap.add_argument("--max-steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stop-mod", type=int, default=7, help="an episode stops when its argmax token is divisible by this (huge = never: full-length episodes)")
    ap.add_argument("--per-rank", action="store_true", help="every rank prints its own wall / CPU seconds (host-side load when ranks share a host)")
    a = ap.parse_args()
    rank, local, world = DD.init_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    D.enable_hip_kernels(["all"])
    cfg = PolicyConfig()
    net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, a.seed, device=dev), device=dev, batch_size=a.episodes_per_rank, max_steps=a.max_steps + 1)
    DD.barrier()
    t0, c0 = time.time(), time.process_time()
    sums, n = run_rollout(net, a.episodes_per_rank, a.max_steps, seed=a.seed + 1000 * rank, stop_token_mod=a.stop_mod)   # seed + rank (VLN-TR:141)
    torch.cuda.synchronize()
    wall, cpu = time.time() - t0, time.process_time() - c0
    if a.per_rank:
        # host-side load of one rank: wall seconds and CPU seconds (user + system of this process, GPU waits included when the
        # runtime spins) per environment step -- what N ranks on one host have to share
        steps = float(sums["steps_taken"])
        print("RANK", json.dumps(dict(rank=rank, world=world, env_steps=steps, wall_s=round(wall, 3), cpu_s=round(cpu, 3),
                                      wall_ms_per_batch_step=round(wall / max(steps / a.episodes_per_rank, 1) * 1e3, 2),
                                      cpu_ms_per_batch_step=round(cpu / max(steps / a.episodes_per_rank, 1) * 1e3, 2))), flush=True)
    res = DD.gather_metrics(sums, n, device=dev)                                                # the ONE collective
    if rank == 0:
        dt = time.time() - t0
        print(json.dumps(dict(world=world, episodes=res["episodes"], seconds=round(dt, 2),
                              env_steps_per_s=round(res["steps_taken"] * res["episodes"] / dt, 1), metrics=res)))
    DD.shutdown()


if __name__ == "__main__":
    main()
