"""Synthetic-episode rollout driver (BASELINE config 4): the shape of `RLTrainer.rollout` / `_eval_checkpoint`
(VLN-TR:332-431, 564-822) without Habitat -- every rank advances its own batch of independent episodes with the
drop-in policy, finished episodes are popped from the 3D memory exactly like the trainer does (VLN-TR:778-784),
and the per-rank metric sums are merged with ONE all_gather (dist.gather_metrics).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m dynam3d_amd.rollout --episodes-per-rank 8 --max-steps 50 [--logits-only]

Two loops:
  * `run_closed_loop` (default): the trainer's loop body VLN-TR:624-806 -- `net(batch, instructions, positions, headings,
    depth_scale=(0.,10.), gt_text=None, delete_old_features=True, num_of_views=1, is_train=False) -> List[str]` (prefill + 20-token
    greedy generation with the KV cache), `convert_text_to_action`, stop / HIGHTOLOW env actions, the pose update inside the synthetic
    environment (`synthetic.ClosedLoopEpisodes`), metrics of finished episodes, `feature_fields.pop(i)`;
  * `run_rollout` (`--logits-only`): the headline metric's step (`forward_logits`, no generation) on open-loop episodes."""
from __future__ import annotations

import argparse
import json
import sys
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


def run_closed_loop(net: Dynam3D_VLN, episodes: int, max_steps: int, seed: int, max_new_tokens: int = 20, instruction: str = INSTRUCTION_64,
                    image_hw: int = 224, trace: list = None):
    """`RLTrainer.rollout(mode='eval')` (VLN-TR:564-806) against `ClosedLoopEpisodes`: the text the policy generates IS what moves the
    agents.  Returns (metric sums, finished episodes, env steps taken, generated sentences of the last step)."""
    from .synthetic import ClosedLoopEpisodes
    ff = net.feature_fields
    envs = ClosedLoopEpisodes(episodes, seed=seed, image_hw=image_hw, depth_hw=image_hw)
    instructions = [instruction] * episodes
    ff.reset(episodes)                                                              # VLN-TR:621
    ff.initialize_camera_setting(hfov=90., vfov=90.)                                # VLN-TR:622
    sums = {k: 0.0 for k in DD.METRIC_KEYS}
    done_n, env_steps, texts = 0, 0, []
    for stepk in range(max_steps):
        env_steps += envs.num_envs                                                  # VLN-TR:625
        fr = envs.observe()                                                         # get_agent_info + sensors (VLN-TR:627-633)
        batch = {"rgb": torch.from_numpy(fr.rgb).to(net.device), "depth": torch.from_numpy(fr.depth).to(net.device)}
        positions, headings = [p.tolist() for p in fr.positions], list(fr.headings)
        B = envs.num_envs
        stop_actions = [False] * B
        target_angles, target_distances = [None] * B, [None] * B
        texts = net(batch, instructions, positions, headings, depth_scale=(0., 10.), gt_text=None, delete_old_features=True, num_of_views=1,
                    is_train=False, patch_segm=fr.patch_segm, max_new_tokens=max_new_tokens)           # VLN-TR:671
        predicted = net.convert_text_to_action(texts)                               # VLN-TR:693
        for b in range(B):
            if predicted[b] == -100:
                stop_actions[b] = True
            else:
                target_angles[b], target_distances[b] = predicted[b]
        actions = []
        for b in range(B):                                                          # VLN-TR:702-719
            if stop_actions[b] or stepk == max_steps - 1 or (target_angles[b] == 0. and target_distances[b] == 0.):
                actions.append(None)
            else:
                actions.append((target_angles[b], target_distances[b]))
        if trace is not None:
            trace.append(dict(step=stepk, texts=list(texts), actions=list(actions), positions=positions, headings=headings))
        dones, infos = envs.step(actions)
        for i in reversed(range(B)):                                                # VLN-TR:735-748, 778-784
            if dones[i]:
                for k in DD.METRIC_KEYS:
                    sums[k] += float(infos[i][k])
                done_n += 1
                ff.pop(i)
                instructions.pop(i)
        if envs.num_envs == 0:                                                      # VLN-TR:802-804
            ff.delete_feature_fields()
            break
    ff.check_numerics()           # the float32 token-builder GEMMs' status word is otherwise only looked at by the NEXT update: end of the loop
    return sums, done_n, env_steps, texts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; without a launcher the command re-launches itself under torch.distributed.run (dist.launch_guard)")
    ap.add_argument("--episodes-per-rank", type=int, default=8)
    ap.add_argument("--max-steps", type=int, default=50)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stop-mod", type=int, default=7, help="an episode stops when its argmax token is divisible by this (huge = never: full-length episodes)")
    ap.add_argument("--per-rank", action="store_true", help="every rank prints its own wall / CPU seconds (host-side load when ranks share a host)")
    ap.add_argument("--logits-only", action="store_true", help="open-loop episodes, forward_logits only (the headline metric's step); default: the closed loop with generation")
    ap.add_argument("--new-tokens", type=int, default=20, help="max_new_tokens of the generation (VLN-POL:463: 20)")
    ap.add_argument("--grammar-stop-mod", type=int, default=24, help="ActionGrammarTokenizer: ~2 of this many sentences end an episode")
    a = ap.parse_args()
    if a.gpus is not None:
        DD.launch_guard(a.gpus, "-m dynam3d_amd.rollout", sys.argv[1:])
    rank, local, world = DD.init_from_env()
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    D.enable_hip_kernels(["all"])
    cfg = PolicyConfig()
    from .policy import ActionGrammarTokenizer
    net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, a.seed, device=dev), device=dev, batch_size=a.episodes_per_rank, max_steps=a.max_steps + 1,
                      tokenizer=None if a.logits_only else ActionGrammarTokenizer(cfg.llm.vocab, stop_mod=a.grammar_stop_mod))
    DD.barrier()
    t0, c0 = time.time(), time.process_time()
    if a.logits_only:
        sums, n = run_rollout(net, a.episodes_per_rank, a.max_steps, seed=a.seed + 1000 * rank, stop_token_mod=a.stop_mod)   # seed + rank (VLN-TR:141)
    else:
        sums, n, _, _ = run_closed_loop(net, a.episodes_per_rank, a.max_steps, seed=a.seed + 1000 * rank, max_new_tokens=a.new_tokens)
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
        print(json.dumps(dict(world=world, mode="logits-only (open loop)" if a.logits_only else f"closed loop, {a.new_tokens}-token generation", episodes=res["episodes"], seconds=round(dt, 2),
                              env_steps_per_s=round(res["steps_taken"] * res["episodes"] / dt, 1), metrics=res)))
    DD.shutdown()


if __name__ == "__main__":
    main()
