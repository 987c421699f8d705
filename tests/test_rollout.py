"""configs[3] closed loop (dynam3d_amd/rollout.py::run_closed_loop = VLN-TR:624-806 against the synthetic environment): the sentences
the policy GENERATES are parsed by `convert_text_to_action` and move the agents; finished episodes are popped from the 3D memory."""
import math

import numpy as np
import pytest
import torch

from dynam3d_amd import dist as DD
from dynam3d_amd.policy import ActionGrammarTokenizer, Dynam3D_VLN, synth_policy_weights
from dynam3d_amd.rollout import run_closed_loop
from dynam3d_amd.synthetic import ClosedLoopEpisodes


def test_closed_loop_environment_applies_the_trainers_actions():
    """HIGHTOLOW (VLN-TR:713-719): rotate counter-clockwise by `angle`, then move `distance` along the new heading; None = stop.
    Finished environments disappear from the batch (indices shift, like `envs.pause_at`)."""
    env = ClosedLoopEpisodes(3, seed=0, image_hw=32, depth_hw=32)
    p0, h0 = [p.copy() for p in env.pos], list(env.head)
    dones, infos = env.step([(math.radians(30), 0.5), None, (2 * math.pi - math.radians(15), 0.0)])
    assert dones == [False, True, False] and infos[1]["steps_taken"] == 1 and infos[1]["path_length"] == 0.0
    assert env.num_envs == 2 and env.ids == [0, 2]
    h = (h0[0] + math.radians(30)) % (2 * math.pi)
    assert abs(env.head[0] - h) < 1e-12 and np.allclose(env.pos[0], p0[0] + np.array([-0.5 * math.sin(h), 0.0, -0.5 * math.cos(h)]))
    assert abs(env.head[1] - (h0[2] - math.radians(15)) % (2 * math.pi)) < 1e-9 and np.allclose(env.pos[1], p0[2])
    fr = env.observe()
    assert fr.rgb.shape == (2, 32, 32, 3) and fr.patch_segm.shape == (2, 1, 24, 24) and len(fr.positions) == 2


def test_action_grammar_reaches_every_branch_of_convert_text_to_action():
    tok = ActionGrammarTokenizer(640, stop_mod=8)
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(400):
        ids = rng.integers(0, 640, size=int(rng.integers(1, 20))).tolist()
        t = tok.decode(ids)
        assert t == tok.decode(ids)
        a = Dynam3D_VLN.convert_text_to_action([t])[0]
        seen.add("stop" if a == -100 else ("turn+move" if a[1] > 0 else "turn"))
        if a != -100:
            assert 0.0 <= a[0] <= 2 * math.pi and a[1] in (0.0, 0.25, 0.5, 0.75, 1.0)
    assert seen == {"stop", "turn+move", "turn"}
    end = tok.SPECIAL["<|end|>"] % 640
    assert tok.decode([5, 9, end, 77, 3]) == tok.decode([5, 9])                 # tokens behind <|end|> do not count


def _check_loop(net, episodes, max_steps, new_tokens, seed):
    trace = []
    sums, n, env_steps, _ = run_closed_loop(net, episodes, max_steps, seed=seed, max_new_tokens=new_tokens, trace=trace)
    assert n == episodes and net.feature_fields.batch_size == 0                   # every episode ended (stop or max_steps) and was popped
    assert env_steps == sums["steps_taken"] == sum(len(t["texts"]) for t in trace)
    # replay: the poses the policy saw are exactly the ones its own earlier sentences produced
    env = ClosedLoopEpisodes(episodes, seed=seed)
    for t in trace:
        assert [p.tolist() for p in env.pos] == t["positions"] and list(env.head) == t["headings"]
        acts = Dynam3D_VLN.convert_text_to_action(t["texts"])
        for b, a in enumerate(acts):
            want = None if (a == -100 or t["step"] == max_steps - 1 or (a[0] == 0.0 and a[1] == 0.0)) else a
            assert t["actions"][b] == want
        env.step(t["actions"])
    return sums, trace


def test_closed_loop_rollout_cpu():
    from tests.cpu_ops import CpuOps
    from tests.test_policy_cpu import SMALL
    net = Dynam3D_VLN(SMALL, synth_policy_weights(SMALL, 0), device="cpu", batch_size=3, ops=CpuOps(), max_steps=6,
                      tokenizer=ActionGrammarTokenizer(SMALL.llm.vocab, stop_mod=5))
    sums, trace = _check_loop(net, 3, 5, new_tokens=4, seed=2)
    res = DD.gather_metrics(sums, 3)
    assert res["episodes"] == 3.0 and 1.0 <= res["steps_taken"] <= 5.0
    # the action history the next prompt is built from holds the generated sentences (VLN-POL:466-468)
    assert all(isinstance(s, str) and s for t in trace for s in t["texts"])


@pytest.mark.gpu
def test_closed_loop_rollout_full_model_8x50():
    """configs[3] on one rank at full size: 8 episodes x up to 50 steps, prefill + 20-token KV-cache generation per step, strict HIP."""
    import time
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.policy import PolicyConfig
    cfg = PolicyConfig()
    D.enable_hip_kernels(["all"])
    D.strict(True)
    D.reset_counts()
    try:
        net = Dynam3D_VLN(cfg, synth_policy_weights(cfg, 0, device="cuda"), device="cuda", batch_size=8, max_steps=51,
                          tokenizer=ActionGrammarTokenizer(cfg.llm.vocab, stop_mod=40))
        torch.cuda.synchronize()
        t0 = time.time()
        sums, trace = _check_loop(net, 8, 50, new_tokens=20, seed=7)
        torch.cuda.synchronize()
        dt = time.time() - t0
    finally:
        D.strict(True)                                              # (the default)
    kinds = {("stop" if a is None else "move") for t in trace for a in t["actions"]}
    alive = [len(t["texts"]) for t in trace] + [0]
    lens = sorted(k + 1 for k in range(len(trace)) for _ in range(alive[k] - alive[k + 1]))          # episode k ended after its step k
    print(f"closed loop 8 x 50: {int(sums['steps_taken'])} env steps in {dt:.2f} s = {sums['steps_taken'] / dt:.1f} env-steps/s incl. 20-token generation; "
          f"episode lengths {lens}; actions seen {sorted(kinds)}; mean path {sums['path_length'] / 8:.2f} m")
    assert sums["steps_taken"] >= 8 and not D.counts()["fallback"]
