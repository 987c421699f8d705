"""Synthetic posed RGB-D episode driver (replaces the Habitat simulator side, which is out of scope).

Spec: SURVEY.md section 8(d) / BASELINE.md section 3 -- seed 0; RGB uint8 (B,224,224,3) uniform;
depth fp32 (B,224,224,1) ~ U(0.05,0.5) with 1 % exact zeros (x10 -> metres); pose x,z ~ U(-2,2),
y = 0, heading ~ U(0,2pi), advancing 0.25 m along the heading and turning U(-30,30) deg per step;
`patch_segm` = 4x4 blocks of 6x6 patches with a seeded label permutation (16 segments), already in
the dense-relabelled form `get_patch_segm` produces (VLN-FF:416-420).

Everything is generated with numpy's PCG64 on the host so the CPU oracle and the GPU path see
bit-identical inputs on any machine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import numpy as np

INSTRUCTION_64 = (
    "walk forward past the kitchen counter and turn left at the dining table then continue down the "
    "hallway until you reach the second door on your right enter the bedroom and walk around the bed "
    "to the window then turn right and stop next to the wooden dresser beside the small reading lamp "
    "and wait there facing the large mirror on the far wall of the room near the open closet door now"
)


@dataclass
class Frame:
    rgb: np.ndarray          # (B,h,w,3) uint8
    depth: np.ndarray        # (B,H,W,1) float32 in [0,1]
    positions: List[np.ndarray]  # habitat (x,y,z) per env, float64 like habitat's agent state
    headings: List[float]
    patch_segm: np.ndarray   # (B,1,24,24) int64 dense labels 0..n-1


class SyntheticEpisodes:
    def __init__(self, batch_size: int = 8, seed: int = 0, image_hw: int = 224, depth_hw: int = 224,
                 n_blocks: int = 4, grid: int = 24, stationary: bool = False, wall: float | None = None, wall_steps=None):
        """wall: a flat wall `wall` metres in front of the camera instead of random depth -- on every step, or only on the steps listed
        in `wall_steps` (long trajectories: random-depth steps grow the memory, a wall step re-observes and deletes a large part of what
        the frustum holds, so ids are recycled and zone snapshots go stale many times over)."""
        self.B = batch_size
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.image_hw, self.depth_hw, self.grid, self.n_blocks = image_hw, depth_hw, grid, n_blocks
        self.stationary, self.wall = stationary, wall
        self.wall_steps = None if wall_steps is None else set(int(t) for t in wall_steps)
        self.pos = [np.array([self.rng.uniform(-2, 2), 0.0, self.rng.uniform(-2, 2)], np.float64) for _ in range(batch_size)]
        self.head = [float(self.rng.uniform(0, 2 * math.pi)) for _ in range(batch_size)]
        self.t = 0

    def _segm(self) -> np.ndarray:
        g, nb = self.grid, self.n_blocks
        blk = g // nb
        out = np.zeros((self.B, 1, g, g), np.int64)
        for b in range(self.B):
            perm = self.rng.permutation(nb * nb)
            lab = perm.reshape(nb, nb)
            out[b, 0] = np.kron(lab, np.ones((blk, blk), np.int64))
        return out

    def next(self) -> Frame:
        B, rng = self.B, self.rng
        rgb = rng.integers(0, 256, size=(B, self.image_hw, self.image_hw, 3), dtype=np.uint8)
        if self.wall is None or (self.wall_steps is not None and self.t not in self.wall_steps):
            depth = rng.uniform(0.05, 0.5, size=(B, self.depth_hw, self.depth_hw, 1)).astype(np.float32)
        else:  # flat wall `wall` metres away (x10 scaling) -> every stored patch is re-observed
            depth = np.full((B, self.depth_hw, self.depth_hw, 1), self.wall / 10.0, np.float32)
        zeros = rng.random(size=depth.shape) < 0.01
        depth[zeros] = 0.0
        fr = Frame(rgb=rgb, depth=depth, positions=[p.copy() for p in self.pos], headings=list(self.head),
                   patch_segm=self._segm())
        if not self.stationary:
            for b in range(B):
                h = self.head[b]
                # habitat: heading CCW about +y, forward = -z rotated by heading
                self.pos[b][0] += -0.25 * math.sin(h)
                self.pos[b][2] += -0.25 * math.cos(h)
                self.head[b] = float((h + math.radians(rng.uniform(-30, 30))) % (2 * math.pi))
        self.t += 1
        return fr


def nearest_resize_indices(src: int, dst: int) -> np.ndarray:
    """cv2.resize(INTER_NEAREST) source index rule (VLN-POL:339): floor(dst_i * src/dst), clamped."""
    idx = np.floor(np.arange(dst) * (src / dst)).astype(np.int64)
    return np.minimum(idx, src - 1)


class ClosedLoopEpisodes:
    """Synthetic stand-in for the Habitat side of `RLTrainer.rollout` (VLN-TR:624-633, 702-719): every environment has a pose that the
    POLICY'S OWN ACTIONS move.  `observe()` returns the frame at the current poses (`get_agent_info`: position, heading + the RGB-D
    sensors -- seeded images, the same statistics as `SyntheticEpisodes`); `step(actions)` applies the trainer's HIGHTOLOW action
    (VLN-TR:713-719): rotate by `angle` (radians, counter-clockwise) and move `distance` metres forward, or ends the episode for a
    stop action.  A synthetic goal 4-7 m from the start gives the reference's metrics something to measure (VLN-TR:735-748:
    distance_to_goal, success = final distance <= 3 m, oracle_success, path_length, spl)."""

    def __init__(self, n_envs: int, seed: int = 0, image_hw: int = 224, depth_hw: int = 224, n_blocks: int = 4, grid: int = 24):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.image_hw, self.depth_hw, self.grid, self.n_blocks = image_hw, depth_hw, grid, n_blocks
        self.ids = list(range(n_envs))                                  # episode ids of the environments still running
        self.pos = [np.array([self.rng.uniform(-2, 2), 0.0, self.rng.uniform(-2, 2)], np.float64) for _ in range(n_envs)]
        self.head = [float(self.rng.uniform(0, 2 * math.pi)) for _ in range(n_envs)]
        ga, gd = self.rng.uniform(0, 2 * math.pi, n_envs), self.rng.uniform(4.0, 7.0, n_envs)
        self.goal = [self.pos[i] + np.array([gd[i] * math.cos(ga[i]), 0.0, gd[i] * math.sin(ga[i])]) for i in range(n_envs)]
        self.path = [[p.copy()] for p in self.pos]
        self.steps = [0] * n_envs

    @property
    def num_envs(self) -> int:
        return len(self.ids)

    def observe(self) -> Frame:
        B, rng = self.num_envs, self.rng
        rgb = rng.integers(0, 256, size=(B, self.image_hw, self.image_hw, 3), dtype=np.uint8)
        depth = rng.uniform(0.05, 0.5, size=(B, self.depth_hw, self.depth_hw, 1)).astype(np.float32)
        depth[rng.random(size=depth.shape) < 0.01] = 0.0
        blk = self.grid // self.n_blocks
        segm = np.zeros((B, 1, self.grid, self.grid), np.int64)
        for b in range(B):
            segm[b, 0] = np.kron(rng.permutation(self.n_blocks ** 2).reshape(self.n_blocks, self.n_blocks), np.ones((blk, blk), np.int64))
        return Frame(rgb=rgb, depth=depth, positions=[p.copy() for p in self.pos], headings=list(self.head), patch_segm=segm)

    def step(self, actions):
        """actions[b] = None (stop, env action 0) or (angle rad CCW, distance m) (env action 4, VLN-TR:713-719).  Returns (dones, infos);
        finished environments are REMOVED (the trainer's `envs.pause_at(i)`, VLN-TR:779-781), so indices shift like the trainer's."""
        dones, infos = [], []
        for b, a in enumerate(actions):
            self.steps[b] += 1
            if a is None:
                d = [float(np.linalg.norm(p - self.goal[b])) for p in self.path[b]]
                pl = float(sum(np.linalg.norm(q - p) for p, q in zip(self.path[b][:-1], self.path[b][1:])))
                ok = 1.0 if d[-1] <= 3.0 else 0.0
                infos.append(dict(episode_id=self.ids[b], steps_taken=self.steps[b], distance_to_goal=d[-1], success=ok,
                                  oracle_success=1.0 if min(d) <= 3.0 else 0.0, path_length=pl, collisions=0.0,
                                  spl=ok * d[0] / max(d[0], pl, 1e-9), ndtw=0.0, sdtw=0.0))
                dones.append(True)
                continue
            angle, dist = a
            self.head[b] = float((self.head[b] + angle) % (2 * math.pi))
            # habitat: heading counter-clockwise about +y, forward = -z rotated by the heading (the convention of SyntheticEpisodes)
            self.pos[b] = self.pos[b] + np.array([-dist * math.sin(self.head[b]), 0.0, -dist * math.cos(self.head[b])])
            self.path[b].append(self.pos[b].copy())
            dones.append(False)
            infos.append(None)
        for b in reversed(range(len(actions))):
            if dones[b]:
                for lst in (self.ids, self.pos, self.head, self.goal, self.path, self.steps):
                    lst.pop(b)
        return dones, infos


def bench_point_grid(step: int, batch_size: int = 8, patches: int = 576, dim: int = 768, seed: int = 2206) -> np.ndarray:
    """Seeded unit-norm CLIP-like grid features of memory-advance step `step`, float16 (B, patches, dim): the numbers both legs of the golden
    parity point feed to the 3D memory (tests/golden/gen_golden_bench_point.py on the CPU oracle, bench.py / the GPU test on the product),
    fp16-representable so that the product's fp16 feature store and the oracle's float32 one hold identical values."""
    rng = np.random.default_rng(seed + 1000 * step)
    g = rng.standard_normal((batch_size, patches, dim)).astype(np.float32)
    g /= np.linalg.norm(g, axis=-1, keepdims=True)
    return g.astype(np.float16)
