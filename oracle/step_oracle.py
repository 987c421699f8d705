"""Whole-step CPU oracle: posed RGB-D -> 3D tokens -> prefix -> Phi-3 prefill -> logits at the last prompt
position, float32 (VLN-POL:329-363, 430-463 restated on top of the pinned pieces: oracle/geometry.py,
ff_oracle.py, towers_ref.py).  TEST INFRASTRUCTURE; also the `cpu_baseline` leg of bench.py ("port")."""
from __future__ import annotations

import time
from typing import Dict, List

import numpy as np
import torch

from . import geometry as G
from . import towers_ref as TR
from .ff_oracle import FeatureFieldsOracle


class StepOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], vit_cfg, llm_cfg, batch_size: int, tokenizer, depth_scale=(0.0, 10.0),
                 clip_dtype=torch.float32, llava_dtype=torch.float32):
        """clip_dtype / llava_dtype: evaluate the towers the way the reference does in those dtypes (towers_ref `lowp`)."""
        self.sd, self.vit, self.llm, self.tok = sd, vit_cfg, llm_cfg, tokenizer
        self.clip_lowp = None if clip_dtype == torch.float32 else clip_dtype
        self.lm_lowp = None if llava_dtype == torch.float32 else llava_dtype
        self.ff = FeatureFieldsOracle(sd, batch_size)
        self.history = [["none\n"] * 4 for _ in range(batch_size)]
        self.depth_scale = depth_scale
        self.timing = {}

    ff_threads = None      # torch intra-op threads for the 3D-memory stage: thousands of set-sized (17 x 768) products per step, which a
                           # many-core host runs several times SLOWER on all its cores than on 8-16 (fork-join cost per op); None = leave as is

    def _ff_threads(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            n = torch.get_num_threads()
            if self.ff_threads:
                torch.set_num_threads(int(self.ff_threads))
            try:
                yield
            finally:
                torch.set_num_threads(n)
        return cm()

    @torch.no_grad()
    def build_inputs(self, rgb: np.ndarray, depth: np.ndarray, instructions: List[str], positions, headings, patch_segm):
        B = rgb.shape[0]
        t0 = time.time()
        d24 = G.preprocess_depth(G.downsample_depth_nearest(depth), (0.0, 10.0)).reshape(B, 1, -1)         # VLN-POL:336-341 (F9 fixed): the
        px = TR.preprocess_rgb(rgb, self.vit.image)                                                         # default scale, not depth_scale
        _, grid = TR.clip_vit_forward(px, self.sd, self.vit.layers, self.vit.heads, self.vit.patch, lowp=self.clip_lowp)   # VLN-POL:344
        t1 = time.time()
        dfull = G.preprocess_depth(depth, self.depth_scale)[..., 0]
        with self._ff_threads():
            self.ff.delete_old_features_from_camera_frustum(dfull.reshape(B, 1, *dfull.shape[1:]), positions, headings)   # VLN-POL:351
            self.ff.update_feature_fields(d24, grid.numpy().reshape(B, 1, *grid.shape[1:]), patch_segm, positions, headings)  # VLN-POL:354
            env = self.ff.get_environment_features(positions, headings)
        info = self.ff.get_patch_3d_info(d24.reshape(B, -1))
        rx, ry, rz, dr, sc = (torch.from_numpy(np.ascontiguousarray(a)) for a in info)
        info6 = torch.cat([rx, ry, rz, torch.sin(dr), torch.cos(dr), sc], -1)
        cat = lambda xs, w: torch.from_numpy(np.concatenate(xs).reshape(-1, w).astype(np.float32))
        patch_pos, inst_tok, zone_tok = TR.prefix_tokens(info6, cat(env["batch_instance_fts"], 768), cat(env["batch_instance_relative_position"], 3),
                                                         cat(env["batch_zone_fts"], 768), cat(env["batch_zone_relative_position"], 3), self.sd,
                                                         lowp=self.lm_lowp)
        t2 = time.time()
        R = TR._rounder(self.lm_lowp)
        patch_tok = R(TR.llava_image_features(px, self.sd, self.vit.layers, self.vit.heads, self.vit.patch, lowp=self.lm_lowp) + patch_pos)   # VLN-POL:448-453
        t3 = time.time()
        emb_w = R(self.sd["language_model.model.embed_tokens.weight"].float())
        ni = [len(x) for x in env["batch_instance_fts"]]
        nz = [len(x) for x in env["batch_zone_fts"]]
        io, zo = np.concatenate([[0], np.cumsum(ni)]), np.concatenate([[0], np.cumsum(nz)])
        rows = []
        for b in range(B):
            n_vis = patch_tok.shape[1] + ni[b] + nz[b]
            text = ("<|user|>\n" + "<image>" * n_vis + "\nInstruction:\n" + instructions[b] + "\nHistory actions:\n" + "".join(self.history[b])
                    + "<|end|>\n<|assistant|>\nNext action:\n")                                                   # VLN-POL:436
            e = emb_w[torch.tensor(self.tok.encode(text))]                                                       # VLN-POL:438-439
            rows.append(torch.cat([e[:2], patch_tok[b], inst_tok[io[b]:io[b + 1]], zone_tok[zo[b]:zo[b + 1]], e[n_vis + 2:]], 0))   # VLN-POL:456
        lengths = [r.shape[0] for r in rows]
        emb = torch.zeros(B, max(lengths), emb_w.shape[1])
        for b, r in enumerate(rows):
            emb[b, :lengths[b]] = r
        self.timing.update(vit_clip=t1 - t0, tokens_3d=t2 - t1, vit_llava=t3 - t2)
        self.counts = dict(Ni=ni, Nz=nz)
        return emb, lengths

    @torch.no_grad()
    def advance_memory(self, depth: np.ndarray, positions, headings, patch_segm, grid: np.ndarray):
        """One step of the 3D memory only (frustum delete + update, VLN-POL:349-354) on GIVEN CLIP grid features (B,576,768): brings
        the memory to a warm operating point without running the towers (bench.py's cpu_baseline)."""
        B = depth.shape[0]
        d24 = G.preprocess_depth(G.downsample_depth_nearest(depth), self.depth_scale).reshape(B, 1, -1)
        dfull = G.preprocess_depth(depth, self.depth_scale)[..., 0]
        with self._ff_threads():
            self.ff.delete_old_features_from_camera_frustum(dfull.reshape(B, 1, *dfull.shape[1:]), positions, headings)
            self.ff.update_feature_fields(d24, np.asarray(grid, np.float32).reshape(B, 1, *grid.shape[1:]), patch_segm, positions, headings)

    @torch.no_grad()
    def forward_logits(self, rgb, depth, instructions, positions, headings, patch_segm) -> np.ndarray:
        emb, lengths = self.build_inputs(rgb, depth, instructions, positions, headings, patch_segm)
        t0 = time.time()
        c = self.llm
        lo = TR.phi3_prefill_logits(emb, lengths, self.sd, c.layers, c.heads, c.kv_heads, c.rms_eps, c.rope_theta, lowp=self.lm_lowp)
        self.timing["phi3_prefill"] = time.time() - t0
        self.last_embeds, self.last_lengths = emb, lengths
        return lo.numpy()

    @torch.no_grad()
    def generate(self, rgb, depth, instructions, positions, headings, patch_segm, max_new_tokens: int = 20) -> List[str]:
        """The call the reference's trainer actually makes (VLN-POL:329 -> List[str]): build the prompt, greedy generation by
        re-running the prefix (towers_ref.phi3_greedy_decode), cut at '<|end|>', push the text into the action history
        (VLN-POL:463-468)."""
        emb, lengths = self.build_inputs(rgb, depth, instructions, positions, headings, patch_segm)
        c = self.llm
        end_id = self.tok.SPECIAL["<|end|>"] % c.vocab
        toks, _ = TR.phi3_greedy_decode(emb, lengths, self.sd, c.layers, c.heads, c.kv_heads, max_new_tokens, end_id, c.rms_eps, c.rope_theta,
                                        lowp=self.lm_lowp)
        self.last_tokens = toks
        texts = []
        for b, ids in enumerate(toks):
            t = self.tok.decode(ids)
            cut = t.find("<|end|>")
            t = t[:cut] if cut >= 0 else t
            texts.append(t)
            self.history[b].pop(0)
            self.history[b].append(t + "\n")
        return texts



class Net3DFFOracle:
    """`Net_3DFF.forward` up to the memory update (PRE-POL:136-189), inference mode, restated on the pinned pieces:
    clockwise re-ordering of the 12 panorama sensors (PRE-POL:156-163), views [0,3,6,9] (PRE-POL:165), CLIP on those images
    (PRE-POL:176), nearest 24x24 depth + preprocess_depth (PRE-POL:179-185), delete + update with `view_ids` (PRE-POL:188-189;
    the memory update itself is pinned by tests/golden/g4_prepano.npz, generated by the reference's Pretrain class)."""
    NUM_IMGS = 12

    def __init__(self, sd: Dict[str, torch.Tensor], vit_cfg, batch_size: int, depth_scale=(0.0, 10.0), view_ids=(0, 3, 6, 9)):
        self.sd, self.vit, self.depth_scale, self.view_ids = sd, vit_cfg, depth_scale, list(view_ids)
        self.ff = FeatureFieldsOracle(sd, batch_size, num_proposals=4)                    # PRE-FF:45

    @torch.no_grad()
    def forward(self, observations: Dict[str, np.ndarray], positions, headings, patch_segm):
        B, V = self.ff.batch_size, len(self.view_ids)
        depth_batch, rgb_batch = [None] * (self.NUM_IMGS * B), [None] * (self.NUM_IMGS * B)
        a_count = 0
        for k, v in observations.items():                                                   # PRE-POL:156-163
            if "depth" in k:
                for bi in range(B):
                    ra = (self.NUM_IMGS - a_count) % self.NUM_IMGS
                    depth_batch[ra + bi * self.NUM_IMGS] = np.asarray(v[bi])
                    rgb_batch[ra + bi * self.NUM_IMGS] = np.asarray(observations[k.replace("depth", "rgb")][bi])
                a_count += 1
        pick = [bi * self.NUM_IMGS + v for bi in range(B) for v in self.view_ids]            # PRE-POL:173-174
        depth = np.stack([depth_batch[i] for i in pick]).astype(np.float32)                  # (B*V,H,W,1)
        rgb = np.stack([rgb_batch[i] for i in pick])
        px = TR.preprocess_rgb(rgb, self.vit.image)
        cls, grid = TR.clip_vit_forward(px, self.sd, self.vit.layers, self.vit.heads, self.vit.patch)
        d24 = G.preprocess_depth(G.downsample_depth_nearest(depth), self.depth_scale).reshape(B, V, -1)       # PRE-POL:181-185
        origin = G.preprocess_depth(depth, self.depth_scale)[..., 0].reshape(B, V, depth.shape[1], depth.shape[2])
        self.ff.delete_old_features_from_camera_frustum(origin, positions, headings, view_ids=self.view_ids)
        self.ff.update_feature_fields(d24, grid.numpy().reshape(B, V, *grid.shape[1:]), patch_segm, positions, headings,
                                      view_ids=self.view_ids)
        return dict(rgb_embedding=cls.numpy().reshape(B, V, -1), grid_fts=grid.numpy().reshape(B, V, *grid.shape[1:]), depth24=d24)
