"""Golden g15: the reference's own `contrastive_loss` / `sim_matrix_cross_entropy` (PRE-FF:828-840) EXECUTED from the Pretrain
`Feature_Fields` class on seeded unit-norm feature pairs, with autograd gradients.  Container-only.
(The `sim_loss` assembly around them, PRE-FF:1302-1330, sits inside `update_feature_fields(is_training=True)`, which does not run
on a CPU -- SURVEY.md F12 -- and is restated in oracle/losses_ref.py.)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    mod = rh.load_ref_module("pretrain")
    FF = mod.Feature_Fields
    out = {}
    rng = np.random.default_rng(150)
    cases = [(5, 768), (64, 768), (257, 768), (1, 768)]
    for i, (n, d) in enumerate(cases):
        a = rng.standard_normal((n, d)).astype(np.float32)
        b = (0.7 * a + 0.7 * rng.standard_normal((n, d))).astype(np.float32)
        a /= np.linalg.norm(a, axis=-1, keepdims=True)
        b /= np.linalg.norm(b, axis=-1, keepdims=True)
        ta, tb = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
        loss = FF.contrastive_loss(None, ta, tb) if False else FF.contrastive_loss(_Self(FF), ta, tb)
        loss.backward()
        out[f"a_{i}"], out[f"b_{i}"] = a, b
        out[f"loss_{i}"], out[f"ga_{i}"], out[f"gb_{i}"] = loss.detach().numpy(), ta.grad.numpy(), tb.grad.numpy()
        sim = torch.from_numpy((10.0 * a @ b.T).astype(np.float32))
        out[f"xent_{i}"] = FF.sim_matrix_cross_entropy(None, sim).numpy()
        print("g15", n, float(loss))
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(OUT, "g15_losses.npz"), **out)


class _Self:
    """`contrastive_loss` only touches `self.sim_matrix_cross_entropy`: bind the reference's own function without building the module."""
    def __init__(self, FF):
        self._ff = FF

    def sim_matrix_cross_entropy(self, sim):
        return self._ff.sim_matrix_cross_entropy(self, sim)


if __name__ == "__main__":
    main()
