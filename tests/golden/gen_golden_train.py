"""Golden g21: the REFERENCE's training branch -- `Feature_Fields.update_feature_fields(is_training=True)` of the Pretrain class
(PRE-FF:843-1345: GT labelling 976-986, ground-truth merges 1029-1047, loss assembly 1302-1345) -- EXECUTED on the CPU by
oracle/ref_harness.RefTrainingRun (the one shim: mixed float32 x float16 `torch.matmul` operands are promoted; module in eval() mode,
i.e. dropout p = 0), with autograd's gradient of sim_loss + segm_loss.  Container-only.

Inputs are seeds (tests/golden_io.train_inputs: the `prepano` panorama walk, a 20 000-point synthetic GT instance cloud per environment,
CLIP image features per view).  Stored per step: the two losses; per environment the nearest-GT-point index of every patch in call
order, the GT ids returned for new / merged instances, the GT id of every instance row, the member list of every zone operation, a
digest of the owner / member dictionaries; every cross-entropy call's class-balanced (score, target) set; per parameter the gradient's
L2 norm, sum and 64 strided probe values."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from tests.golden_io import int_hash, pack_ragged, train_inputs  # noqa: E402
from dynam3d_amd.weights import ff_param_spec, render_param_spec, synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def probe(g, n=64):
    f = np.asarray(g, np.float32).reshape(-1)
    return f[:: max(1, f.size // n)][:n].copy()


def main():
    torch.set_num_threads(16)
    case, gts, steps = train_inputs()
    B = case["B"]
    sd = synth_state_dict(ff_param_spec() + render_param_spec(), seed=0)
    ref = rh.RefTrainingRun(B, sd, [g[0] for g in gts], [g[1] for g in gts])
    out = {"B": np.int64(B), "steps": np.int64(case["steps"])}
    for t, (inp, img) in enumerate(steps):
        r = ref.train_step(torch.from_numpy(inp["depth_full"]), inp["depth24"], inp["grid"], torch.from_numpy(inp["patch_segm"]), inp["positions"],
                           inp["headings"], case["view_ids"], torch.from_numpy(img))
        p = f"t{t}_"
        out[p + "sim_loss"], out[p + "segm_loss"], out[p + "has_segm"] = np.float64(r["sim_loss"]), np.float64(r["segm_loss"]), np.int64(r["has_segm"])
        F = ref.F
        for b in range(B):
            q = f"{p}b{b}_"
            out[q + "gt_nn"], out[q + "gt_nn_off"] = pack_ragged(r["gt_nn"][b])
            out[q + "gt3d"], out[q + "row_gt"] = r["gt3d"][b], r["row_gt"][b]
            out[q + "zone_mem"], out[q + "zone_mem_off"] = pack_ragged(r["gt_in_zone"][b])
            own = F.global_patch_to_instance_dict[b]
            ks = np.array(sorted(own.keys()), np.int64)
            mem = F.global_instance_to_patch_dict[b]
            im, imo = pack_ragged([mem[k] for k in mem])
            out[q + "inst_order"] = np.array(list(mem.keys()), np.int64)
            out[q + "book_hash"] = int_hash(ks, np.array([own[k] for k in ks.tolist()], np.int64), im, imo)
            out[q + "pred3d_head"] = r["pred3d"][b][:, :16].copy()
            out[q + "pred3d_rowsum"] = r["pred3d"][b].astype(np.float64).sum(1)
        out[p + "ce_target"], out[p + "ce_off"] = pack_ragged([c[1] for c in r["ce_calls"]])
        out[p + "ce_score"] = np.concatenate([c[0] for c in r["ce_calls"]], 0).astype(np.float32) if r["ce_calls"] else np.zeros((0, 2), np.float32)
        for k, g in r["grads"].items():
            out[p + "gnorm_" + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
            out[p + "gsum_" + k] = np.float64(g.astype(np.float64).sum())
            out[p + "gprobe_" + k] = probe(g)
        print("g21 step", t, r["sim_loss"], r["segm_loss"], len(r["ce_calls"]), [len(x) for x in r["gt3d"]], flush=True)
    np.savez_compressed(os.path.join(OUT, "g21_train.npz"), **out)
    print("g21 ok", os.path.getsize(os.path.join(OUT, "g21_train.npz")))


if __name__ == "__main__":
    main()
