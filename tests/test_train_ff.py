"""Pre-training step of the feature field (PRE-FF:843-1345 `is_training=True`, PRE-TR:479-526):

  * CPU  : the host logic of `FFTrainer` / `pretrain_step` on the numpy kernel emulation (tests/cpu_ops.py) -- the collected loss and
           EVERY parameter's gradient against the float64 restatement that evaluates the reference's expressions segment by segment
           (oracle/train_ref.py), then an AdamW step whose weights reach the inference path;
  * gloo : two ranks with different episodes: NaN vote, bucketed gradient all-reduce, identical weights afterwards (DDP's contract);
  * GPU  : the same step on the HIP kernels (Linear forward / dx / dW on d3d_gemm_nt_f32, GT labelling by d3d_knn over 2e5 points)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from tests.golden_io import make_gt  # noqa: E402,F401


def run_steps(device, ops, n_gt, rank=0, steps=2, lr=1e-3, with_oracle=True):
    """The product's pre-training steps, with the ORACLE'S OWN training branch (oracle.ff_oracle ... update_feature_fields(train=...))
    advanced in lockstep FROM THE SAME RAW INPUTS under the weights the product holds at that step.  -> (ff, model, trainer,
    [(result, weights before the step, oracle views, oracle ints, product debug records)])."""
    from dynam3d_amd.feature_fields import Feature_Fields
    from dynam3d_amd.train_ff import FFTrainer, TrainableFF, pretrain_step
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    from oracle.ff_oracle import FeatureFieldsOracle
    from tests.golden_io import TRAJ_CASES, traj_inputs
    case = dict(TRAJ_CASES["prepano"], seed=5 + 17 * rank, grid_seed=9 + rank)          # the Pretrain class's 4-view panorama update
    sd = synth_state_dict(ff_param_spec(), seed=0)
    ff = Feature_Fields(case["B"], device=device, state_dict=sd, ops=ops, max_steps=(case["steps"] + 1) * 4, variant="pretrain")
    ff.initialize_camera_setting(90.0, 90.0)
    model = TrainableFF(sd, device)
    gts = [make_gt(n_gt, seed=3 + b) for b in range(case["B"])]
    trainer = FFTrainer(model, [g[0] for g in gts], [g[1] for g in gts])
    orc = FeatureFieldsOracle(sd, case["B"], num_proposals=4) if with_oracle else None
    opt = torch.optim.AdamW(model.parameters(), lr=lr)
    rng = np.random.default_rng(77 + rank)
    outs = []
    for t, inp in enumerate(traj_inputs(case)):
        if t >= steps:
            break
        img = rng.standard_normal((case["B"], 4, 768)).astype(np.float32)                # CLIP image feature of every view (batch_image_ft)
        before = {k: v.clone() for k, v in model.named_state().items()}
        views = ints = None
        if orc is not None:
            orc.sd = {k: v.detach().cpu().float() for k, v in before.items()}            # the weights this step runs under
            orc.delete_old_features_from_camera_frustum(inp["depth_full"], inp["positions"], inp["headings"], view_ids=case["view_ids"])
            orc.update_feature_fields(inp["depth24"], inp["grid"], inp["patch_segm"], inp["positions"], inp["headings"], view_ids=case["view_ids"],
                                      train=dict(gt_xyz=[g[0] for g in gts], gt_label=[g[1] for g in gts], image_ft=img))
            views, ints = orc.train_views, orc.train_ints
        ff.delete_old_features_from_camera_frustum(torch.from_numpy(inp["depth_full"]), inp["positions"], inp["headings"], view_ids=case["view_ids"])
        kw = dict(batch_depth=inp["depth24"], batch_grid_ft=inp["grid"], batch_position=inp["positions"], batch_heading=inp["headings"],
                  patch_segm=inp["patch_segm"], view_ids=case["view_ids"], batch_image_ft=img)
        res = pretrain_step(ff, trainer, opt, kw)
        outs.append((res, before, views, ints, trainer.debug))
    return ff, model, trainer, outs


def check_against_oracle(model_before, trainer, res, tol, views, debug):
    """Product vs the oracle's training branch built from the raw inputs (NOT from anything the product exported):
      * integers, exactly: the GT label of every 2D segment and the ground-truth merge target of every (segment, proposal) pair, per
        (environment, view);
      * the loss, and EVERY parameter's gradient (kept in `trainer.last_grads`) against the float64 evaluation of the oracle's records."""
    from oracle.train_ref import training_loss_and_grads
    V = len(debug)
    B = int(debug[0]["B"])
    assert len(views) == B * V
    for ix, dv in enumerate(debug):                                  # product: one record per view over all environments, groups environment-major
        env = np.asarray(dv["env_of_group"])
        gt = dv["gt"].cpu().numpy()
        for b in range(B):
            ov = views[b * V + ix]                                   # oracle: one record per (environment, view)
            assert np.array_equal(gt[env == b], ov["gt"]), (ix, b)
            assert np.array_equal(np.asarray(dv["lens"])[env == b], ov["lens"]), (ix, b)
            if ov["pairs"] is None:
                assert dv["pairs"] is None or not bool((dv["pairs"]["pe"] == b).any())
                continue
            pe = dv["pairs"]["pe"].cpu().numpy()
            assert np.array_equal(dv["pairs"]["target"].cpu().numpy()[pe == b], ov["pairs"]["target"]), (ix, b)
            g_prod = dv["pairs"]["g"].cpu().numpy()[pe == b]
            assert np.array_equal(g_prod - g_prod.min(), ov["pairs"]["g"]), (ix, b)        # same (segment, proposal) order
    loss, sim, segm, grads = training_loss_and_grads({k: v.cpu() for k, v in model_before.items()}, views)
    assert abs(loss - res["loss"]) < 2e-4 * max(1.0, abs(loss)), (loss, res)
    assert (segm is None) == (res["segm_loss"] is None)
    num = den = 0.0
    for k, g in grads.items():
        mine = trainer.last_grads[k].double().cpu()
        num += float((mine - g).pow(2).sum())
        den += float(g.pow(2).sum())
        if float(g.norm()) > 1e-6:
            r = float((mine - g).norm() / g.norm())
            assert r < 10 * tol, (k, r)
    assert (num / den) ** 0.5 < tol, (num / den) ** 0.5
    return loss, segm


def test_oracle_training_branch_matches_the_reference_golden():
    """oracle (from raw inputs) -> golden g21 = the REFERENCE's `update_feature_fields(is_training=True)` executed on the CPU
    (tests/golden/gen_golden_train.py): integers exact -- nearest GT point of every patch, GT ids of new / merged instances and of every
    instance row, zone member lists, the owner / member dictionaries, the class-balanced cross-entropy sets --, losses to 1e-6,
    gradients (norm, sum, 64 probes per parameter) to 2e-5."""
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    from oracle.ff_oracle import FeatureFieldsOracle
    from oracle.train_ref import training_loss_and_grads
    from tests.golden_io import int_hash, load, pack_ragged, train_inputs, unpack_ragged
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    g = load("g21_train.npz")
    case, gts, steps = train_inputs()
    B = case["B"]
    sd = synth_state_dict(ff_param_spec(), seed=0)
    orc = FeatureFieldsOracle(sd, B, num_proposals=4)
    for t, (inp, img) in enumerate(steps):
        orc.delete_old_features_from_camera_frustum(inp["depth_full"], inp["positions"], inp["headings"], view_ids=case["view_ids"])
        orc.update_feature_fields(inp["depth24"], inp["grid"], inp["patch_segm"], inp["positions"], inp["headings"], view_ids=case["view_ids"],
                                  train=dict(gt_xyz=[x[0] for x in gts], gt_label=[x[1] for x in gts], image_ft=img))
        p = f"t{t}_"
        for b in range(B):
            q = f"{p}b{b}_"
            e, ints = orc.env[b], orc.train_ints
            ref_nn = unpack_ragged(g[q + "gt_nn"], g[q + "gt_nn_off"])
            assert len(ref_nn) == len(ints["gt_nn"][b]) and all(np.array_equal(a, c) for a, c in zip(ref_nn, ints["gt_nn"][b]))
            assert np.array_equal(g[q + "gt3d"], np.asarray(ints["gt3d"][b], np.int64))
            assert np.array_equal(g[q + "row_gt"], e.row_gt)
            ref_z = unpack_ragged(g[q + "zone_mem"], g[q + "zone_mem_off"])
            assert len(ref_z) == len(ints["gt_in_zone"][b]) and all(np.array_equal(a, c) for a, c in zip(ref_z, ints["gt_in_zone"][b]))
            ks = np.array(sorted(e.owner.keys()), np.int64)
            im, imo = pack_ragged([e.members[k] for k in e.members])
            assert np.array_equal(g[q + "inst_order"], np.array(list(e.members.keys()), np.int64))
            assert int_hash(ks, np.array([e.owner[k] for k in ks.tolist()], np.int64), im, imo) == int(g[q + "book_hash"])
        if t == case["steps"] - 1 or t == 1:                       # the float64 evaluation: a warm step with merges + the last one
            loss, sim, segm, grads = training_loss_and_grads(sd, orc.train_views)
            assert abs(sim - float(g[p + "sim_loss"])) < 1e-6 * abs(sim) and abs(segm - float(g[p + "segm_loss"])) < 1e-6, (sim, segm)
            ce = training_loss_and_grads.last_ce_records
            ref_t = unpack_ragged(g[p + "ce_target"], g[p + "ce_off"])
            assert len(ce) == len(ref_t) and all(np.array_equal(c[1], r) for c, r in zip(ce, ref_t))
            assert np.abs(np.concatenate([c[0] for c in ce], 0) - g[p + "ce_score"]).max() < 1e-5
            for k, gr in grads.items():
                gr = gr.numpy()
                n_ref = float(g[p + "gnorm_" + k])
                assert abs(np.linalg.norm(gr) - n_ref) < 2e-5 * max(n_ref, 1e-6), k
                f = gr.reshape(-1)
                pr = f[:: max(1, f.size // 64)][:64]
                assert np.abs(pr - g[p + "gprobe_" + k]).max() < 2e-5 * max(np.abs(g[p + "gprobe_" + k]).max(), 1e-5) + 1e-9, k


def test_pretrain_step_on_cpu_emulation_matches_float64_oracle():
    from tests.cpu_ops import CpuOps
    ff, model, trainer, outs = run_steps("cpu", CpuOps(), n_gt=20000)
    (r0, b0, v0, i0, d0), (r1, b1, v1, i1, d1) = outs
    assert not r0["skipped"] and not r1["skipped"]
    assert r1["segm_loss"] is not None, "the second step must see merge proposals with both classes (ground-truth merges)"
    loss, segm = check_against_oracle(b1, trainer, r1, 2e-4, v1, d1)
    # the optimizer moved the weights and the inference-path copies follow
    moved = sum(float((model.named_state()[k] - b1[k]).abs().max()) > 0 for k in b1)
    assert moved >= len(b1) - 2
    for k, v in model.named_state().items():
        assert torch.equal(ff.dense.w[k].cpu(), v.cpu())
    # the memory merged by ground truth: instance rows carry GT ids
    assert any((t >= 0).any() for t in trainer.gt_ids_of_slot.values())


WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch
from dynam3d_amd import dist as D
from tests.cpu_ops import CpuOps
from tests.test_train_ff import run_steps
rank, local, world = D.init_from_env("gloo")
torch.set_num_threads(2)
ff, model, trainer, outs = run_steps("cpu", CpuOps(), n_gt=5000, rank=rank, steps=2, with_oracle=False)
st = model.named_state()
flat = torch.cat([v.reshape(-1) for v in st.values()])
avg = torch.cat([trainer.last_grads[k].reshape(-1) for k in st])            # the gradient after the all-reduce (identical on all ranks)
loc = torch.cat([trainer.local_grads[k].reshape(-1) for k in st])           # this rank's own RAW gradient (before the reduction)
print("RESULT", json.dumps(dict(rank=rank, collectives=outs[-1][0]["collectives"], loss=outs[-1][0]["loss"], wsum=float(flat.double().sum()),
                                wabs=float(flat.double().abs().sum()), avg=[float(avg.double().sum()), float(avg.double().abs().sum())],
                                loc=[float(loc.double().sum()), float(loc.double().abs().sum())], probe=avg[::997][:64].double().tolist(),
                                lprobe=loc[::997][:64].double().tolist())))
D.barrier()
D.shutdown()
"""


def test_pretrain_step_two_ranks_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    w = tmp_path / "w.py"
    w.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    import json
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=900)
        assert p.returncode == 0, o[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT")][0][7:]))
    a, b = sorted(outs, key=lambda o: o["rank"])
    assert a["collectives"] >= 1 and a["collectives"] == b["collectives"]
    assert a["loss"] != b["loss"]                                                   # different episodes per rank
    assert a["wsum"] == b["wsum"] and a["wabs"] == b["wabs"]                        # identical weights after the step (DDP's contract)
    assert np.allclose(a["probe"], b["probe"], rtol=0, atol=0)                      # the reduced gradient is the same tensor on both ranks ...
    # ... = clip(mean of the ranks' own raw gradients): the reference clips AFTER DDP's averaging (PRE-TR:512-517), not before
    assert np.allclose(np.array(a["probe"]), np.clip((np.array(a["lprobe"]) + np.array(b["lprobe"])) / 2, -10.0, 10.0), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_pretrain_step_on_hip_kernels_matches_float64_oracle():
    from dynam3d_amd.ops import HipOps
    ff, model, trainer, outs = run_steps("cuda", HipOps(), n_gt=200000)
    (r0, b0, v0, i0, d0), (r1, b1, v1, i1, d1) = outs
    assert not r1["skipped"] and r1["segm_loss"] is not None
    loss, segm = check_against_oracle(b1, trainer, r1, 5e-4, v1, d1)
    print(f"pre-training step on the GPU: loss {r1['loss']:.5f} (sim {r1['sim_loss']:.5f}, segm {r1['segm_loss']:.5f}); float64 oracle loss {loss:.5f}; "
          f"gradient relative L2 error within 5e-4 over {len(b1)} parameter tensors")
    for k, v in model.named_state().items():
        assert torch.equal(ff.dense.w[k], v)


def test_dead_instance_rows_lose_their_gt_id():
    """PRE-FF:728: an instance that dies in delete_old_features_from_camera_frustum loses its ground-truth id (-10000), so a proposal that
    lands on the dead row (fewer than K live instances) can never match a segment's GT id and is never merged into.  The product keeps the
    GT table next to the instance pool and takes the reset from the pool's tomb-stones."""
    from types import SimpleNamespace
    from dynam3d_amd.train_ff import FFTrainer
    tr = FFTrainer.__new__(FFTrainer)
    tr.gt_rows = None
    pools = SimpleNamespace(inst_pos=torch.zeros((2, 6, 3)))
    t = tr._gt_rows(pools)
    assert t.shape == (2, 6) and bool((t == -1).all())
    t[0, :4] = torch.tensor([7, 3, 7, 9])
    t[1, :2] = torch.tensor([3, 3])
    pools.inst_pos[0, 2] = -10000.0                       # instance row 2 of slot 0 dies (fill_rows / d3d_ffdev_apply_hits tomb-stone it)
    pools.inst_pos[1, 0] = -10000.0
    t = tr._gt_rows(pools)
    assert t[0].tolist() == [7, 3, -10000, 9, -1, -1] and t[1].tolist() == [-10000, 3, -1, -1, -1, -1]
    seg_gt = torch.tensor([7, 3])                         # two segments whose proposals land on (slot 0, row 2) and (slot 1, row 0): dead rows
    assert not bool((t[torch.tensor([0, 1]), torch.tensor([2, 0])] == seg_gt).any())
    pools.inst_pos[0, 2] = torch.tensor([1.0, 2.0, 3.0])  # the row is recycled: new position, new id
    tr._gt_rows(pools)[0, 2] = 11
    assert tr._gt_rows(pools)[0].tolist() == [7, 3, 11, 9, -1, -1]
