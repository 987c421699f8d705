"""b1: the net as the UNCHANGED reference trainer uses it (Dynam3D_VLN/vlnce_baselines/ss_trainer_Dynam3D.py, "VLN-TR"), replayed
call by call against a stub of `ILPolicy` (models/policy.py:12-19: an nn.Module whose only child is `self.net`).  CPU, small
towers, HIP kernels swapped for tests/cpu_ops.py; the sequence itself is what is under test:

  VLN-TR:183      self.policy.to(self.device)
  VLN-TR:192-198  Adafactor(self.policy.net.parameters(), ...)
  VLN-TR:211-219  'module' checkpoints: DataParallel wrap -> policy.load_state_dict(strict=False) -> unwrap; else load_state_dict(strict=False)
  VLN-TR:222-225  sum(p.numel() for p in self.policy.parameters()), ... if p.requires_grad
  VLN-TR:305-311  self.policy.train(); self.policy.net.rgb_encoder.eval(); self.policy.net.depth_encoder.eval()
  VLN-TR:374      self.policy.eval()
  VLN-TR:566-568  policy_net = self.policy.net; hasattr(self.policy.net, 'module')
  VLN-TR:621-622  policy_net.feature_fields.reset(batch_size); .initialize_camera_setting(hfov=90., vfov=90.)
  VLN-TR:656-664  policy_net.feature_fields.keep_target_waypoint[b]; policy_net.get_gt_text(...)
  VLN-TR:671      self.policy.net(batch, instructions, positions, headings, depth_scale=(0.,10.), gt_text=..., delete_old_features=True,
                                   num_of_views=1, is_train=False) -> List[str]
  VLN-TR:693      policy_net.convert_text_to_action(generated_text)
  VLN-TR:783      policy_net.feature_fields.pop(i)
  VLN-TR:803      policy_net.feature_fields.delete_feature_fields()
"""
import copy
import json
import os

import pytest
import torch
from torch import nn

from dynam3d_amd.policy import PREFIX_MLPS, Dynam3D_VLN, prefix_param_spec, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
from dynam3d_amd.towers import clip_param_spec, llava_vision_param_spec, phi3_param_spec
from dynam3d_amd.weights import ff_param_spec
from tests.cpu_ops import CpuOps
from tests.test_policy_cpu import SMALL

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class ILPolicyStub(nn.Module):
    """models/policy.py:12-19 without habitat: `super(Policy, self).__init__(); self.net = net; self.dim_actions = dim_actions`."""

    def __init__(self, net, dim_actions=4):
        super().__init__()
        self.net = net
        self.dim_actions = dim_actions


def _net(seed=0, B=2, max_steps=4):
    sd = synth_policy_weights(SMALL, seed=seed)
    return Dynam3D_VLN(SMALL, sd, device="cpu", batch_size=B, ops=CpuOps(), max_steps=max_steps), sd


def _expected_keys(cfg):
    keys = ["feature_fields." + n for n, _ in ff_param_spec(768)]
    keys += [n for n, _ in prefix_param_spec(768, cfg.llm.hidden)]
    keys += ["llava." + n for n, _ in phi3_param_spec(cfg.llm) + llava_vision_param_spec(cfg.vit)]
    keys += ["rgb_encoder.model." + n for n, _ in clip_param_spec(cfg.vit)]
    return keys


def _frames(B, n, seed=11):
    ep = SyntheticEpisodes(B, seed=seed, image_hw=224, depth_hw=224)
    out = []
    for _ in range(n):
        fr = ep.next()
        out.append((dict(rgb=torch.from_numpy(fr.rgb), depth=torch.from_numpy(fr.depth)), [p.tolist() for p in fr.positions], list(fr.headings),
                    fr.patch_segm))
    return out


def test_module_tree_has_the_reference_state_dict_keys():
    net, sd = _net()
    policy = ILPolicyStub(net)
    assert isinstance(net, nn.Module) and isinstance(net.feature_fields, nn.Module) and isinstance(net.rgb_encoder, nn.Module)
    for name in PREFIX_MLPS:                                                            # VLN-POL:83-111: real nn.Sequential(Linear, LayerNorm, GELU, Linear)
        seq = getattr(net, name)
        assert isinstance(seq, nn.Sequential) and [type(m) for m in seq] == [nn.Linear, nn.LayerNorm, nn.GELU, nn.Linear]
    got = policy.state_dict()
    want = ["net." + k for k in _expected_keys(SMALL)]
    assert sorted(got) == sorted(want)
    assert not hasattr(policy.net, "module")                                            # VLN-TR:568
    # every tensor equals the checkpoint tensor it was built from (in its storage dtype)
    flat_src = {**{"feature_fields." + k: sd[k] for k, _ in ff_param_spec(768)}, **{k: sd[k] for k, _ in prefix_param_spec(768, SMALL.llm.hidden)},
                **{"llava." + k: sd[k] for k, _ in phi3_param_spec(SMALL.llm) + llava_vision_param_spec(SMALL.vit)},
                **{"rgb_encoder.model." + k: sd[k] for k, _ in clip_param_spec(SMALL.vit)}}
    for k, v in got.items():
        assert torch.equal(v.float().cpu(), flat_src[k[4:]].to(v.dtype).float()), k
    # trainable flags: VLN-POL:150-155 (feature field, vision tower, projector frozen), resnet_encoders.py:262-264 (CLIP frozen)
    rg = {k: p.requires_grad for k, p in policy.named_parameters()}
    assert not any(v for k, v in rg.items() if k.startswith(("net.feature_fields.", "net.llava.vision_tower.", "net.llava.multi_modal_projector.", "net.rgb_encoder.")))
    assert all(v for k, v in rg.items() if k.startswith("net.llava.language_model.") or k.split(".")[1] in PREFIX_MLPS)
    n_all = sum(p.numel() for p in policy.parameters())                                 # VLN-TR:222-225
    n_train = sum(p.numel() for p in policy.parameters() if p.requires_grad)
    assert 0 < n_train < n_all


def test_trainer_call_sequence_and_checkpoint_round_trip():
    B = 2
    frames = _frames(B, 3)
    instr = [INSTRUCTION_64] * B
    device = torch.device("cpu")

    # ---- the checkpoint a trainer would have written from ANOTHER set of weights (seed 1): ckpt["state_dict"] = policy.state_dict()
    donor, _ = _net(seed=1, B=B)
    ckpt = {"state_dict": {k: v.clone() for k, v in ILPolicyStub(donor).state_dict().items()}, "iteration": 7}
    ckpt["state_dict"]["net.depth_encoder.visual_encoder.backbone.conv1.0.weight"] = torch.zeros(4)      # keys of modules outside the hot path
    ckpt["state_dict"]["net.feature_fields.FastSAM.model.model.0.conv.weight"] = torch.zeros(4)          # are tolerated (strict=False)
    donor.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
    want_logits = [donor.forward_logits(o, instr, p, h, patch_segm=s).clone() for o, p, h, s in frames]
    del donor

    # ---- _initialize_policy (VLN-TR:170-225) on a policy built from seed-0 weights
    net, _ = _net(seed=0, B=1)
    policy = ILPolicyStub(net)
    policy.to(device)                                                                   # :183
    opt = torch.optim.SGD([p for p in policy.net.parameters() if p.requires_grad], lr=1e-3)     # :192 (Adafactor there; any optimizer takes the list)
    assert len(opt.param_groups[0]["params"]) > 0
    before = policy.net.forward_logits  # noqa: F841  (bound method survives the load)
    missing, unexpected = policy.load_state_dict(ckpt["state_dict"], strict=False)      # :219
    assert missing == [], missing
    assert sorted(unexpected) == ["net.feature_fields.FastSAM.model.model.0.conv.weight"], unexpected    # depth_encoder.* swallowed by its slot
    policy.train()                                                                      # :305
    policy.net.rgb_encoder.eval()                                                       # :309
    policy.net.depth_encoder.eval()                                                     # :310
    assert policy.training and not policy.net.rgb_encoder.training and not policy.net.depth_encoder.training
    policy.eval()                                                                       # :374
    assert not policy.net.training and not policy.net.feature_fields.training

    # ---- rollout (VLN-TR:566-806)
    policy_net = policy.net
    if hasattr(policy.net, "module"):
        policy_net = policy.net.module
    policy_net.feature_fields.reset(B)                                                  # :621
    policy_net.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)             # :622
    assert policy_net.feature_fields.keep_target_waypoint == [None] * B                 # :656
    got_logits = []
    for t, (obs, pos, hd, segm) in enumerate(frames):
        got_logits.append(policy_net.forward_logits(obs, instr, pos, hd, patch_segm=segm).clone())
    # the loaded weights ARE the ones the kernels compute with: logits equal the donor's bit for bit
    for a, b in zip(got_logits, want_logits):
        assert torch.equal(a, b)

    # the per-step call itself, with the trainer's keyword arguments, returns one sentence per environment
    policy_net.feature_fields.reset(B)
    policy_net.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
    policy_net.feature_fields.segmenter = lambda imgs, **kw: frames[0][3]               # the a6 contract: (B*V,1,24,24) dense labels
    out = policy.net(frames[0][0], instr, frames[0][1], frames[0][2], depth_scale=(0., 10.), gt_text=None, delete_old_features=True,
                     num_of_views=1, is_train=False)                                    # :671
    assert isinstance(out, list) and len(out) == B and all(isinstance(s, str) for s in out)
    acts = policy_net.convert_text_to_action(out)                                       # :693
    assert len(acts) == B
    policy_net.feature_fields.pop(1)                                                    # :783
    assert policy_net.feature_fields.batch_size == B - 1
    policy_net.feature_fields.delete_feature_fields()                                   # :803
    assert policy_net.feature_fields.batch_size == 0

    # ---- save_checkpoint / resume: state_dict() round-trips every tensor
    again = policy.state_dict()
    for k, v in ckpt["state_dict"].items():
        if k in again:
            assert torch.equal(again[k], v), k


def test_dataparallel_module_checkpoint_branch():
    """VLN-TR:211-217: a checkpoint saved from a DDP / DataParallel run has `net.module.*` keys; with one GPU the trainer wraps the
    net in DataParallel, loads, and unwraps (`self.policy.net = self.policy.net.module`)."""
    donor, _ = _net(seed=2, B=1)
    sd = {k.replace("net.", "net.module.", 1): v.clone() for k, v in ILPolicyStub(donor).state_dict().items()}
    net, _ = _net(seed=0, B=1)
    policy = ILPolicyStub(net)
    assert "module" in list(sd.keys())[0]
    policy.net = torch.nn.DataParallel(policy.net, device_ids=None)
    missing, unexpected = policy.load_state_dict(sd, strict=False)
    assert missing == [] and unexpected == []
    policy.net = policy.net.module
    assert isinstance(policy.net, Dynam3D_VLN)
    fr = _frames(1, 1)[0]
    for n in (donor, policy.net):
        n.feature_fields.reset(1)
        n.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
    a = donor.forward_logits(fr[0], [INSTRUCTION_64], fr[1], fr[2], patch_segm=fr[3])
    b = policy.net.forward_logits(fr[0], [INSTRUCTION_64], fr[1], fr[2], patch_segm=fr[3])
    assert torch.equal(a, b)


def test_to_dtype_and_deepcopy_keep_kernel_views_in_sync():
    """`_apply` (to / float / double ...) replaces parameter storages: the kernel-side layouts are re-derived, results unchanged."""
    net, _ = _net(seed=0, B=1)
    fr = _frames(1, 1)[0]
    net.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
    a = net.forward_logits(fr[0], [INSTRUCTION_64], fr[1], fr[2], patch_segm=fr[3]).clone()
    net.to(torch.device("cpu"))
    net.float()                                                      # all-float32 small config: same values, new storages are allowed
    net.feature_fields.reset(1)
    b = net.forward_logits(fr[0], [INSTRUCTION_64], fr[1], fr[2], patch_segm=fr[3])
    assert torch.equal(a, b)
    # an in-place overwrite of a RE-LAID-OUT weight (HF q/k/v are fused into one GEMM operand) + refresh() reaches the kernels
    with torch.no_grad():
        p = dict(net.named_parameters())["llava.vision_tower.vision_model.encoder.layers.0.self_attn.q_proj.weight"]
        p.mul_(0.5)
    net.refresh()
    net.feature_fields.reset(1)
    c = net.forward_logits(fr[0], [INSTRUCTION_64], fr[1], fr[2], patch_segm=fr[3])
    assert not torch.equal(a, c)


def test_feature_fields_constructs_and_loads_like_the_reference():
    """VLN-POL:77-80: `Feature_Fields(batch_size=1, device=...)`, then `load_state_dict(torch.load("dynam3d.pth"), strict=True)`, `.eval()`."""
    from dynam3d_amd.feature_fields import Feature_Fields
    from dynam3d_amd.weights import synth_state_dict
    ff = Feature_Fields(batch_size=1, device="cpu", ops=CpuOps())
    sd = synth_state_dict(ff_param_spec(768), seed=5)
    res = ff.load_state_dict(sd, strict=True)
    assert list(res.missing_keys) == [] and list(res.unexpected_keys) == []
    assert ff.eval() is ff
    for k, v in ff.state_dict().items():
        assert torch.equal(v, sd[k]), k
        assert ff.dense.w[k].data_ptr() == dict(ff.named_parameters())[k].data_ptr()          # the kernels alias the parameters
    for p in ff.parameters():
        p.requires_grad_(False)                                                                # VLN-POL:150-151


def test_get_gt_text_matches_reference_golden():
    """g20: the reference's own `get_gt_text` (VLN-POL:294-327) executed in the build container on seeded cases."""
    cases = json.load(open(os.path.join(GOLD, "g20_gt_text.json")))
    net, _ = _net(seed=0, B=1)
    n_err = 0
    for c in cases:
        B = len(c["angles"])
        net.feature_fields.reset(B)
        net.feature_fields.history_actions = copy.deepcopy(c["history"])
        got = net.get_gt_text(list(c["angles"]), list(c["distances"]), list(c["stops"]))
        assert got == c["text"], (got, c["text"])
        keep = net.feature_fields.keep_target_waypoint
        for k, w in zip(keep, c["keep"]):
            assert (k is None) == (w is None)
            if w is not None:
                assert float(k[0]) == w[0] and float(k[1]) == w[1]
        n_err += sum(t == "error.<|end|>" for t in got)
    assert n_err > 0


@pytest.mark.gpu
def test_checkpoint_round_trip_on_the_hip_kernels():
    """The same protocol on the MI355X in the reference's dtypes (fp16 CLIP, bf16 llava): the re-laid-out kernel operands (gate/up rows
    interleaved per 16, fused q/k/v, padded patch embedding) follow `policy.load_state_dict`, and `.to('cuda')` of a net that already
    lives there costs nothing.  Strict HIP mode: no PyTorch fallback."""
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.ops import HipOps
    from tests.test_policy_cpu import MID
    D.strict(True)
    try:
        B = 2
        frames = _frames(B, 2)
        instr = [INSTRUCTION_64] * B
        mk = lambda seed: Dynam3D_VLN(MID, synth_policy_weights(MID, seed=seed), device="cuda", batch_size=B, ops=HipOps(), max_steps=4)
        donor = mk(1)
        ckpt = {k: v.clone() for k, v in ILPolicyStub(donor).state_dict().items()}
        donor.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
        want = [donor.forward_logits(o, instr, p, h, patch_segm=s).clone() for o, p, h, s in frames]
        policy = ILPolicyStub(mk(0))
        gu = policy.net.llm.layers[0]["gu_w"]
        policy.to(torch.device("cuda"))
        assert policy.net.llm.layers[0]["gu_w"] is gu                      # nothing moved: the kernel layouts were not rebuilt
        assert {v.dtype for k, v in policy.state_dict().items() if k.startswith("net.llava.")} == {torch.bfloat16}
        assert policy.state_dict()["net.rgb_encoder.model.visual.conv1.weight"].dtype == torch.float16
        assert policy.state_dict()["net.rgb_encoder.model.visual.ln_pre.weight"].dtype == torch.float32   # convert_weights leaves LayerNorm in fp32
        missing, unexpected = policy.load_state_dict(ckpt, strict=False)
        assert missing == [] and unexpected == []
        policy.eval()
        policy.net.feature_fields.reset(B)
        policy.net.feature_fields.initialize_camera_setting(hfov=90., vfov=90.)
        got = [policy.net.forward_logits(o, instr, p, h, patch_segm=s) for o, p, h, s in frames]
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        policy.net.feature_fields.reset(B)
        texts = policy.net(frames[0][0], instr, frames[0][1], frames[0][2], depth_scale=(0., 10.), gt_text=None, delete_old_features=True,
                           num_of_views=1, is_train=False, patch_segm=frames[0][3])
        assert len(texts) == B and len(policy.net.convert_text_to_action(texts)) == B
    finally:
        D.strict(True)                                              # (the default)
