"""CPU: the product policy's host logic (depth path, shared preprocessing, prefix MLPs, splice, right padding)
against the whole-step oracle, small tower configs, float32, HIP kernels swapped for tests/cpu_ops.py."""
import numpy as np
import torch

from dynam3d_amd.policy import Dynam3D_VLN, PolicyConfig, SyntheticTokenizer, synth_policy_weights
from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
from dynam3d_amd.towers import Phi3Config, VitConfig
from oracle.step_oracle import StepOracle
from tests.cpu_ops import CpuOps

# 24x24 patch grid is structural (576 patches); keep it and shrink widths/depths instead
SMALL = PolicyConfig(vit=VitConfig(image=336, patch=14, width=64, layers=2, heads=2, mlp=128, out_dim=768, proj_dim=96),
                     llm=Phi3Config(vocab=640, hidden=96, layers=2, heads=4, kv_heads=4, mlp=192),
                     clip_dtype=torch.float32, llava_dtype=torch.float32)


# The smallest configuration every dense primitive of which is a HIP kernel (strict mode): GEMM N % 128 == 0 and K % 64 == 0, head
# dims 64 (ViT) and 96 (LM), 16-bit towers.  Used by smoke() and the strict GPU step tests.
MID = PolicyConfig(vit=VitConfig(image=336, patch=14, width=256, layers=2, heads=4, mlp=512, out_dim=768, proj_dim=384),
                   llm=Phi3Config(vocab=640, hidden=384, layers=2, heads=4, kv_heads=4, mlp=512),
                   clip_dtype=torch.float16, llava_dtype=torch.bfloat16)


def toy_dense(cfg, device):
    """Context for a run of `cfg` on `device`: toy tower configurations (float32 towers, widths below the HIP kernels' 128 x 64 tiles)
    on a GPU have no HIP kernel for their dense primitives -- they opt out of dense_ops' strict default EXPLICITLY; everything
    else (MID, the full configuration, the CPU) runs under the default."""
    import contextlib
    from dynam3d_amd import dense_ops as D
    toy = cfg.clip_dtype == torch.float32 or cfg.llava_dtype == torch.float32 or cfg.vit.width % 128 or cfg.llm.hidden % 128
    return D.allow_fallback() if (str(device) != "cpu" and toy) else contextlib.nullcontext()


def run_policy_vs_oracle(ops, device, cfg, steps=3, B=2, tol=2e-4, check_embeds=True, lowp_oracle=False):
    """lowp_oracle: the oracle evaluates the towers in cfg's dtypes with the reference's rounding points (towers_ref `lowp`)."""
    with toy_dense(cfg, device):
        return _run_policy_vs_oracle(ops, device, cfg, steps, B, tol, check_embeds, lowp_oracle)


def _run_policy_vs_oracle(ops, device, cfg, steps, B, tol, check_embeds, lowp_oracle):
    sd = synth_policy_weights(cfg, seed=0)
    net = Dynam3D_VLN(cfg, sd, device=device, batch_size=B, ops=ops, max_steps=steps + 1)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    kw = dict(clip_dtype=cfg.clip_dtype, llava_dtype=cfg.llava_dtype) if lowp_oracle else {}
    orc = StepOracle(sd, cfg.vit, cfg.llm, B, SyntheticTokenizer(cfg.llm.vocab), **kw)
    ep = SyntheticEpisodes(B, seed=3, image_hw=224, depth_hw=224)
    instr = [INSTRUCTION_64] * B
    worst = 0.0
    for t in range(steps):
        fr = ep.next()
        pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
        obs = {"rgb": torch.from_numpy(fr.rgb), "depth": torch.from_numpy(fr.depth)}
        lo = net.forward_logits(obs, instr, pos, hd, patch_segm=fr.patch_segm).float().cpu().numpy()
        ref = orc.forward_logits(fr.rgb, fr.depth, instr, pos, hd, fr.patch_segm)
        assert net.last_lengths == orc.last_lengths and net.last_counts == orc.counts
        r = np.linalg.norm(lo - ref) / np.linalg.norm(ref)
        worst = max(worst, r)
        assert r < tol, (t, r)
        assert np.array_equal(lo.argmax(-1), ref.argmax(-1)) or tol > 1e-3
    return worst


def run_policy_three_way(ops, device, cfg, steps=2, B=2):
    """The product in cfg's 16-bit dtypes beside BOTH oracles in lockstep: float32, and `lowp` (the reference's rounding points in
    cfg's dtypes).  Returns the worst relative-L2 distances (product-lowp, product-float32, lowp-float32): the last one is the noise
    band a 16-bit evaluation of this network has by construction."""
    sd = synth_policy_weights(cfg, seed=0)
    net = Dynam3D_VLN(cfg, sd, device=device, batch_size=B, ops=ops, max_steps=steps + 1)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    tok = SyntheticTokenizer(cfg.llm.vocab)
    o32 = StepOracle(sd, cfg.vit, cfg.llm, B, tok)
    o16 = StepOracle(sd, cfg.vit, cfg.llm, B, tok, clip_dtype=cfg.clip_dtype, llava_dtype=cfg.llava_dtype)
    ep = SyntheticEpisodes(B, seed=3, image_hw=224, depth_hw=224)
    instr = [INSTRUCTION_64] * B
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    worst = [0.0, 0.0, 0.0]
    for t in range(steps):
        fr = ep.next()
        pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
        obs = {"rgb": torch.from_numpy(fr.rgb), "depth": torch.from_numpy(fr.depth)}
        lo = net.forward_logits(obs, instr, pos, hd, patch_segm=fr.patch_segm).float().cpu().numpy()
        r32 = o32.forward_logits(fr.rgb, fr.depth, instr, pos, hd, fr.patch_segm)
        r16 = o16.forward_logits(fr.rgb, fr.depth, instr, pos, hd, fr.patch_segm)
        assert net.last_lengths == o32.last_lengths == o16.last_lengths and net.last_counts == o32.counts == o16.counts
        for i, d in enumerate((rel(lo, r16), rel(lo, r32), rel(r16, r32))):
            worst[i] = max(worst[i], d)
    return tuple(worst)


def test_policy_host_logic_matches_step_oracle():
    run_policy_vs_oracle(CpuOps(), "cpu", SMALL)


def test_policy_forward_text_matches_oracle_generation():
    """`policy.net(...) -> List[str]` (VLN-POL:329) against the oracle's greedy generation, two steps so that the generated
    text of step 1 is part of step 2's prompt through the action history (VLN-POL:466-468)."""
    cfg, B = SMALL, 2
    sd = synth_policy_weights(cfg, seed=0)
    net = Dynam3D_VLN(cfg, sd, device="cpu", batch_size=B, ops=CpuOps(), max_steps=3)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    orc = StepOracle(sd, cfg.vit, cfg.llm, B, SyntheticTokenizer(cfg.llm.vocab))
    ep = SyntheticEpisodes(B, seed=5, image_hw=224, depth_hw=224)
    instr = [INSTRUCTION_64] * B
    for _ in range(2):
        fr = ep.next()
        pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
        got = net({"rgb": torch.from_numpy(fr.rgb), "depth": torch.from_numpy(fr.depth)}, instr, pos, hd, patch_segm=fr.patch_segm, max_new_tokens=3)
        ref = orc.generate(fr.rgb, fr.depth, instr, pos, hd, fr.patch_segm, max_new_tokens=3)
        assert got == ref, (got, ref)
        assert [h[-1] for h in net.feature_fields.history_actions] == [h[-1] for h in orc.history]
        assert net.convert_text_to_action(got) == net.convert_text_to_action(ref)



def test_synthetic_tokenizer_fast_split_equals_reference_arithmetic():
    """`SyntheticTokenizer.split_prompt` (no regex over the placeholders) == the generic `ids[:2]` / `ids[n+2:]` of the full prompt
    (VLN-POL:436-438, 456), with and without BOS, head shorter / longer than two tokens."""
    import itertools
    from dynam3d_amd.adapters import PromptTokenizer
    for bos in (True, False):
        t = SyntheticTokenizer(32064, add_bos=bos)
        for head, n, tail in itertools.product(["<|user|>\n", "", "<|user|>\n a b c d"], [0, 1, 2, 3, 700], ["\nInstruction:\nwalk to x\n<|end|>\n", ""]):
            assert t.split_prompt(head, n, tail) == PromptTokenizer.split_prompt(t, head, n, tail), (bos, head, n, tail)
