"""CPU: the boundary adapters INTEGRATION.md names.

  * HFTokenizerAdapter on a REAL `transformers` tokenizer (built locally with `tokenizers`: the llava-phi-3 files are not
    available offline): the ids are the tokenizer's own, and the prompt splice is the reference's index arithmetic
    (VLN-POL:436-438, 456) evaluated on them -- with and without a BOS-adding post-processor, which changes WHICH tokens end
    up around the visual prefix; the policy's packed and padded prompt rows are built from exactly those ids.
  * FastSAMSegmenter with a stand-in for the reference's vendored `fastsam` package returning the masks of golden g10: the
    labels must equal what the reference's `get_patch_segm` produced from the same masks (VLN-FF:400-430)."""
import types

import numpy as np
import pytest
import torch

from tests.golden_io import load

WORDS = ("Instruction: History actions: Next action: none turn left right move steps, steps. stop walk forward past the table and "
         "wait near door go to kitchen then 1 2 3 4 5").split()
SPECIALS = ["<|user|>", "<|end|>", "<|assistant|>", "<image>", "<|endoftext|>"]


def _build_tokenizer_dir(tmp_path, add_bos: bool):
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Regex, Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "\n": 2}
    for w in WORDS:
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(r"\n|[^\s]+"), behavior="removed", invert=True)
    tok.add_special_tokens(SPECIALS)
    if add_bos:
        tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", unk_token="<unk>", additional_special_tokens=SPECIALS)
    d = tmp_path / ("tok_bos" if add_bos else "tok_nobos")
    fast.save_pretrained(str(d))
    return str(d), fast


@pytest.mark.parametrize("add_bos", [True, False])
def test_hf_tokenizer_adapter_ids_and_reference_splice(tmp_path, add_bos):
    from dynam3d_amd.adapters import HFTokenizerAdapter
    path, hf = _build_tokenizer_dir(tmp_path, add_bos)
    ad = HFTokenizerAdapter(path)
    text = "\nInstruction:\nwalk forward past the table\nHistory actions:\nnone\nnone\n<|end|>\n<|assistant|>\nNext action:\n"
    n_vis = 7
    full = "<|user|>\n" + "<image>" * n_vis + text                                  # VLN-POL:436
    ids = hf(full)["input_ids"]                                                     # what llava_processor(text=...) tokenises
    assert ad.encode(full) == ids
    head, tail = ad.split_prompt("<|user|>\n", n_vis, text)
    assert head == ids[:2] and tail == ids[n_vis + 2:]                              # VLN-POL:456
    img, user, nl = ad.SPECIAL["<image>"], ad.SPECIAL["<|user|>"], hf.convert_tokens_to_ids("\n")
    assert ids.count(img) == n_vis
    if add_bos:       # [<s>, <|user|>] stay in front; the newline is dropped and the LAST placeholder's embedding leads the tail
        assert head == [hf.bos_token_id, user] and tail[0] == img and tail[1] == nl
    else:             # [<|user|>, \n] stay in front; the tail starts at the newline behind the placeholders
        assert head == [user, nl] and tail[0] == nl and img not in tail
    assert "<|end|>" in ad.decode(ids) and ad.SPECIAL["<|end|>"] == hf.convert_tokens_to_ids("<|end|>")


def test_policy_prompt_rows_are_the_hf_ids(tmp_path):
    """`Dynam3D_VLN(tokenizer=HFTokenizerAdapter(...))`: every prompt = E[ids[:2]] + visual tokens + E[ids[n_vis+2:]] with the HF
    tokenizer's ids, in the padded layout and in the packed one."""
    import dataclasses
    from dynam3d_amd.adapters import HFTokenizerAdapter
    from dynam3d_amd.policy import Dynam3D_VLN, synth_policy_weights
    from dynam3d_amd.synthetic import SyntheticEpisodes
    from tests.cpu_ops import CpuOps
    from tests.test_policy_cpu import SMALL
    path, hf = _build_tokenizer_dir(tmp_path, True)
    B = 2
    net = Dynam3D_VLN(SMALL, synth_policy_weights(SMALL, 0), device="cpu", batch_size=B, ops=CpuOps(), tokenizer=HFTokenizerAdapter(path), max_steps=2)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    fr = SyntheticEpisodes(B, seed=3, image_hw=224, depth_hw=224).next()
    obs = {"rgb": torch.from_numpy(fr.rgb), "depth": torch.from_numpy(fr.depth)}
    instr = ["walk forward past the table and wait near the door", "go to the kitchen then stop"]
    pos, hd = [p.tolist() for p in fr.positions], list(fr.headings)
    rows, lengths = net.build_inputs(obs, instr, pos, hd, patch_segm=fr.patch_segm, return_rows=True)
    E = net.llm.embed_w.float()
    for b in range(B):
        n_vis = 576 + net.last_counts["Ni"][b] + net.last_counts["Nz"][b]
        ids = hf("<|user|>\n" + "<image>" * n_vis + net._prompt_text(b, instr))["input_ids"]
        assert lengths[b] == len(ids)                                                # one row per token of the reference's prompt
        assert torch.equal(rows[b][:2], E[ids[:2]]) and torch.equal(rows[b][2 + n_vis:], E[ids[n_vis + 2:]])
    # the packed assembly (benchmark path) lays the same rows back to back
    net2 = Dynam3D_VLN(SMALL, synth_policy_weights(SMALL, 0), device="cpu", batch_size=B, ops=CpuOps(), tokenizer=HFTokenizerAdapter(path), max_steps=2)
    net2.feature_fields.initialize_camera_setting(90.0, 90.0)
    x, lens2 = net2.build_inputs(obs, instr, pos, hd, patch_segm=fr.patch_segm, return_rows="packed")
    assert lens2 == lengths and torch.allclose(x[:sum(lengths)], torch.cat(rows), atol=0, rtol=0)


def test_fastsam_segmenter_equals_reference_get_patch_segm():
    from dynam3d_amd.adapters import FastSAMSegmenter
    from tests.cpu_ops import CpuOps
    g = load("g10_patch_segm.npz")
    cases = [(g[f"masks_{i}"], g[f"segm_{i}"]) for i in range(int(g["n"]))]
    calls = []

    class FakeFastSAM:                                              # stands in for vlnce_baselines.models.fastsam.FastSAM
        def __init__(self, weights):
            self.weights = weights

        def __call__(self, image, device=None, **kw):
            calls.append(kw)
            if image == "broken":
                raise RuntimeError("no detections")                  # VLN-FF:424: "FastSAM error, skip..."
            return image

    class FakePrompt:
        def __init__(self, image, results, device=None):
            self.i = results

        def everything_prompt(self):
            return torch.from_numpy(cases[self.i][0])

    mod = types.SimpleNamespace(FastSAM=FakeFastSAM, FastSAMPrompt=FakePrompt)
    seg = FastSAMSegmenter("FastSAM.pt", ops=CpuOps(), device="cpu", fastsam_module=mod)
    by_shape = {}
    for i, (m, _) in enumerate(cases):
        by_shape.setdefault(m.shape[1:], []).append(i)
    for idx in by_shape.values():
        out = seg(idx).cpu().numpy()
        for j, i in enumerate(idx):
            assert np.array_equal(out[j], cases[i][1][0])
    assert calls[0] == dict(retina_masks=True, imgsz=(576, 576), conf=0.4, iou=0.8)      # the reference's call (VLN-FF:401, 407)
    assert not seg(["broken"]).any()
