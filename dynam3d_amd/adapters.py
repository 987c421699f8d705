"""Adapters to the reference's third-party front ends that are NOT on the hot path but sit on its boundary:

  * `HFTokenizerAdapter`  -- the llava-phi-3-mini tokenizer the reference reaches through `AutoProcessor.from_pretrained(
                             "xtuner/llava-phi-3-mini-hf")` (VLN-POL:113-131, 436-438, 463-465), wrapped into the small
                             `encode / decode / SPECIAL / split_prompt` surface `policy.Dynam3D_VLN` uses.
  * `FastSAMSegmenter`    -- `Feature_Fields.get_patch_segm`'s FastSAM call (VLN-FF:400-430) as the `mask_fn` of
                             `segm.MaskSegmenter`: the network stays the reference's (ultralytics, vendored there, weights
                             `FastSAM.pt`), everything after it -- label image, nearest resize, dense relabel -- is the HIP kernel.

Neither the tokenizer files nor FastSAM.pt / ultralytics exist on the build or GPU machines (no network), so both adapters
import their dependency lazily and fail with a clear message; tests drive them with a locally built tokenizer / a stub network.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch

IMAGE_TOKEN = "<image>"


class PromptTokenizer:
    """What `policy.Dynam3D_VLN` needs from a tokenizer.  `split_prompt` is the reference's splice arithmetic (VLN-POL:436-461):
    the prompt is tokenised WITH one "<image>" placeholder per visual token,

        ids = tok("<|user|>\\n" + "<image>" * n_visual + text)            (VLN-POL:436-438)
        embeds = [ E[ids[:2]], visual tokens, E[ids[n_visual + 2:]] ]       (VLN-POL:456)

    so which ids end up in front of / behind the visual prefix depends on the tokenizer (a BOS-adding tokenizer keeps
    [<s>, <|user|>] in front and leaves the LAST placeholder's embedding in the tail; one without BOS keeps [<|user|>, \\n] and the
    tail starts at the newline).  The arithmetic is reproduced literally on whatever the tokenizer returns."""
    SPECIAL = {}

    def encode(self, text: str) -> List[int]:
        raise NotImplementedError

    def decode(self, ids: Sequence[int]) -> str:
        raise NotImplementedError

    def split_prompt(self, head: str, n_visual: int, tail: str) -> Tuple[List[int], List[int]]:
        ids = self.encode(head + IMAGE_TOKEN * n_visual + tail)
        return list(ids[:2]), list(ids[n_visual + 2:])


class HFTokenizerAdapter(PromptTokenizer):
    """`HFTokenizerAdapter("xtuner/llava-phi-3-mini-hf")` or a local directory with the tokenizer files.  Encoding is the
    tokenizer's own `__call__` (its BOS policy included), which is what `llava_processor(text=...)` runs for the text side."""

    def __init__(self, name_or_path: str, **kw):
        try:
            from transformers import AutoTokenizer
        except ImportError as e:                                   # pragma: no cover
            raise RuntimeError("HFTokenizerAdapter needs `transformers` (the reference pins 4.46.0)") from e
        self.tok = AutoTokenizer.from_pretrained(name_or_path, **kw)
        names = ["<|user|>", "<|end|>", "<|assistant|>", IMAGE_TOKEN, "<|endoftext|>"]
        self.SPECIAL = {}
        for n in names:
            i = self.tok.convert_tokens_to_ids(n)
            if i is not None and i != self.tok.unk_token_id:
                self.SPECIAL[n] = int(i)
        missing = [n for n in ("<|end|>", IMAGE_TOKEN) if n not in self.SPECIAL]
        if missing:
            raise ValueError(f"tokenizer at {name_or_path!r} has no id for {missing}: not a llava-phi-3 tokenizer")

    def encode(self, text: str) -> List[int]:
        return [int(i) for i in self.tok(text)["input_ids"]]

    def decode(self, ids: Sequence[int]) -> str:
        return self.tok.decode([int(i) for i in ids], skip_special_tokens=False)            # VLN-POL:464


class FastSAMSegmenter:
    """segmenter(batch_image) -> (N,1,24,24) dense int64 labels, FastSAM as the mask generator (VLN-FF:400-430).

    `weights` = path of FastSAM.pt; `fastsam_module` = the reference's vendored package (`vlnce_baselines.models.fastsam`: FastSAM,
    FastSAMPrompt), imported from the reference tree when not given.  imgsz / conf / iou default to the reference's call."""

    def __init__(self, weights: str = "FastSAM.pt", ops=None, device="cuda", fastsam_module=None, imgsz=(576, 576), conf: float = 0.4,
                 iou: float = 0.8, grid_hw=(24, 24)):
        if fastsam_module is None:
            try:
                from vlnce_baselines.models import fastsam as fastsam_module        # the reference's vendored wrapper
            except ImportError as e:
                raise RuntimeError("FastSAMSegmenter: put the reference's Dynam3D_VLN directory (vlnce_baselines.models.fastsam + ultralytics) "
                                   "on sys.path, or pass fastsam_module=") from e
        from .segm import MaskSegmenter
        if ops is None:
            from .ops import HipOps
            ops = HipOps()
        self._mod = fastsam_module
        self.model = fastsam_module.FastSAM(weights)
        self.device, self.kw = device, dict(retina_masks=True, imgsz=imgsz, conf=conf, iou=iou)
        self._seg = MaskSegmenter(self._masks, ops, grid_hw=grid_hw, device=device)

    def _masks(self, image, **kw):
        """VLN-FF:407-410; an exception (FastSAM found nothing, ...) is caught by MaskSegmenter like the reference's `except`."""
        res = self.model(image, device=self.device, **{**self.kw, **kw})
        return self._mod.FastSAMPrompt(image, res, device=self.device).everything_prompt()

    def __call__(self, batch_image, **kw) -> torch.Tensor:
        return self._seg(batch_image, **kw)
