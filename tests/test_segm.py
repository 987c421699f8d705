"""a6: get_patch_segm post-processing (VLN-FF:407-420) -- oracle vs the golden produced by the reference's own function with
FastSAM's output injected; the MaskSegmenter wrapper on the CPU test double and, on the GPU, `d3d_patch_segm_from_masks`."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from tests.golden_io import load


def _cases():
    g = load("g10_patch_segm.npz")
    return [(g[f"masks_{i}"], g[f"segm_{i}"]) for i in range(int(g["n"]))]


def test_oracle_patch_segm_matches_reference_golden():
    for masks, ref in _cases():
        assert np.array_equal(G.patch_segm_from_masks(masks), ref[0])


def _run_segmenter(ops, device):
    from dynam3d_amd.segm import MaskSegmenter
    cases = _cases()
    by_shape = {}
    for masks, ref in cases:
        by_shape.setdefault(masks.shape[1:], []).append((masks, ref))
    for shape, items in by_shape.items():                          # one batched launch per mask resolution
        table = {id(m): m for m, _ in items}
        seg = MaskSegmenter(lambda img: table[img], ops, device=device)
        out = seg([id(m) for m, _ in items]).cpu().numpy()
        for j, (_, ref) in enumerate(items):
            assert np.array_equal(out[j], ref[0]), shape

    def boom(img):
        raise RuntimeError("segmenter failed")
    assert not MaskSegmenter(boom, ops, device=device)([0, 1]).any()           # VLN-FF:424-426


def test_mask_segmenter_cpu_double():
    from tests.cpu_ops import CpuOps
    _run_segmenter(CpuOps(), "cpu")


@pytest.mark.gpu
def test_patch_segm_kernel_matches_reference_golden():
    from dynam3d_amd.ops import HipOps
    _run_segmenter(HipOps(), "cuda")
    # many masks, ragged batch incl. an image without masks
    ops = HipOps()
    rng = np.random.default_rng(5)
    sets = [(rng.random((n, 96, 128)) < 0.02).astype(np.uint8) for n in (700, 0, 33, 1)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).tolist()
    segm, n_seg = ops.patch_segm_from_masks(torch.from_numpy(np.concatenate(sets)).cuda(), off, 24, 24)
    for i, s in enumerate(sets):
        ref = G.patch_segm_from_masks(s)
        assert np.array_equal(segm[i].cpu().numpy(), ref) and int(n_seg[i]) == int(ref.max()) + 1


def _batch_case():
    """g10b: ONE call of the reference's get_patch_segm on B * V = 8 images of 336 x 336, 40-120 overlapping FastSAM-shaped masks each, one
    image whose segmenter call fails (VLN-FF:424-426)."""
    g = load("g10b_patch_segm_batch.npz")
    counts, H, W = g["counts"].tolist(), int(g["H"]), int(g["W"])
    allm = np.unpackbits(g["masks_packed"])[: sum(counts) * H * W].reshape(sum(counts), H, W)
    return counts, allm, g["segm"]


def test_oracle_patch_segm_matches_reference_golden_batch_of_8():
    counts, allm, ref = _batch_case()
    off = np.concatenate([[0], np.cumsum(counts)])
    for i, n in enumerate(counts):
        assert np.array_equal(G.patch_segm_from_masks(allm[off[i]:off[i + 1]].reshape(n, *allm.shape[1:])), ref[i]), i
    assert not ref[3].any()                                           # the failed image: the reference's `except` branch returns zeros


@pytest.mark.gpu
def test_patch_segm_kernel_batch_of_8_at_336_matches_reference_golden():
    """SURVEY.md 8 f-3 contract at the step's own size, on the GPU: the adapter's whole batch (8 images, 458 masks of 336 x 336, one image
    without masks) in ONE d3d_patch_segm_from_masks launch against the label maps the reference's own function returned."""
    from dynam3d_amd.ops import HipOps
    from dynam3d_amd.segm import MaskSegmenter
    counts, allm, ref = _batch_case()
    off = np.concatenate([[0], np.cumsum(counts)]).tolist()
    ops = HipOps()
    segm, n_seg = ops.patch_segm_from_masks(torch.from_numpy(allm).cuda().contiguous(), off, 24, 24)
    assert np.array_equal(segm.cpu().numpy(), ref)
    assert n_seg.cpu().tolist() == [int(ref[i].max()) + 1 for i in range(len(counts))]
    # ... and through the adapter object the policy holds (a segmenter callable per image; the failing one raises like FastSAM's wrapper)
    stacks = {i: allm[off[i]:off[i + 1]].astype(np.float32) for i in range(len(counts))}

    def fastsam_like(i):
        if counts[i] == 0:
            raise RuntimeError("FastSAM error")
        return stacks[i]
    out = MaskSegmenter(fastsam_like, ops, device="cuda")(list(range(len(counts))))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_feature_fields_uses_configured_segmenter():
    """`update_feature_fields(..., batch_image)` without an explicit patch_segm goes through `get_patch_segm` (VLN-FF:504-506) ->
    the configured MaskSegmenter; the result must equal the run that was handed the same labels directly."""
    from dynam3d_amd.feature_fields import Feature_Fields
    from dynam3d_amd.segm import MaskSegmenter
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    from tests.cpu_ops import CpuOps
    from tests.golden_io import TRAJ_CASES, traj_inputs
    ops = CpuOps()
    sd = synth_state_dict(ff_param_spec(), seed=0)
    inp = next(iter(traj_inputs(dict(TRAJ_CASES["walk"], steps=1))))
    B = len(inp["positions"])
    segm = inp["patch_segm"].reshape(B, 24, 24)

    def masks_of(img_index):                                       # one {0,1} mask per label, at 48x48 (nearest-resized back to 24x24)
        lab = np.kron(segm[img_index], np.ones((2, 2), np.int64))
        return np.stack([(lab == k).astype(np.float32) for k in range(int(lab.max()) + 1)])

    a = Feature_Fields(B, "cpu", sd, ops=ops, segmenter=MaskSegmenter(masks_of, ops, device="cpu"))
    b = Feature_Fields(B, "cpu", sd, ops=ops)
    for ff, kw in ((a, dict(batch_image=list(range(B)))), (b, dict(patch_segm=inp["patch_segm"]))):
        ff.initialize_camera_setting(90.0, 90.0)
        ff.update_feature_fields(inp["depth24"], inp["grid"], batch_position=inp["positions"], batch_heading=inp["headings"], **kw)
    for e in range(B):
        ea, eb = a.export_env(e), b.export_env(e)
        assert ea["owner"] == eb["owner"] and np.array_equal(ea["ifts"], eb["ifts"]) and np.array_equal(ea["ipos"], eb["ipos"])
