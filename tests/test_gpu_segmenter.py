"""GPU: the on-device segmenter behind `Feature_Fields.get_patch_segm` (VLN-FF:400-430).

FastSAM's weights are not available offline, so the MASKS themselves cannot be compared with the reference's; the contract the memory
update depends on can:
  * `d3d_segment_slic` == its float32 restatement (oracle/segment_ref.py) BIT FOR BIT: labels and masks, several sizes / seed grids,
  * masks -> label map goes through the SAME `d3d_patch_segm_from_masks` kernel that g10 pins against the reference's own
    `get_patch_segm` post-processing: dense labels 0 .. n-1, every label present, int64 (N,1,24,24) -- VLN-FF:411-420,
  * on a frame made of flat colour blocks the segmenter recovers the blocks,
  * end to end: `update_feature_fields(patch_segm=None, batch_image=rgb)` and the full policy step run without a segmentation input."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dynam3d_amd.ops import HipOps
    return HipOps()


@pytest.mark.parametrize("hw,seeds,iters", [((224, 224), (4, 4), 5), ((96, 130), (3, 5), 3), ((64, 64), (8, 8), 0)])
def test_slic_kernel_bit_exact_vs_oracle(ops, hw, seeds, iters):
    from dynam3d_amd.segm import SlicSegmenter
    from oracle.segment_ref import segment_slic
    rng = np.random.default_rng(3)
    H, W = hw
    # smooth random colour fields + noise: real boundaries, many near-ties
    base = rng.integers(0, 256, (2, 6, 6, 3)).astype(np.float32)
    img = np.stack([np.kron(b, np.ones((H // 6 + 1, W // 6 + 1, 1)))[:H, :W] for b in base])
    img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)
    seg = SlicSegmenter(ops, seeds=seeds, iters=iters, compactness=20.0)
    masks, labels = seg.masks(torch.from_numpy(img), return_labels=True)
    for i in range(2):
        lab, mk = segment_slic(img[i], seeds[1], seeds[0], iters, 20.0)
        assert np.array_equal(labels[i].cpu().numpy(), lab), (i, float((labels[i].cpu().numpy() != lab).mean()))
        assert np.array_equal(masks[i].cpu().numpy(), mk)
    assert int(masks.sum()) == 2 * H * W                                           # the masks partition every frame


def test_patch_segm_contract_and_block_recovery(ops):
    from dynam3d_amd.segm import SlicSegmenter
    rng = np.random.default_rng(4)
    # 2 x 2 flat colour blocks + mild noise: 4 segments must come back as 4 quadrants of the 24 x 24 label map
    cols = np.array([[200, 30, 30], [30, 200, 30], [30, 30, 200], [220, 220, 40]], np.float32)
    img = np.zeros((3, 224, 224, 3), np.float32)
    for q in range(4):
        img[:, (q // 2) * 112:(q // 2 + 1) * 112, (q % 2) * 112:(q % 2 + 1) * 112] = cols[q]
    img = np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)
    seg = SlicSegmenter(ops, seeds=(2, 2), iters=5)
    ps = seg(torch.from_numpy(img))
    assert ps.dtype == torch.int64 and tuple(ps.shape) == (3, 1, 24, 24)
    for i in range(3):
        m = ps[i, 0].cpu().numpy()
        assert sorted(np.unique(m).tolist()) == [0, 1, 2, 3]                       # dense relabel, every label present (VLN-FF:416-420)
        for q in range(4):
            blk = m[(q // 2) * 12:(q // 2 + 1) * 12, (q % 2) * 12:(q % 2 + 1) * 12]
            assert (blk == blk[0, 0]).all()
    # random frames with the default 4 x 4 seeds: still dense, deterministic
    rnd = torch.from_numpy(rng.integers(0, 256, (4, 224, 224, 3), dtype=np.uint8))
    seg16 = SlicSegmenter(ops)
    a, b = seg16(rnd), seg16(rnd)
    assert torch.equal(a, b)
    for i in range(4):
        u = np.unique(a[i].cpu().numpy())
        assert u.tolist() == list(range(len(u))) and 1 <= len(u) <= 16


def test_memory_update_and_policy_step_without_segmentation_input(ops):
    """RGB-D in, tokens / logits out: `patch_segm` comes from the device segmenter through `get_patch_segm` (VLN-FF:513-515)."""
    from dynam3d_amd.policy import Dynam3D_VLN
    from dynam3d_amd.segm import SlicSegmenter
    from dynam3d_amd.synthetic import INSTRUCTION_64, SyntheticEpisodes
    from tests.test_policy_cpu import MID
    B = 2
    net = Dynam3D_VLN(MID, seed=0, device="cuda", batch_size=B, segmenter=SlicSegmenter(ops))
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    ep = SyntheticEpisodes(B, seed=2)
    for _ in range(2):
        fr = ep.next()
        obs = dict(rgb=torch.from_numpy(fr.rgb).cuda(), depth=torch.from_numpy(fr.depth).cuda())
        lo = net.forward_logits(obs, [INSTRUCTION_64] * B, [p.tolist() for p in fr.positions], list(fr.headings))      # no patch_segm=
    assert torch.isfinite(lo).all() and lo.shape[0] == B
    st = net.feature_fields.state
    assert all(st.count(e, st.LIVE) >= 1 for e in range(B))
