"""CPU: the stand-in segmenter's oracle (oracle/segment_ref.py) on a frame of flat colour blocks, and the mask -> dense label contract
through the numpy emulation of `d3d_patch_segm_from_masks` (the same routine g10 pins against the reference's `get_patch_segm`)."""
import numpy as np

from oracle.segment_ref import segment_slic


def test_oracle_segmenter_recovers_colour_blocks_and_partitions_the_frame():
    rng = np.random.default_rng(1)
    cols = np.array([[200, 30, 30], [30, 200, 30], [30, 30, 200], [220, 220, 40]], np.float32)
    img = np.zeros((64, 64, 3), np.float32)
    for q in range(4):
        img[(q // 2) * 32:(q // 2 + 1) * 32, (q % 2) * 32:(q % 2 + 1) * 32] = cols[q]
    img = np.clip(img + rng.normal(0, 3, img.shape), 0, 255).astype(np.uint8)
    lab, masks = segment_slic(img, 2, 2, 4, 20.0)
    assert masks.shape == (4, 64, 64) and masks.sum() == 64 * 64 and np.array_equal(masks.argmax(0), lab)
    for q in range(4):
        blk = lab[(q // 2) * 32:(q // 2 + 1) * 32, (q % 2) * 32:(q % 2 + 1) * 32]
        assert (blk == blk[0, 0]).all()
    assert len(np.unique(lab)) == 4
    # deterministic
    assert np.array_equal(segment_slic(img, 2, 2, 4, 20.0)[0], lab)
