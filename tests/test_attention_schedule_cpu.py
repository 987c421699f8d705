"""CPU: the host-built workgroup table of the scheduled causal attention launch (hip_dense.HipDense.attention_schedule)."""
import numpy as np


def test_schedule_covers_every_block_once_heaviest_first_on_its_xcd():
    from dynam3d_amd.hip_dense import HipDense
    lens, H = [730, 784, 734, 816, 710, 720, 784, 821], 32
    t = HipDense.attention_schedule(lens, H)
    b, h, q = t >> 20, (t >> 8) & 0xfff, t & 0xff
    nqb = [(n + 127) // 128 for n in lens]
    assert len(t) == sum(nqb) * H and len(set(t.tolist())) == len(t)
    assert all(q[i] < nqb[b[i]] for i in range(len(t)))
    assert np.all(np.diff(q) <= 0)                                    # work = 2 (q + 1) key tiles: non-increasing
    assert np.all((b * H + h) % 8 == np.arange(len(t)) % 8)           # entry p runs on XCD p % 8 = the XCD of its (sequence, head)
    # makespan model (unit = one 128 x 64 tile, 512 slots, list scheduling in table order): the table packs the batch into ~26 tile-times,
    # the paired launch (blocks i and n-1-i in one workgroup, two rounds) needs 32
    import heapq
    slots = [0] * 512
    heapq.heapify(slots)
    for w in (2 * (q + 1)).tolist():
        heapq.heappush(slots, heapq.heappop(slots) + w)
    assert max(slots) <= 27


def test_schedule_small_heads():
    from dynam3d_amd.hip_dense import HipDense
    t = HipDense.attention_schedule([1, 129, 300], 4)
    assert len(t) == (1 + 2 + 3) * 4 and len(set(t.tolist())) == len(t)
