"""Differential driver: the device planner (csrc/ff_plan.h through `ops.ffdev_*`) against the host state machine (csrc/ff_state.cpp) on
RANDOM decision streams -- random frustum hits, segmentations, KNN tables, merge logits, zone cells --, many more deletions, id
recyclings, zone deaths and multi-segment merges than the golden trajectories contain.  After every call the two must agree on every
output (new slots, merged instances and their member rows in push order, touched zones, modes, rows, members) and on the exported
dictionaries (owner, members, zone snapshots, zone keys in dict order, live ids in dict order).

`ops` decides where the planner runs: tests/cpu_ops.CpuOps = the planner source compiled over host arrays (CPU suite); ops.HipOps = the
HIP kernels (GPU suite)."""
import ctypes as C

import numpy as np
import torch

from dynam3d_amd._ffstate import FFState
from dynam3d_amd.ff_plan import REPORT_WORDS, V_KEFF, V_NDIRTY, V_NTOUCHED, DevicePlanner
from dynam3d_amd.ops import FTS, Pools


def _t(a, device, dtype=torch.int32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def run_random(ops, host_lib: C.CDLL, device, compat: str, seed: int, B: int = 2, steps: int = 25, P: int = 48, K: int = 2, k_max: int = 2,
               hit_rate: float = 0.25, max_seg: int = 9):
    rng = np.random.default_rng(seed)
    host = FFState(host_lib, compat, P, K)
    host.reset(B)
    tomb = (-5000, -5000, -5000)
    host.set_tomb_cell(tomb)
    R, M, Z = P * (steps + 1), 256, 256
    dev = DevicePlanner(compat, P, K, device)
    dev.reset(B, R, M, Z, tomb)
    pools = Pools.allocate(B, R, M, Z, device)
    pools.inst_pos.fill_(1.0); pools.inst_fts.fill_(1.0); pools.zone_pos.fill_(1.0); pools.zone_fts.fill_(1.0)
    slot = _t(np.arange(B), device)
    stats = dict(dead_inst=0, dead_zone=0, merges=0, multi=0, shrink=0, recycled=0)

    def compare_state(tag):
        for e in range(B):
            a, b = host.export(e), dev.export(e)
            assert a["owner"] == b["owner"], (tag, e, "owner")
            assert list(a["members"]) == list(b["members"]), (tag, e, "instance dict order", list(a["members"]), list(b["members"]))
            for i in a["members"]:
                assert np.array_equal(a["members"][i], b["members"][i]), (tag, e, "members", i, a["members"][i], b["members"][i])
            assert list(a["zmembers"]) == list(b["zmembers"]), (tag, e, "zone dict order")
            for z in a["zmembers"]:
                assert np.array_equal(a["zmembers"][z], b["zmembers"][z]), (tag, e, "zone snapshot", z, a["zmembers"][z], b["zmembers"][z])
            assert list(a["zkey_cells"].items()) == list(b["zkey_cells"].items()), (tag, e, "zone keys")
            h = dev.header(e)
            for which, word in ((host.ROWS, 0), (host.OWNED, 1), (host.SLOTS, 2), (host.LIVE, 3), (host.ZROWS, 4), (host.ZLIVE, 5)):
                assert host.count(e, which) == int(h[word]), (tag, e, "count", which, host.count(e, which), int(h[word]))
        mx = max(1, max(host.count(e, host.SLOTS) for e in range(B)), max(host.count(e, host.ZROWS) for e in range(B)) + 8)
        ii, ni, zi, nz = ops.ffdev_live_ids(dev, slot, mx)
        ii, ni, zi, nz = ii.cpu().numpy(), ni.cpu().numpy(), zi.cpu().numpy(), nz.cpu().numpy()
        for e in range(B):
            a, b = host.live_ids(e)
            assert np.array_equal(a, ii[e, :ni[e]]) and np.array_equal(b, zi[e, :nz[e]]), (tag, e, "live ids")

    for step in range(steps):
        # ---- deletion pass -----------------------------------------------------------------------------------------------
        n_rows = [host.count(e, host.ROWS) for e in range(B)]
        mx = max(n_rows)
        if mx:
            hits = np.zeros((B, mx), np.int32)
            n_hits = np.zeros(B, np.int32)
            dead = []
            for e in range(B):
                rate = hit_rate * rng.uniform(0, 2) if rng.uniform() < 0.8 else 0.95        # sometimes nearly everything goes
                h = rng.permutation(np.nonzero(rng.uniform(size=n_rows[e]) < rate)[0]).astype(np.int32)
                hits[e, :len(h)], n_hits[e] = h, len(h)
                dead.append(host.apply_hits(e, h))
                host.end_view(e)
            ops.ffdev_apply_hits(dev, slot, _t(hits, device), _t(n_hits, device), pools)
            ipos, zpos = pools.inst_pos.cpu().numpy(), pools.zone_pos.cpu().numpy()
            for e in range(B):
                di, dz = dead[e]
                stats["dead_inst"] += len(di); stats["dead_zone"] += len(dz)
                assert np.all(ipos[e, di] == -10000.0) and np.all(zpos[e, dz] == -10000.0), (step, e, "tomb-stones")
                assert np.all(pools.inst_fts[e, torch.from_numpy(di).long()].cpu().numpy() == 0.0)
                keep = np.setdiff1d(np.arange(M), di)
                assert np.all(ipos[e, keep] != -10000.0), (step, e, "an instance that lives was tomb-stoned")
                pools.inst_pos[e, torch.from_numpy(di).long()] = 1.0                                # (re-arm the marker)
                pools.inst_fts[e, torch.from_numpy(di).long()] = 1.0
                pools.zone_pos[e, torch.from_numpy(dz).long()] = 1.0
            compare_state(("after hits", step))
        # ---- one view ------------------------------------------------------------------------------------------------------
        hb = [host.begin_view(e) for e in range(B)]
        k0_d, tree_d = ops.ffdev_begin_view(dev, slot)
        k0_h = [k if t else 0 for (_, k, t) in hb]
        assert k0_d.cpu().tolist() == k0_h, (step, "k0", k0_d.cpu().tolist(), k0_h)
        n_seg = rng.integers(1, max_seg + 1, size=B)
        n_max = int(n_seg.max())
        segm = np.zeros((B, P), np.int32)
        d2 = np.full((B, n_max, k_max), np.inf, np.float32)
        idx = np.full((B, n_max, k_max), -1, np.int32)
        logits = rng.standard_normal((B, n_max, k_max, 2)).astype(np.float32)
        cells = rng.integers(-1, 2, size=(B * n_max, 3)).astype(np.int32)
        for e in range(B):
            lab = rng.integers(0, n_seg[e], size=P)
            lab[:n_seg[e]] = np.arange(n_seg[e])                                # dense labels
            segm[e] = rng.permutation(lab)
            live = host.live_ids(e)[0]
            k = k0_h[e]
            if k:
                for s in range(n_seg[e]):
                    idx[e, s, :k] = rng.choice(live, size=k, replace=False)
                    d2[e, s, :k] = np.sort(rng.uniform(0, 4, size=k)).astype(np.float32)
                if k > 1 and rng.uniform() < 0.2:                               # the last proposal column hit tomb-stones: k shrinks (VLN-FF:607-610)
                    d2[e, :n_seg[e], k - 1] = 3e8
                    stats["shrink"] += 1
            if rng.uniform() < 0.3:
                logits[e, :, :, 1] += 2.0                                       # a merge-happy frame
        order = np.argsort(segm, axis=1, kind="stable").astype(np.int32)
        counts = np.stack([np.bincount(segm[e], minlength=n_max) for e in range(B)])
        seg_off = np.concatenate([np.zeros((B, 1), np.int64), np.cumsum(counts, 1)], 1)
        rep = torch.zeros((B * REPORT_WORDS,), dtype=torch.int32, device=device)
        rows_stride = max(n_rows) + P
        n_seg_d, cells_d = _t(n_seg, device), _t(cells, device)
        seg_slot, dirty_inst, dirty_off, dirty_rows = ops.ffdev_plan_merge(dev, slot, _t(order, device), _t(np.take_along_axis(segm, order, 1), device),
                                                                           _t(seg_off, device), n_seg_d, n_max, k_max, k0_d, _t(d2, device, torch.float32),
                                                                           _t(idx, device), _t(logits, device, torch.float32), cells_d, rows_stride, rep)
        G_ub = B * n_max
        goff = torch.zeros((G_ub + 1,), dtype=torch.int32, device=device)
        tot = torch.zeros((2 + 2 * B,), dtype=torch.int32, device=device)
        ts, tr, gs, gi = ops.ffdev_flatten_merge(slot, n_max, dirty_inst, dirty_off, dirty_rows, rep, goff, tot)
        rep_h = rep.cpu().numpy().reshape(B, REPORT_WORDS).copy()
        seg_slot_h, dirty_h, doff_h, drows_h = seg_slot.cpu().numpy(), dirty_inst.cpu().numpy(), dirty_off.cpu().numpy(), dirty_rows.cpu().numpy()
        goff_h, tot_h, ts_h, tr_h, gs_h, gi_h = (x.cpu().numpy() for x in (goff, tot, ts, tr, gs, gi))
        host_plans, g = [], 0
        merged_cells = np.zeros((G_ub, 3), np.int32)
        for e in range(B):
            n = int(n_seg[e])
            slots_before = host.count(e, host.SLOTS)
            keff, sslot, dirty, doff, drows = host.plan_merge(e, segm[e], n, k0_h[e], k_max, d2[e, :n], idx[e, :n], logits[e, :n], cells[e * n_max:e * n_max + n])
            assert keff == rep_h[e, V_KEFF], (step, e, "k_eff", keff, rep_h[e, V_KEFF])
            assert np.array_equal(sslot, seg_slot_h[e, :n]), (step, e, "new slots", sslot, seg_slot_h[e, :n])
            nd = len(dirty)
            assert nd == rep_h[e, V_NDIRTY] and np.array_equal(dirty, dirty_h[e, :nd]), (step, e, "merged instances", dirty, dirty_h[e, :rep_h[e, V_NDIRTY]])
            assert np.array_equal(doff, doff_h[e, :nd + 1]) and np.array_equal(drows, drows_h[e, :doff[-1] if nd else 0]), (step, e, "member rows")
            for i in range(nd):                                                     # ... and the flat tables of the float kernels
                assert gs_h[g] == e and gi_h[g] == dirty[i]
                a, b = goff_h[g], goff_h[g + 1]
                assert np.array_equal(tr_h[a:b], drows[doff[i]:doff[i + 1]]) and np.all(ts_h[a:b] == e)
                g += 1
            stats["merges"] += nd
            stats["multi"] += int((sslot < 0).sum() - nd)
            stats["recycled"] += len([s for s in sslot if 0 <= s < slots_before])      # a dead instance's slot re-opened (lowest unused id)
            dc = rng.integers(-1, 2, size=(nd, 3)).astype(np.int32)
            merged_cells[g - nd:g] = dc
            host_plans.append(host.plan_zones(e, dc, n))
        assert tot_h[0] == g and np.all(gi_h[g:] == -1) and np.all(goff_h[g:] == tot_h[1])
        zone_row, zone_mode, zone_off, zone_mem = ops.ffdev_plan_zones(dev, slot, dirty_inst, _t(merged_cells, device), cells_d, n_seg_d, n_max, M, rep)
        zgoff = torch.zeros((G_ub + 1,), dtype=torch.int32, device=device)
        ztot = torch.zeros((2 + 2 * B,), dtype=torch.int32, device=device)
        zts, zti, zmode, zgs, zgr = ops.ffdev_flatten_zones(slot, n_max, zone_row, zone_mode, zone_off, zone_mem, rep, zgoff, ztot)
        rep_h = rep.cpu().numpy().reshape(B, REPORT_WORDS)
        dev.take_report_envs(range(B), rep_h)
        zr_h, zm_h, zo_h, zmem_h = zone_row.cpu().numpy(), zone_mode.cpu().numpy(), zone_off.cpu().numpy(), zone_mem.cpu().numpy()
        zgoff_h, ztot_h, zti_h, zmode_h, zgs_h, zgr_h = (x.cpu().numpy() for x in (zgoff, ztot, zti, zmode, zgs, zgr))
        g = 0
        for e in range(B):
            zrow, zmd, zoff, zmem = host_plans[e]
            nt = len(zrow)
            assert nt == rep_h[e, V_NTOUCHED], (step, e, "touched zones")
            assert np.array_equal(zrow, zr_h[e, :nt]) and np.array_equal(zmd, zm_h[e, :nt]), (step, e, "zone rows / modes", zrow, zr_h[e, :nt], zmd, zm_h[e, :nt])
            assert np.array_equal(zoff, zo_h[e, :nt + 1]) and np.array_equal(zmem, zmem_h[e, :zoff[-1] if nt else 0]), (step, e, "zone members")
            for t in range(nt):
                a, b = zgoff_h[g], zgoff_h[g + 1]
                assert zgs_h[g] == e and zgr_h[g] == zrow[t] and zmode_h[g] == zmd[t] and np.array_equal(zti_h[a:b], zmem[zoff[t]:zoff[t + 1]])
                g += 1
            host.end_view(e)
            dev.n_rows[e] += P
        assert ztot_h[0] == g
        compare_state(("after view", step))
    return stats
