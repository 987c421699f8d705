// ff_plan.h -- the patch -> instance -> zone bookkeeping as DATA-PARALLEL code over flat int32 arrays that live in device memory.
//
// Same decisions as the host state machine (ff_state.cpp, which follows Dynam3D_VLN/vlnce_baselines/models/feature_fields.py:
// deletion cascade 362-393, id allocation 433-475, new / merge bookkeeping 623-691, zone update 694-756, live-id order 825/844), but
// nothing here is a dict or a list: the memory update plans itself ON THE GPU and the host reads back one small report per view.
//
//   * dict `patch id -> instance`       owner[pid]                         (-1 = not a key)
//   * list `instance -> member patches` not stored: members(inst) = { pid_of_stamp[u] : u ascending, stamp_of_pid[pid] == u,
//                                       owner[pid] == inst }  -- the push order of the reference's lists IS the insertion stamp
//                                       (view-major, then segment-major, then patch index), so an ORDERED compaction over the stamps
//                                       reproduces every list, including the order the set encoder sees
//   * dict `instance id -> ...`         live[i], istamp[i] (dict insertion order), icnt[i] = len(members), icell[i] (its 2 m zone cell)
//   * dict `zone key -> zone id`        zlive[z], zkey[z], zkey_stamp[z], zstamp[z]
//   * list `zone -> member instances`   an edge table (edge_z, edge_i): the reference keeps the SNAPSHOT taken at the zone's last update
//                                       (stale members stay until the zone is touched again), so it has to be explicit; double-buffered,
//                                       compacted at every zone update
//
// Every function is written once against a small execution context `Cx` (par / one / compact / atomic_add / sync):
//   ff_plan_kernels.hip  runs it with one workgroup per environment (BlockCtx: ballot + LDS ordered compaction),
//   ff_plan_host.cpp     runs it with a one-lane serial context over host arrays -- only in the CPU-only test library, where the whole
//                        planner is replayed against the reference-generated golden trajectories without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FFP_HD __host__ __device__ inline
#else
#define FFP_HD inline
#endif

namespace ffplan {

// per-slot header words (int32)
enum { H_NROWS = 0, H_NOWNED, H_NSLOTS, H_NLIVE, H_NZROWS, H_NZLIVE, H_NZIDS, H_STAMP, H_HAS_TREE, H_TREE_SLOTS, H_NEDGES, H_EDGE_SEL, H_ERR,
       H_WORDS = 16 };
// per-environment report words of one view (read back by the host once per view)
enum { V_NDIRTY = 0, V_DIRTY_ROWS, V_KEFF, V_NTOUCHED, V_ZONE_MEMBERS, V_NSLOTS, V_NZROWS, V_NZIDS, V_NEDGES, V_ERR, V_NLIVE, V_NZLIVE,
       V_NOWNED, V_WORDS = 16 };
enum { ERR_PROPOSAL_NOT_LIVE = 1, ERR_ROWS_OVERFLOW = 2, ERR_EDGE_OVERFLOW = 4, ERR_SLOT_OVERFLOW = 8 };

constexpr int32_t NO_LIMIT = 0x7fffffff;

// The arrays of ONE storage slot.  Layout (see d3d_ffdev_state in include/dynam3d_hip.h): rows [3][R], inst [6][M], zone [8][Z],
// edges [2 buffers][2][E], scratch [W].
struct View {
    int32_t* hdr;
    int32_t *owner, *stamp_of_pid, *pid_of_stamp;                  // [R]
    int32_t *live, *istamp, *icnt, *icx, *icy, *icz;                // [M]
    int32_t *zlive, *zstamp, *zkey_stamp, *zcnt, *zvisit, *zkx, *zky, *zkz;   // [Z]
    int32_t* edges;                                                 // [2][2][E]
    int32_t* scratch;                                               // [W]
    int32_t R, M, Z, E, W;

    FFP_HD int32_t* edge_z(int sel) const { return edges + (int64_t)sel * 2 * E; }
    FFP_HD int32_t* edge_i(int sel) const { return edges + (int64_t)sel * 2 * E + E; }
};

struct State {     // == d3d_ffdev_state
    int32_t *hdr, *rows, *inst, *zone, *edges, *scratch;
    int32_t R, M, Z, E, W;
    int32_t compat_fixed, P, K;
    int32_t tomb[3];
};

FFP_HD View view_of(const State& s, int slot) {
    View v;
    v.R = s.R; v.M = s.M; v.Z = s.Z; v.E = s.E; v.W = s.W;
    v.hdr = s.hdr + (int64_t)slot * H_WORDS;
    int32_t* r = s.rows + (int64_t)slot * 3 * s.R;
    v.owner = r; v.stamp_of_pid = r + s.R; v.pid_of_stamp = r + 2 * (int64_t)s.R;
    int32_t* i = s.inst + (int64_t)slot * 6 * s.M;
    v.live = i; v.istamp = i + s.M; v.icnt = i + 2 * (int64_t)s.M; v.icx = i + 3 * (int64_t)s.M; v.icy = i + 4 * (int64_t)s.M; v.icz = i + 5 * (int64_t)s.M;
    int32_t* z = s.zone + (int64_t)slot * 8 * s.Z;
    v.zlive = z; v.zstamp = z + s.Z; v.zkey_stamp = z + 2 * (int64_t)s.Z; v.zcnt = z + 3 * (int64_t)s.Z; v.zvisit = z + 4 * (int64_t)s.Z;
    v.zkx = z + 5 * (int64_t)s.Z; v.zky = z + 6 * (int64_t)s.Z; v.zkz = z + 7 * (int64_t)s.Z;
    v.edges = s.edges + (int64_t)slot * 4 * s.E;
    v.scratch = s.scratch + (int64_t)slot * s.W;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the one-lane context (host arrays; also documents the contract of the workgroup context)
// ---------------------------------------------------------------------------------------------------------------------------------
struct SerialCtx {
    FFP_HD void sync() const {}
    template <class F> FFP_HD void par(int n, F f) const { for (int i = 0; i < n; ++i) f(i); }          // independent iterations
    template <class F> FFP_HD void one(F f) const { f(); }                                               // sequential section; ends with a sync
    // ordered compaction: emit(i, rank) for the first `limit` indices i in [0, n) with pred(i), rank = position among them; returns the count
    template <class P, class E> FFP_HD int compact(int n, int limit, P pred, E emit) const {
        int c = 0;
        for (int i = 0; i < n && c < limit; ++i)
            if (pred(i)) { emit(i, c); ++c; }
        return c;
    }
    FFP_HD void atomic_add(int32_t* p, int32_t v) const { *p += v; }
    FFP_HD void atomic_or(int32_t* p, int32_t v) const { *p |= v; }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// begin_view / end_view (VLN-FF:532, 243-247)
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx> FFP_HD void begin_view(Cx& cx, const View& v, int K, int32_t* k0_out, int32_t* tree_slots_out) {
    cx.one([&] {
        const int ht = v.hdr[H_HAS_TREE];
        const int nl = v.hdr[H_NLIVE];
        *k0_out = ht ? (nl < K ? nl : K) : 0;
        *tree_slots_out = ht ? v.hdr[H_TREE_SLOTS] : 0;
    });
}

FFP_HD void end_view_lane0(const View& v) {     // kd-tree rebuild: the tree covers every slot, dead ones at -10000 (VLN-FF:396, 815)
    v.hdr[H_HAS_TREE] = v.hdr[H_NSLOTS] > 0 ? 1 : 0;
    v.hdr[H_TREE_SLOTS] = v.hdr[H_NSLOTS];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// deletion cascade (VLN-FF:362-393): the frustum kernel's hit list -> patches leave their instance, emptied instances die (position
// -10000, feature 0, cell -> tomb cell), the zone snapshot of a dead instance's cell loses it, emptied zones die.  The final state does
// not depend on the order of the hits (sets), so every stage is a parallel sweep.
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx>
FFP_HD void apply_hits(Cx& cx, const View& v, const int32_t* hits, int n_hits, const int32_t* tomb, float* inst_pos, float* inst_fts,
                       float* zone_pos, float* zone_fts, int fts_dim) {
    cx.sync();
    const int n_rows = v.hdr[H_NROWS];
    cx.par(n_hits, [&](int h) {
        const int pid = hits[h];
        if (pid < 0 || pid >= n_rows) return;
        const int inst = v.owner[pid];
        if (inst < 0) return;                                      // `if patch_id not in dict: continue` (VLN-FF:365)
        v.owner[pid] = -1;
        cx.atomic_add(&v.icnt[inst], -1);
        cx.atomic_add(&v.hdr[H_NOWNED], -1);
    });
    cx.sync();
    const int ns = v.hdr[H_NSLOTS], nz = v.hdr[H_NZIDS], ne = v.hdr[H_NEDGES], sel = v.hdr[H_EDGE_SEL];
    int32_t* dead = v.scratch;                                     // [<= M] dying instances, then [<= Z] dying zones
    cx.par(nz, [&](int z) { v.zvisit[z] = 0; });
    const int nd = cx.compact(ns, NO_LIMIT, [&](int i) { return v.live[i] != 0 && v.icnt[i] == 0; }, [&](int i, int c) { dead[c] = i; });
    int32_t* ez = v.edge_z(sel);
    const int32_t* ei = v.edge_i(sel);
    cx.par(nd, [&](int c) {                                        // instance removed (VLN-FF:372-379)
        const int i = dead[c];
        v.live[i] = 0;
        const int kx = v.icx[i], ky = v.icy[i], kz = v.icz[i];
        v.icx[i] = tomb[0]; v.icy[i] = tomb[1]; v.icz[i] = tomb[2];
        for (int z = 0; z < nz; ++z) {
            if (!v.zlive[z] || v.zkx[z] != kx || v.zky[z] != ky || v.zkz[z] != kz) continue;
            for (int k = 0; k < ne; ++k)
                if (ez[k] == z && ei[k] == i) { ez[k] = -1; cx.atomic_add(&v.zcnt[z], -1); }
            v.zvisit[z] = 1;                                       // "if the snapshot is empty now, the zone goes" is checked for every visited zone
            break;
        }
    });
    cx.sync();
    const int nzd = cx.compact(nz, NO_LIMIT, [&](int z) { return v.zvisit[z] != 0 && v.zlive[z] != 0 && v.zcnt[z] == 0; },
                               [&](int z, int c) { dead[nd + c] = z; });
    cx.par(nzd, [&](int c) { v.zlive[dead[nd + c]] = 0; });        // zone removed (VLN-FF:388-393)
    cx.one([&] {
        v.hdr[H_NLIVE] -= nd;
        v.hdr[H_NZLIVE] -= nzd;
        end_view_lane0(v);                                         // the caller rebuilds the tree from the instance positions (VLN-FF:396)
    });
    cx.par(nd * 3, [&](int t) { inst_pos[(int64_t)dead[t / 3] * 3 + t % 3] = -10000.0f; });
    cx.par(nd * fts_dim, [&](int t) { inst_fts[(int64_t)dead[t / fts_dim] * fts_dim + t % fts_dim] = 0.0f; });
    cx.par(nzd * 3, [&](int t) { zone_pos[(int64_t)dead[nd + t / 3] * 3 + t % 3] = -10000.0f; });      // (the zone ID indexes the row: quirk Z1)
    cx.par(nzd * fts_dim, [&](int t) { zone_fts[(int64_t)dead[nd + t / fts_dim] * fts_dim + t % fts_dim] = 0.0f; });
    cx.sync();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// new / merge bookkeeping of one view (VLN-FF:604-691).  Inputs of this environment: the P patches sorted by (segment, patch) --
// `order[t]` = patch index, `tok_seg[t]` = its segment, `seg_off` the CSR over t --, the KNN table (d2, idx: [n_seg][k_max]), the merge
// logits ([n_seg][k_max][2]) and the new segments' zone cells ([n_seg][3]).  Outputs: seg_slot[s] = slot of the NEW instance segment s
// opens (-1 = merged), the merged ("dirty") instances in first-touch order with their full member lists (CSR), the report words.
// The reference walks the segments one after the other; here only the ranks are sequential (ordered compactions), the rest is a sweep.
// scratch use: new_pid [P] | seg_inst [P] | new_inst [P] | new_rank [P] | misc [8] | rank_seg [P].
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx>
FFP_HD void plan_merge(Cx& cx, const View& v, int compat_fixed, int P, const int32_t* order, const int32_t* tok_seg, const int32_t* seg_off,
                       int n_seg, int k0, int k_max, const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells,
                       int32_t* seg_slot, int32_t* dirty_inst, int32_t* dirty_off, int32_t* dirty_rows, int rows_cap, int32_t* vh) {
    int32_t* new_pid = v.scratch;
    int32_t* seg_inst = v.scratch + P;
    int32_t* new_inst = v.scratch + 2 * (int64_t)P;
    int32_t* new_rank = v.scratch + 3 * (int64_t)P;
    int32_t* misc = v.scratch + 4 * (int64_t)P;                   // [0] k_eff  [1] error bits
    int32_t* rank_seg = v.scratch + 4 * (int64_t)P + 8;           // segment that opens the c-th new instance
    cx.sync();
    const int row_base = v.hdr[H_NROWS];
    const int ns0 = v.hdr[H_NSLOTS];
    cx.one([&] {
        int k = v.hdr[H_HAS_TREE] ? k0 : 0;
        if (k > 0) {                                               // tomb-stone shrink (VLN-FF:607-610)
            double total = 0;
            for (int q = 0; q < n_seg; ++q)
                for (int j = 0; j < k; ++j) total += (double)d2[(int64_t)q * k_max + j];
            if (total > 1e6) {
                int kk = 0;
                for (int j = 0; j < k; ++j) {
                    double col = 0;
                    for (int q = 0; q < n_seg; ++q) col += (double)d2[(int64_t)q * k_max + j];
                    if (col < 1e6) ++kk;
                }
                k = kk;
            }
        }
        misc[0] = k;
        misc[1] = 0;
    });
    const int k = misc[0];
    cx.par(n_seg, [&](int q) {                                     // merge_target = argmax(softmax(logits)) (VLN-FF:619-621): first positive proposal
        int fp = -1;
        for (int j = 0; j < k; ++j) {
            const float* l = logits + ((int64_t)q * k_max + j) * 2;
            if (l[1] > l[0]) { fp = j; break; }
        }
        seg_slot[q] = fp;
    });
    const int n_new = cx.compact(n_seg, NO_LIMIT, [&](int q) { return seg_slot[q] < 0; }, [&](int q, int c) { new_rank[q] = c; rank_seg[c] = q; });
    // lowest unused instance ids / patch ids (VLN-FF:433-475)
    cx.compact(ns0 + n_new, n_new, [&](int i) { return !(i < ns0 && v.live[i] != 0); }, [&](int i, int c) { new_inst[c] = i; });
    if (compat_fixed) {
        cx.par(P, [&](int p) { new_pid[p] = row_base + p; });
    } else {
        cx.compact(row_base + P, P, [&](int i) { return v.owner[i] < 0; }, [&](int i, int c) { new_pid[c] = i; });
    }
    cx.par(n_new, [&](int c) { if (new_inst[c] < v.M) v.icnt[new_inst[c]] = 0; });   // (counts below are accumulated: a merge may target a slot opened in this view)
    cx.sync();
    const int stamp0 = v.hdr[H_STAMP];
    cx.par(n_seg, [&](int s) {
        const int cnt = seg_off[s + 1] - seg_off[s];
        if (seg_slot[s] < 0) {                                     // new instance (VLN-FF:633-648)
            const int c = new_rank[s], inst = new_inst[c];
            if (inst >= v.M) { cx.atomic_or(&misc[1], ERR_SLOT_OVERFLOW); seg_inst[s] = -1; return; }
            v.live[inst] = 1;
            v.istamp[inst] = stamp0 + c + 1;
            v.icx[inst] = new_cells[s * 3]; v.icy[inst] = new_cells[s * 3 + 1]; v.icz[inst] = new_cells[s * 3 + 2];
            cx.atomic_add(&v.icnt[inst], cnt);
            seg_slot[s] = inst;
            seg_inst[s] = inst;
        } else {                                                   // merge into the first positive proposal only (VLN-FF:651-691)
            const int inst = idx[(int64_t)s * k_max + seg_slot[s]];
            seg_slot[s] = -1;
            // Is the proposed slot a key of the instance dict WHEN THE REFERENCE REACHES SEGMENT s?  The reference walks the segments in
            // order, so a dead slot recycled in this very view exists for the segments behind the one that opens it and is a KeyError for
            // the ones in front.  live[] of such a slot is being written by the branch above in this same sweep: it is decided from the
            // (ascending) list of recycled slots and the opening segment's index instead, never by reading live[] of a recycled slot.
            bool ok = inst >= 0 && inst < ns0;
            if (ok) {
                int lo = 0, hi = n_new;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (new_inst[mid] < inst) lo = mid + 1; else hi = mid; }
                ok = (lo < n_new && new_inst[lo] == inst) ? (rank_seg[lo] < s) : (v.live[inst] != 0);
            }
            if (!ok) { cx.atomic_or(&misc[1], ERR_PROPOSAL_NOT_LIVE); seg_inst[s] = -1; return; }   // KeyError in the reference
            seg_inst[s] = inst;
            cx.atomic_add(&v.icnt[inst], cnt);
        }
    });
    // merged instances in first-touch order
    const int nd = cx.compact(n_seg, NO_LIMIT,
                              [&](int s) {
                                  if (seg_slot[s] >= 0 || seg_inst[s] < 0) return false;
                                  for (int q = 0; q < s; ++q)
                                      if (seg_slot[q] < 0 && seg_inst[q] == seg_inst[s]) return false;
                                  return true;
                              },
                              [&](int s, int c) { dirty_inst[c] = seg_inst[s]; });
    cx.one([&] {
        int nslots = ns0;
        if (n_new > 0 && new_inst[n_new - 1] < v.M && new_inst[n_new - 1] + 1 > nslots) nslots = new_inst[n_new - 1] + 1;
        v.hdr[H_NSLOTS] = nslots;
        v.hdr[H_NLIVE] += n_new;
        v.hdr[H_STAMP] = stamp0 + n_new;
        v.hdr[H_NROWS] = row_base + P;
        v.hdr[H_NOWNED] += P;
        dirty_off[0] = 0;
    });
    cx.par(P, [&](int t) {
        const int inst = seg_inst[tok_seg[t]];
        if (inst < 0) return;
        const int pid = new_pid[order[t]];
        v.owner[pid] = inst;
        v.stamp_of_pid[pid] = row_base + t;
        v.pid_of_stamp[row_base + t] = pid;
    });
    cx.sync();
    // member lists of the merged instances, in push order; the ids are used as ROW indices downstream (F11)
    const int n_stamps = row_base + P;
    int off = 0, bad = 0;
    for (int d = 0; d < nd; ++d) {
        const int inst = dirty_inst[d];
        const int room = rows_cap - off;
        const int cnt = cx.compact(n_stamps, room > 0 ? room : 0,
                                   [&](int u) {
                                       const int pid = v.pid_of_stamp[u];
                                       return pid >= 0 && v.stamp_of_pid[pid] == u && v.owner[pid] == inst;
                                   },
                                   [&](int u, int c) { dirty_rows[off + c] = v.pid_of_stamp[u]; });
        if (cnt != v.icnt[inst]) bad = ERR_ROWS_OVERFLOW;           // (also the self-check len(members) == icnt)
        off += cnt;
        cx.one([&] { dirty_off[d + 1] = off; });
    }
    cx.one([&] {
        vh[V_KEFF] = k;
        vh[V_NDIRTY] = nd;
        vh[V_DIRTY_ROWS] = off;
        vh[V_NSLOTS] = v.hdr[H_NSLOTS];
        vh[V_NLIVE] = v.hdr[H_NLIVE];
        vh[V_NOWNED] = v.hdr[H_NOWNED];
        vh[V_ERR] = misc[1] | bad;
        if (vh[V_ERR]) v.hdr[H_ERR] |= vh[V_ERR];
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// zone update of one view (VLN-FF:694-756 / 777-812).  merged_cells: the zone cells of the merged instances' NEW centroids, in
// dirty_inst order; new_cells: the cells of this frame's segments.  Outputs: the touched zones in key order with mode (0 new: position =
// mean of the members' positions; 1 existing: mean of the members' cell centres, VLN-FF:739-741), data row and member instances (CSR).
// scratch use: keys [3n] | firsts [n] | found [n] | new_rank [n] | zids [n] | zid_of [n]   (n = n_seg).
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx>
FFP_HD void plan_zones(Cx& cx, const View& v, int compat_fixed, const int32_t* dirty_inst, int n_dirty, const int32_t* merged_cells,
                       const int32_t* new_cells, int n_seg, int32_t* zone_row, int32_t* zone_mode, int32_t* zone_off, int32_t* zone_mem,
                       int mem_cap, int32_t* vh) {
    const int64_t n = n_seg;
    int32_t* keys = v.scratch;
    int32_t* firsts = v.scratch + 3 * n;
    int32_t* found = v.scratch + 4 * n;
    int32_t* new_rank = v.scratch + 5 * n;
    int32_t* zids = v.scratch + 6 * n;
    int32_t* zid_of = v.scratch + 7 * n;
    int32_t* misc = v.scratch + 8 * n;                             // [0] error bits
    cx.sync();
    cx.par(n_dirty, [&](int d) {
        const int i = dirty_inst[d];
        v.icx[i] = merged_cells[d * 3]; v.icy[i] = merged_cells[d * 3 + 1]; v.icz[i] = merged_cells[d * 3 + 2];
    });
    cx.one([&] { misc[0] = 0; });
    auto cell_less = [&](int a, int b) {                           // lexicographic order of two segments' cells
        const int ax = new_cells[a * 3], ay = new_cells[a * 3 + 1], az = new_cells[a * 3 + 2];
        const int bx = new_cells[b * 3], by = new_cells[b * 3 + 1], bz = new_cells[b * 3 + 2];
        return ax < bx || (ax == bx && (ay < by || (ay == by && az < bz)));
    };
    auto cell_same = [&](int a, int b) {
        return new_cells[a * 3] == new_cells[b * 3] && new_cells[a * 3 + 1] == new_cells[b * 3 + 1] && new_cells[a * 3 + 2] == new_cells[b * 3 + 2];
    };
    // torch.unique(dim=0): the distinct cells of this frame's 2D instances, lexicographically sorted
    const int nt = cx.compact(n_seg, NO_LIMIT,
                              [&](int s) {
                                  for (int q = 0; q < s; ++q)
                                      if (cell_same(q, s)) return false;
                                  return true;
                              },
                              [&](int s, int c) { firsts[c] = s; });
    cx.par(nt, [&](int c) {
        const int s = firsts[c];
        int r = 0;
        for (int c2 = 0; c2 < nt; ++c2) r += cell_less(firsts[c2], s) ? 1 : 0;
        keys[r * 3] = new_cells[s * 3]; keys[r * 3 + 1] = new_cells[s * 3 + 1]; keys[r * 3 + 2] = new_cells[s * 3 + 2];
    });
    // drop the invalidated edges: copy the valid ones into the other buffer
    const int sel0 = v.hdr[H_EDGE_SEL], ne0 = v.hdr[H_NEDGES];
    const int sel = 1 - sel0;
    const int32_t *oz = v.edge_z(sel0), *oi = v.edge_i(sel0);
    int32_t *ez = v.edge_z(sel), *ei = v.edge_i(sel);
    int ne = cx.compact(ne0, NO_LIMIT, [&](int e) { return oz[e] >= 0; }, [&](int e, int c) { ez[c] = oz[e]; ei[c] = oi[e]; });
    const int nzids0 = v.hdr[H_NZIDS], nzrows0 = v.hdr[H_NZROWS], stamp0 = v.hdr[H_STAMP], ns = v.hdr[H_NSLOTS];
    cx.par(nt, [&](int t) {                                        // dict lookup `zone key -> id`
        const int kx = keys[t * 3], ky = keys[t * 3 + 1], kz = keys[t * 3 + 2];
        int f = -1;
        for (int z = 0; z < nzids0; ++z)
            if (v.zlive[z] != 0 && v.zkx[z] == kx && v.zky[z] == ky && v.zkz[z] == kz) { f = z; break; }
        found[t] = f;
    });
    const int n_newz = cx.compact(nt, NO_LIMIT, [&](int t) { return found[t] < 0; }, [&](int t, int c) { new_rank[t] = c; });
    cx.compact(nzids0 + n_newz, n_newz, [&](int z) { return !(z < nzids0 && v.zlive[z] != 0); }, [&](int z, int c) { zids[c] = z; });
    cx.par(nt, [&](int t) {
        if (found[t] < 0) {                                        // new zone (VLN-FF:707-730)
            const int c = new_rank[t], zid = zids[c];
            if (zid >= v.Z || (!compat_fixed && nzrows0 + c >= v.Z)) { cx.atomic_or(&misc[0], ERR_SLOT_OVERFLOW); zid_of[t] = -1; zone_mode[t] = 0; zone_row[t] = 0; return; }
            v.zkx[zid] = keys[t * 3]; v.zky[zid] = keys[t * 3 + 1]; v.zkz[zid] = keys[t * 3 + 2];
            v.zkey_stamp[zid] = stamp0 + 2 * c + 1;
            v.zstamp[zid] = stamp0 + 2 * c + 2;
            v.zlive[zid] = 1;
            zone_mode[t] = 0;
            zone_row[t] = compat_fixed ? zid : nzrows0 + c;       // appended regardless of the id (quirk Z1)
            zid_of[t] = zid;
        } else {                                                   // existing zone (VLN-FF:734-756)
            zone_mode[t] = 1;
            zone_row[t] = found[t];
            zid_of[t] = found[t];
        }
    });
    cx.one([&] {
        v.hdr[H_STAMP] = stamp0 + 2 * n_newz;
        v.hdr[H_NZLIVE] += n_newz;
        int top = nzids0;
        if (n_newz > 0 && zids[n_newz - 1] < v.Z && zids[n_newz - 1] + 1 > top) top = zids[n_newz - 1] + 1;
        v.hdr[H_NZIDS] = top;
        if (compat_fixed) { if (top > nzrows0) v.hdr[H_NZROWS] = top; }
        else v.hdr[H_NZROWS] = nzrows0 + n_newz;
        zone_off[0] = 0;
    });
    // the snapshots of the touched zones are REPLACED (zmembers[zid] = mem)
    cx.par(ne, [&](int e) {
        const int z = ez[e];
        for (int t = 0; t < nt; ++t)
            if (zid_of[t] == z) { ez[e] = -1; break; }
    });
    cx.sync();
    int off = 0, bad = 0;
    for (int t = 0; t < nt; ++t) {
        const int kx = keys[t * 3], ky = keys[t * 3 + 1], kz = keys[t * 3 + 2];
        const int room = mem_cap - off;
        const int cnt = cx.compact(ns, room > 0 ? room : 0, [&](int i) { return v.icx[i] == kx && v.icy[i] == ky && v.icz[i] == kz; },
                                   [&](int i, int c) { zone_mem[off + c] = i; });
        const int zid = zid_of[t];
        const int fit = (zid >= 0 && ne + cnt <= v.E) ? cnt : 0;
        if (zid >= 0 && fit != cnt) bad = ERR_EDGE_OVERFLOW;
        cx.par(fit, [&](int c) { ez[ne + c] = zid; ei[ne + c] = zone_mem[off + c]; });
        cx.one([&] {
            zone_off[t + 1] = off + cnt;
            if (zid >= 0) v.zcnt[zid] = cnt;
        });
        ne += fit;
        off += cnt;
    }
    cx.one([&] {
        v.hdr[H_EDGE_SEL] = sel;
        v.hdr[H_NEDGES] = ne;
        end_view_lane0(v);
        vh[V_NTOUCHED] = nt;
        vh[V_ZONE_MEMBERS] = off;
        vh[V_NSLOTS] = v.hdr[H_NSLOTS];
        vh[V_NZROWS] = v.hdr[H_NZROWS];
        vh[V_NZIDS] = v.hdr[H_NZIDS];
        vh[V_NEDGES] = ne;
        vh[V_NLIVE] = v.hdr[H_NLIVE];
        vh[V_NZLIVE] = v.hdr[H_NZLIVE];
        vh[V_ERR] |= misc[0] | bad;
        if (vh[V_ERR]) v.hdr[H_ERR] |= vh[V_ERR];
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// live ids in dict insertion order (VLN-FF:825 / 844): rank by stamp
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx>
FFP_HD void live_ids(Cx& cx, const View& v, int32_t* inst_ids, int32_t* n_inst, int32_t* zone_ids, int32_t* n_zone, int cap) {
    cx.sync();
    const int ns = v.hdr[H_NSLOTS], nz = v.hdr[H_NZIDS];
    cx.par(ns, [&](int i) {
        if (!v.live[i]) return;
        const int s = v.istamp[i];
        int r = 0;
        for (int j = 0; j < ns; ++j) r += (v.live[j] != 0 && v.istamp[j] < s) ? 1 : 0;
        if (r < cap) inst_ids[r] = i;
    });
    cx.par(nz, [&](int z) {
        if (!v.zlive[z]) return;
        const int s = v.zstamp[z];
        int r = 0;
        for (int j = 0; j < nz; ++j) r += (v.zlive[j] != 0 && v.zstamp[j] < s) ? 1 : 0;
        if (r < cap) zone_ids[r] = z;
    });
    cx.one([&] {
        *n_inst = v.hdr[H_NLIVE];
        *n_zone = v.hdr[H_NZLIVE];
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------
// per-environment plans -> the flat CSR tables the float kernels consume (one context over ALL environments of the batch).
// Groups past the real ones are empty (offset = total) and point nowhere (inst = -1), so a launch sized for the upper bound is harmless.
// ---------------------------------------------------------------------------------------------------------------------------------
template <class Cx>
FFP_HD void flatten_merge(Cx& cx, int B, int n_max, const int32_t* slot, const int32_t* dirty_inst, const int32_t* dirty_off,
                          const int32_t* dirty_rows, int64_t rows_stride, const int32_t* vh, int32_t* tok_slot, int32_t* tok_row, int64_t tok_cap,
                          int32_t* grp_off, int32_t* grp_slot, int32_t* grp_inst, int32_t* totals /* [2 + 2B]: groups, tokens, token base / group base per env */) {
    const int G_ub = B * n_max;
    cx.one([&] {
        int g = 0, tot = 0;
        for (int e = 0; e < B; ++e) {
            const int nd = vh[e * V_WORDS + V_NDIRTY];
            totals[2 + e] = tot;
            totals[2 + B + e] = g;
            g += nd;
            tot += dirty_off[e * (n_max + 1) + nd];
        }
        totals[0] = g;
        totals[1] = tot;
    });
    const int n_groups = totals[0], total = totals[1];
    cx.par(G_ub, [&](int i) {                                      // (environment, d-th merged instance) -> its flat group
        const int e = i / n_max, d = i % n_max;
        if (d < vh[e * V_WORDS + V_NDIRTY]) {
            const int g = totals[2 + B + e] + d;
            grp_off[g] = totals[2 + e] + dirty_off[e * (n_max + 1) + d];
            grp_slot[g] = slot[e];
            grp_inst[g] = dirty_inst[e * n_max + d];
        }
        if (i >= n_groups) { grp_off[i] = total; grp_slot[i] = 0; grp_inst[i] = -1; }
        if (i == 0) grp_off[G_ub] = total;
    });
    for (int e = 0; e < B; ++e) {
        const int nd = vh[e * V_WORDS + V_NDIRTY];
        const int n = dirty_off[e * (n_max + 1) + nd], base = totals[2 + e];
        const int s = slot[e];
        const int32_t* src = dirty_rows + (int64_t)e * rows_stride;
        cx.par(n, [&](int i) {
            if (base + i < tok_cap) { tok_row[base + i] = src[i]; tok_slot[base + i] = s; }
        });
    }
    cx.sync();
}

template <class Cx>
FFP_HD void flatten_zones(Cx& cx, int B, int n_max, const int32_t* slot, const int32_t* zone_row, const int32_t* zone_mode, const int32_t* zone_off,
                          const int32_t* zone_mem, int64_t mem_stride, const int32_t* vh, int32_t* tok_slot, int32_t* tok_inst, int64_t tok_cap,
                          int32_t* grp_off, int32_t* grp_mode, int32_t* grp_slot, int32_t* grp_row, int32_t* totals) {
    const int G_ub = B * n_max;
    cx.one([&] {
        int g = 0, tot = 0;
        for (int e = 0; e < B; ++e) {
            const int nt = vh[e * V_WORDS + V_NTOUCHED];
            totals[2 + e] = tot;
            totals[2 + B + e] = g;
            g += nt;
            tot += zone_off[e * (n_max + 1) + nt];
        }
        totals[0] = g;
        totals[1] = tot;
    });
    const int n_groups = totals[0], total = totals[1];
    cx.par(G_ub, [&](int i) {
        const int e = i / n_max, t = i % n_max;
        if (t < vh[e * V_WORDS + V_NTOUCHED]) {
            const int g = totals[2 + B + e] + t;
            grp_off[g] = totals[2 + e] + zone_off[e * (n_max + 1) + t];
            grp_mode[g] = zone_mode[e * n_max + t];
            grp_slot[g] = slot[e];
            grp_row[g] = zone_row[e * n_max + t];
        }
        if (i >= n_groups) { grp_off[i] = total; grp_mode[i] = 0; grp_slot[i] = 0; grp_row[i] = 0; }
        if (i == 0) grp_off[G_ub] = total;
    });
    for (int e = 0; e < B; ++e) {
        const int nt = vh[e * V_WORDS + V_NTOUCHED];
        const int n = zone_off[e * (n_max + 1) + nt], base = totals[2 + e];
        const int s = slot[e];
        const int32_t* src = zone_mem + (int64_t)e * mem_stride;
        cx.par(n, [&](int i) {
            if (base + i < tok_cap) { tok_inst[base + i] = src[i]; tok_slot[base + i] = s; }
        });
    }
    cx.sync();
}

}  // namespace ffplan
