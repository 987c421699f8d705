// ff_state.cpp -- host-side integer control plane of the patch -> instance -> zone memory.
//
// Re-designs the dict bookkeeping of the reference's Feature_Fields
// (Dynam3D_VLN/vlnce_baselines/models/feature_fields.py: deletion cascade 362-393, id allocation
// 433-475, new/merge bookkeeping 623-691, zone update 694-756, live-id order 825/844) as a small C++
// state machine behind the C ABI.  It never touches float data: positions reach it only as integer
// cell indices (floor(p / cell_len)) computed by the HIP kernels, and everything it emits is an
// index list for the kernels (rows to tomb-stone, CSR member lists to reduce / re-encode).
//
// compat == 0 ('reference') reproduces two quirks so that trajectories match the reference
// bit-for-bit:  F11 -- patch ids are the lowest unused ids but index append-only rows;
//               Z1  -- a new zone's data is appended even when its id is a recycled low id.
// compat == 1 ('fixed'): id == row for patches and zones.
//
// Plain C++17, no HIP dependency: the same file is built into libdynam3d_hip.so (product) and into
// tests' CPU-only libd3d_ffstate.so so the bookkeeping is exercised by `pytest -m "not gpu"`.
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dynam3d_hip.h"

extern "C" void d3d_set_error_(const char* msg);

namespace {

using Cell = std::array<int32_t, 3>;

struct Env {
    int64_t n_rows = 0;
    std::vector<int32_t> owner;  // patch id -> instance slot, -1 = not in dict
    int64_t n_owned = 0;
    // instances (id == slot)
    int32_t n_slots = 0, n_live = 0;
    std::vector<uint8_t> live;
    std::vector<std::vector<int32_t>> members;
    std::vector<uint64_t> istamp;
    std::vector<Cell> cell;
    // zones
    int32_t n_zone_rows = 0, n_zlive = 0;
    std::vector<uint8_t> zlive;
    std::vector<std::vector<int32_t>> zmembers;
    std::vector<uint64_t> zstamp;
    std::vector<Cell> zkey_of;
    std::vector<uint64_t> zkey_stamp;
    std::map<Cell, int32_t> zkey;
    uint64_t stamp = 0;
    bool has_tree = false;
    int32_t tree_slots = 0;
    // scratch carried from plan_merge to plan_zones
    std::vector<Cell> frame_cells;
    std::vector<int32_t> dirty;

    void ensure_slot(int32_t s) {
        if (s >= (int32_t)live.size()) {
            size_t n = (size_t)s + 1;
            live.resize(n, 0);
            members.resize(n);
            istamp.resize(n, 0);
            cell.resize(n, Cell{0, 0, 0});
        }
    }
    void ensure_zone(int32_t z) {
        if (z >= (int32_t)zlive.size()) {
            size_t n = (size_t)z + 1;
            zlive.resize(n, 0);
            zmembers.resize(n);
            zstamp.resize(n, 0);
            zkey_of.resize(n, Cell{0, 0, 0});
            zkey_stamp.resize(n, 0);
        }
    }
};

template <class Flags>
void lowest_unused(const Flags& used, int64_t n_used_domain, int32_t want, std::vector<int32_t>& out) {
    // VLN-FF:433-475: first `want` non-negative integers that are not keys.
    out.clear();
    for (int64_t i = 0; (int32_t)out.size() < want; ++i) {
        if (i < n_used_domain && used(i)) continue;
        out.push_back((int32_t)i);
    }
}

}  // namespace

// One handle = the bookkeeping of one Feature_Fields object.  Every entry point takes the handle's mutex, so a handle may be shared by
// threads (calls are serialised; distinct handles never contend).
struct d3d_ff {
    mutable std::mutex mu;
    int32_t compat_fixed = 0, P = 576, K = 2;
    std::vector<Env> env;
    Cell tomb_cell{-5000, -5000, -5000};
};

static int32_t fail(int32_t code, const std::string& m) {
    d3d_set_error_(m.c_str());
    return code;
}

extern "C" {

d3d_ff* d3d_ff_create(int32_t compat_fixed, int32_t patches_per_view, int32_t num_proposals) {
    auto* f = new d3d_ff();
    f->compat_fixed = compat_fixed;
    f->P = patches_per_view;
    f->K = num_proposals;
    return f;
}

void d3d_ff_destroy(d3d_ff* ff) { delete ff; }

int32_t d3d_ff_set_tomb_cell(d3d_ff* ff, int32_t cx, int32_t cy, int32_t cz) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    ff->tomb_cell = Cell{cx, cy, cz};
    return D3D_OK;
}

int32_t d3d_ff_reset(d3d_ff* ff, int32_t batch_size) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (batch_size < 0) return fail(D3D_EINVAL, "reset: negative batch");
    ff->env.assign((size_t)batch_size, Env());
    return D3D_OK;
}

int32_t d3d_ff_pop(d3d_ff* ff, int32_t e) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "pop: bad env index");
    ff->env.erase(ff->env.begin() + e);
    return D3D_OK;
}

int32_t d3d_ff_batch_size(const d3d_ff* ff) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    return (int32_t)ff->env.size();
}

int64_t d3d_ff_count(const d3d_ff* ff, int32_t e, int32_t which) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return -1;
    const Env& v = ff->env[e];
    switch (which) {
        case 0: return v.n_rows;
        case 1: return v.n_slots;
        case 2: return v.n_live;
        case 3: return v.n_zone_rows;
        case 4: return v.n_zlive;
        case 5: return v.n_owned;
        case 6: return v.has_tree ? v.tree_slots : -1;
    }
    return -1;
}

int32_t d3d_ff_apply_hits(d3d_ff* ff, int32_t e, const int32_t* hits, int32_t n_hits, int32_t* dead_inst,
                          int32_t* n_dead_inst, int32_t* dead_zone, int32_t* n_dead_zone, int32_t cap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "apply_hits: bad env");
    Env& v = ff->env[e];
    int32_t ni = 0, nz = 0;
    for (int32_t h = 0; h < n_hits; ++h) {
        const int64_t pid = hits[h];
        if (pid < 0 || pid >= (int64_t)v.owner.size()) continue;
        const int32_t inst = v.owner[pid];
        if (inst < 0) continue;  // `if patch_id not in dict: continue`  (VLN-FF:365)
        v.owner[pid] = -1;
        --v.n_owned;
        auto& mem = v.members[inst];
        mem.erase(std::remove(mem.begin(), mem.end(), (int32_t)pid), mem.end());
        if (!mem.empty()) continue;
        // instance removed (VLN-FF:372-379)
        v.live[inst] = 0;
        --v.n_live;
        if (ni >= cap) return fail(D3D_ECAP, "apply_hits: dead instance list overflow");
        dead_inst[ni++] = inst;
        const Cell key = v.cell[inst];
        v.cell[inst] = ff->tomb_cell;
        auto it = v.zkey.find(key);
        if (it == v.zkey.end()) continue;
        const int32_t zid = it->second;
        auto& zm = v.zmembers[zid];
        zm.erase(std::remove(zm.begin(), zm.end(), inst), zm.end());
        if (!zm.empty()) continue;
        v.zkey.erase(it);  // zone removed (VLN-FF:388-393)
        v.zlive[zid] = 0;
        --v.n_zlive;
        if (nz >= cap) return fail(D3D_ECAP, "apply_hits: dead zone list overflow");
        dead_zone[nz++] = zid;
    }
    *n_dead_inst = ni;
    *n_dead_zone = nz;
    return D3D_OK;
}

int32_t d3d_ff_begin_view(d3d_ff* ff, int32_t e, int32_t* row_base, int32_t* k0, int32_t* has_tree) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "begin_view: bad env");
    Env& v = ff->env[e];
    *row_base = (int32_t)v.n_rows;
    *k0 = std::min(v.n_live, ff->K);  // VLN-FF:532
    *has_tree = v.has_tree ? 1 : 0;
    v.n_rows += ff->P;
    v.owner.resize((size_t)v.n_rows, -1);
    return D3D_OK;
}

int32_t d3d_ff_plan_merge(d3d_ff* ff, int32_t e, const int32_t* segm, int32_t n_seg, int32_t k0, int32_t k_max,
                          const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells,
                          int32_t* k_eff_out, int32_t* seg_slot, int32_t* dirty_inst, int32_t* n_dirty,
                          int32_t* dirty_off, int32_t* dirty_rows, int32_t rows_cap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "plan_merge: bad env");
    Env& v = ff->env[e];
    const int32_t P = ff->P;
    const int64_t row_base = v.n_rows - P;
    int32_t k = v.has_tree ? k0 : 0;
    if (k > 0) {  // tomb-stone shrink (VLN-FF:607-610)
        double total = 0;
        for (int32_t q = 0; q < n_seg; ++q)
            for (int32_t j = 0; j < k; ++j) total += (double)d2[(int64_t)q * k_max + j];
        if (total > 1e6) {
            int32_t kk = 0;
            for (int32_t j = 0; j < k; ++j) {
                double col = 0;
                for (int32_t q = 0; q < n_seg; ++q) col += (double)d2[(int64_t)q * k_max + j];
                if (col < 1e6) ++kk;
            }
            k = kk;
        }
    }
    *k_eff_out = k;
    // merge_target = argmax(softmax(logits)) (VLN-FF:619-621): 1 iff logit[1] > logit[0]
    std::vector<int32_t> first_pos((size_t)n_seg, -1);
    int32_t n_new = 0;
    for (int32_t q = 0; q < n_seg; ++q) {
        for (int32_t j = 0; j < k; ++j) {
            const float* l = logits + ((int64_t)q * k_max + j) * 2;
            if (l[1] > l[0]) {
                first_pos[q] = j;
                break;
            }
        }
        if (first_pos[q] < 0) ++n_new;
    }
    std::vector<int32_t> new_inst, new_pid;
    lowest_unused([&](int64_t i) { return v.live[(size_t)i] != 0; }, (int64_t)v.live.size(), n_new, new_inst);
    if (ff->compat_fixed) {
        new_pid.resize((size_t)P);
        for (int32_t p = 0; p < P; ++p) new_pid[p] = (int32_t)(row_base + p);
    } else {
        lowest_unused([&](int64_t i) { return v.owner[(size_t)i] >= 0; }, (int64_t)v.owner.size(), P, new_pid);
    }
    // CSR of patches by segment (ascending p inside a segment)
    std::vector<int32_t> soff((size_t)n_seg + 1, 0), sp((size_t)P);
    for (int32_t p = 0; p < P; ++p) {
        if (segm[p] < 0 || segm[p] >= n_seg) return fail(D3D_EINVAL, "plan_merge: patch_segm label out of range (labels must be dense 0..n-1)");
        ++soff[(size_t)segm[p] + 1];
    }
    for (int32_t s = 0; s < n_seg; ++s) soff[s + 1] += soff[s];
    {
        std::vector<int32_t> cur(soff.begin(), soff.end() - 1);
        for (int32_t p = 0; p < P; ++p) sp[(size_t)cur[segm[p]]++] = p;
    }
    v.frame_cells.resize((size_t)n_seg);
    for (int32_t s = 0; s < n_seg; ++s) v.frame_cells[s] = Cell{new_cells[s * 3], new_cells[s * 3 + 1], new_cells[s * 3 + 2]};
    v.dirty.clear();
    int32_t nxt = 0;
    for (int32_t s = 0; s < n_seg; ++s) {
        int32_t inst;
        if (first_pos[s] < 0) {  // new instance (VLN-FF:633-648)
            inst = new_inst[(size_t)nxt++];
            v.ensure_slot(inst);
            v.members[inst].clear();
            v.live[inst] = 1;
            ++v.n_live;
            v.istamp[inst] = ++v.stamp;
            v.cell[inst] = v.frame_cells[s];
            if (inst >= v.n_slots) v.n_slots = inst + 1;
            seg_slot[s] = inst;
        } else {  // merge into the first positive proposal only (VLN-FF:651-691)
            inst = idx[(int64_t)s * k_max + first_pos[s]];
            if (inst < 0 || inst >= (int32_t)v.live.size() || !v.live[inst])
                return fail(D3D_ESTATE, "plan_merge: proposal is not a live instance (KeyError in the reference)");
            if (std::find(v.dirty.begin(), v.dirty.end(), inst) == v.dirty.end()) v.dirty.push_back(inst);
            seg_slot[s] = -1;
        }
        auto& mem = v.members[inst];
        for (int32_t t = soff[s]; t < soff[s + 1]; ++t) {
            const int32_t pid = new_pid[(size_t)sp[(size_t)t]];
            mem.push_back(pid);
            if (v.owner[(size_t)pid] < 0) ++v.n_owned;
            v.owner[(size_t)pid] = inst;
        }
    }
    // member rows of the merged instances: ids are used as ROW indices (F11)
    int32_t nd = (int32_t)v.dirty.size();
    int64_t tot = 0;
    dirty_off[0] = 0;
    for (int32_t i = 0; i < nd; ++i) {
        const auto& mem = v.members[v.dirty[i]];
        if (tot + (int64_t)mem.size() > rows_cap) return fail(D3D_ECAP, "plan_merge: merged member rows overflow");
        std::memcpy(dirty_rows + tot, mem.data(), mem.size() * sizeof(int32_t));
        tot += (int64_t)mem.size();
        dirty_inst[i] = v.dirty[i];
        dirty_off[i + 1] = (int32_t)tot;
    }
    *n_dirty = nd;
    return D3D_OK;
}

int32_t d3d_ff_plan_zones(d3d_ff* ff, int32_t e, const int32_t* dirty_cells, int32_t* n_touched, int32_t* zone_row,
                          int32_t* zone_mode, int32_t* zone_off, int32_t* zone_members, int32_t zcap, int32_t mcap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "plan_zones: bad env");
    Env& v = ff->env[e];
    for (size_t i = 0; i < v.dirty.size(); ++i)
        v.cell[v.dirty[i]] = Cell{dirty_cells[i * 3], dirty_cells[i * 3 + 1], dirty_cells[i * 3 + 2]};
    // torch.unique(dim=0): lexicographically sorted distinct cells of this frame's 2D instances
    std::vector<Cell> keys(v.frame_cells);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    const int32_t nt = (int32_t)keys.size();
    if (nt > zcap) return fail(D3D_ECAP, "plan_zones: touched zone list overflow");
    std::vector<int32_t> zids;
    lowest_unused([&](int64_t i) { return v.zlive[(size_t)i] != 0; }, (int64_t)v.zlive.size(), nt, zids);
    int32_t nz = 0;
    int64_t tot = 0;
    zone_off[0] = 0;
    for (int32_t t = 0; t < nt; ++t) {
        const Cell& key = keys[t];
        std::vector<int32_t> mem;
        for (int32_t i = 0; i < v.n_slots; ++i)
            if (v.cell[i] == key) mem.push_back(i);
        int32_t zid;
        auto it = v.zkey.find(key);
        if (it == v.zkey.end()) {  // new zone (VLN-FF:707-730)
            zid = zids[(size_t)nz++];
            v.ensure_zone(zid);
            v.zkey[key] = zid;
            v.zkey_of[zid] = key;
            v.zkey_stamp[zid] = ++v.stamp;
            v.zlive[zid] = 1;
            ++v.n_zlive;
            v.zstamp[zid] = ++v.stamp;
            zone_mode[t] = 0;
            if (ff->compat_fixed) {
                zone_row[t] = zid;
                v.n_zone_rows = std::max(v.n_zone_rows, zid + 1);
            } else {
                zone_row[t] = v.n_zone_rows++;  // appended regardless of the id (quirk Z1)
            }
        } else {  // existing zone (VLN-FF:734-756)
            zid = it->second;
            zone_mode[t] = 1;
            zone_row[t] = zid;
        }
        if (tot + (int64_t)mem.size() > mcap) return fail(D3D_ECAP, "plan_zones: zone member list overflow");
        std::memcpy(zone_members + tot, mem.data(), mem.size() * sizeof(int32_t));
        tot += (int64_t)mem.size();
        zone_off[t + 1] = (int32_t)tot;
        v.zmembers[zid] = std::move(mem);
    }
    *n_touched = nt;
    return D3D_OK;
}

int32_t d3d_ff_end_view(d3d_ff* ff, int32_t e, int32_t* tree_slots) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "end_view: bad env");
    Env& v = ff->env[e];
    v.has_tree = v.n_slots > 0;  // get_instance_tree: [] when there are no slots (VLN-FF:243-247)
    v.tree_slots = v.n_slots;
    *tree_slots = v.has_tree ? v.n_slots : 0;
    return D3D_OK;
}

int32_t d3d_ff_rebuild_tree(d3d_ff* ff, int32_t e, int32_t* tree_slots) { return d3d_ff_end_view(ff, e, tree_slots); }   // (locks inside)

int32_t d3d_ff_live_ids(const d3d_ff* ff, int32_t e, int32_t* inst_ids, int32_t* n_inst, int32_t* zone_ids,
                        int32_t* n_zone, int32_t cap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    if (e < 0 || e >= (int32_t)ff->env.size()) return fail(D3D_EINVAL, "live_ids: bad env");
    const Env& v = ff->env[e];
    std::vector<std::pair<uint64_t, int32_t>> o;
    for (int32_t i = 0; i < (int32_t)v.live.size(); ++i)
        if (v.live[i]) o.emplace_back(v.istamp[i], i);
    std::sort(o.begin(), o.end());
    if ((int32_t)o.size() > cap) return fail(D3D_ECAP, "live_ids: instance list overflow");
    for (size_t i = 0; i < o.size(); ++i) inst_ids[i] = o[i].second;
    *n_inst = (int32_t)o.size();
    o.clear();
    for (int32_t i = 0; i < (int32_t)v.zlive.size(); ++i)
        if (v.zlive[i]) o.emplace_back(v.zstamp[i], i);
    std::sort(o.begin(), o.end());
    if ((int32_t)o.size() > cap) return fail(D3D_ECAP, "live_ids: zone list overflow");
    for (size_t i = 0; i < o.size(); ++i) zone_ids[i] = o[i].second;
    *n_zone = (int32_t)o.size();
    return D3D_OK;
}

int32_t d3d_ff_export_owner(const d3d_ff* ff, int32_t e, int32_t* owner, int64_t n) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    const Env& v = ff->env[e];
    if (n < (int64_t)v.owner.size()) return fail(D3D_ECAP, "export_owner: buffer too small");
    std::memcpy(owner, v.owner.data(), v.owner.size() * sizeof(int32_t));
    return D3D_OK;
}

int32_t d3d_ff_export_members(const d3d_ff* ff, int32_t e, int32_t which, int32_t* ids, int32_t* off, int32_t* flat,
                              int64_t flat_cap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    const Env& v = ff->env[e];
    const auto& lv = which == 0 ? v.live : v.zlive;
    const auto& st = which == 0 ? v.istamp : v.zstamp;
    const auto& mm = which == 0 ? v.members : v.zmembers;
    std::vector<std::pair<uint64_t, int32_t>> o;
    for (int32_t i = 0; i < (int32_t)lv.size(); ++i)
        if (lv[i]) o.emplace_back(st[i], i);
    std::sort(o.begin(), o.end());
    int64_t tot = 0;
    off[0] = 0;
    for (size_t i = 0; i < o.size(); ++i) {
        const auto& m = mm[o[i].second];
        if (tot + (int64_t)m.size() > flat_cap) return fail(D3D_ECAP, "export_members: buffer too small");
        std::memcpy(flat + tot, m.data(), m.size() * sizeof(int32_t));
        tot += (int64_t)m.size();
        ids[i] = o[i].second;
        off[i + 1] = (int32_t)tot;
    }
    return (int32_t)o.size();
}

int32_t d3d_ff_export_zone_keys(const d3d_ff* ff, int32_t e, int32_t* cells, int32_t* ids, int32_t cap) {
    std::lock_guard<std::mutex> d3d_lock_(ff->mu);
    const Env& v = ff->env[e];
    std::vector<std::pair<uint64_t, int32_t>> o;
    for (int32_t i = 0; i < (int32_t)v.zlive.size(); ++i)
        if (v.zlive[i]) o.emplace_back(v.zkey_stamp[i], i);
    std::sort(o.begin(), o.end());
    if ((int32_t)o.size() > cap) return fail(D3D_ECAP, "export_zone_keys: buffer too small");
    for (size_t i = 0; i < o.size(); ++i) {
        ids[i] = o[i].second;
        for (int a = 0; a < 3; ++a) cells[i * 3 + a] = v.zkey_of[o[i].second][a];
    }
    return (int32_t)o.size();
}

}  // extern "C"
