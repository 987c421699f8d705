// ff_plan_host.cpp -- TEST-ONLY: the d3d_ffdev_* entry points over HOST arrays (csrc/ff_plan.h with the one-lane context).
// Built only into the CPU-only test library (libd3d_ffstate_host.so, `python -m dynam3d_amd.build --host`), never into
// libdynam3d_hip.so: it lets `pytest -m "not gpu"` replay the device planner -- the same source the kernels compile -- against the
// reference-generated golden trajectories and against the host state machine (ff_state.cpp) in the GPU-less build container.
#include <algorithm>
#include <string>

#include "../../include/dynam3d_hip.h"
#include "ff_plan.h"

extern "C" void d3d_set_error_(const char* msg);

namespace {
using namespace ffplan;

State as_state(const d3d_ffdev_state& s) {
    State t;
    t.hdr = s.hdr; t.rows = s.rows; t.inst = s.inst; t.zone = s.zone; t.edges = s.edges; t.scratch = s.scratch;
    t.R = s.R; t.M = s.M; t.Z = s.Z; t.E = s.E; t.W = s.W;
    t.compat_fixed = s.compat_fixed; t.P = s.P; t.K = s.K;
    t.tomb[0] = s.tomb[0]; t.tomb[1] = s.tomb[1]; t.tomb[2] = s.tomb[2];
    return t;
}

int32_t check_state(const d3d_ffdev_state* st, const char* who) {
    if (!st || !st->hdr || !st->rows || !st->inst || !st->zone || !st->edges || !st->scratch) {
        d3d_set_error_((std::string(who) + ": incomplete planner state").c_str());
        return D3D_EINVAL;
    }
    const int64_t need = std::max<int64_t>(8 * (int64_t)st->P + 16, (int64_t)st->M + st->Z);
    if (st->W < need) {
        d3d_set_error_((std::string(who) + ": scratch smaller than max(8 * P + 16, M + Z)").c_str());
        return D3D_EINVAL;
    }
    return D3D_OK;
}
}  // namespace

extern "C" {

int32_t d3d_ffdev_begin_view(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* k0, int32_t* tree_slots, void*) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_begin_view")) return rc;
    SerialCtx cx;
    for (int e = 0; e < B; ++e) begin_view(cx, view_of(as_state(*st), slot[e]), st->K, k0 + e, tree_slots + e);
    return D3D_OK;
}

int32_t d3d_ffdev_apply_hits(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* hits, int64_t hits_stride,
                             const int32_t* n_hits, float* inst_pos, float* inst_fts, int64_t m_cap, float* zone_pos, float* zone_fts,
                             int64_t z_cap, int32_t fts_dim, void*) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_apply_hits")) return rc;
    SerialCtx cx;
    for (int e = 0; e < B; ++e) {
        const int sl = slot[e];
        apply_hits(cx, view_of(as_state(*st), sl), hits + (int64_t)e * hits_stride, n_hits[e], st->tomb, inst_pos + (int64_t)sl * m_cap * 3,
                   inst_fts + (int64_t)sl * m_cap * fts_dim, zone_pos + (int64_t)sl * z_cap * 3, zone_fts + (int64_t)sl * z_cap * fts_dim, fts_dim);
    }
    return D3D_OK;
}

int32_t d3d_ffdev_plan_merge(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* order, const int32_t* tok_seg,
                             const int32_t* seg_off, const int32_t* n_seg, int32_t n_max, int32_t k_max, const int32_t* k0,
                             const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells, int32_t* seg_slot,
                             int32_t* dirty_inst, int32_t* dirty_off, int32_t* dirty_rows, int64_t rows_stride, int32_t* report, void*) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_plan_merge")) return rc;
    if (n_max > st->P) {
        d3d_set_error_("d3d_ffdev_plan_merge: more segments than patches");
        return D3D_EINVAL;
    }
    SerialCtx cx;
    for (int e = 0; e < B; ++e) {
        const int64_t q = (int64_t)e * n_max;
        plan_merge(cx, view_of(as_state(*st), slot[e]), st->compat_fixed, st->P, order + (int64_t)e * st->P, tok_seg + (int64_t)e * st->P,
                   seg_off + (int64_t)e * (n_max + 1), n_seg[e], k0[e], k_max, d2 + q * k_max, idx + q * k_max, logits + q * k_max * 2, new_cells + q * 3,
                   seg_slot + q, dirty_inst + q, dirty_off + (int64_t)e * (n_max + 1), dirty_rows + (int64_t)e * rows_stride, (int)rows_stride,
                   report + (int64_t)e * V_WORDS);
    }
    return D3D_OK;
}

int32_t d3d_ffdev_flatten_merge(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* dirty_inst, const int32_t* dirty_off,
                                const int32_t* dirty_rows, int64_t rows_stride, const int32_t* report, int32_t* tok_slot,
                                int32_t* tok_row, int64_t tok_cap, int32_t* grp_off, int32_t* grp_slot, int32_t* grp_inst,
                                int32_t* totals, void*) {
    if (B <= 0) return D3D_OK;
    SerialCtx cx;
    flatten_merge(cx, B, n_max, slot, dirty_inst, dirty_off, dirty_rows, rows_stride, report, tok_slot, tok_row, tok_cap, grp_off, grp_slot, grp_inst, totals);
    return D3D_OK;
}

int32_t d3d_ffdev_plan_zones(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* dirty_inst,
                             const int32_t* merged_cells, const int32_t* new_cells, const int32_t* n_seg, int32_t n_max,
                             int32_t* zone_row, int32_t* zone_mode, int32_t* zone_off, int32_t* zone_mem, int64_t mem_stride,
                             int32_t* report, void*) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_plan_zones")) return rc;
    SerialCtx cx;
    int gbase = 0;
    for (int e = 0; e < B; ++e) {
        const int64_t q = (int64_t)e * n_max;
        const int nd = report[(int64_t)e * V_WORDS + V_NDIRTY];
        plan_zones(cx, view_of(as_state(*st), slot[e]), st->compat_fixed, dirty_inst + q, nd, merged_cells + (int64_t)gbase * 3, new_cells + q * 3, n_seg[e],
                   zone_row + q, zone_mode + q, zone_off + (int64_t)e * (n_max + 1), zone_mem + (int64_t)e * mem_stride, (int)mem_stride,
                   report + (int64_t)e * V_WORDS);
        gbase += nd;
    }
    return D3D_OK;
}

int32_t d3d_ffdev_flatten_zones(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* zone_row, const int32_t* zone_mode,
                                const int32_t* zone_off, const int32_t* zone_mem, int64_t mem_stride, const int32_t* report,
                                int32_t* tok_slot, int32_t* tok_inst, int64_t tok_cap, int32_t* grp_off, int32_t* grp_mode,
                                int32_t* grp_slot, int32_t* grp_row, int32_t* totals, void*) {
    if (B <= 0) return D3D_OK;
    SerialCtx cx;
    flatten_zones(cx, B, n_max, slot, zone_row, zone_mode, zone_off, zone_mem, mem_stride, report, tok_slot, tok_inst, tok_cap, grp_off, grp_mode, grp_slot,
                  grp_row, totals);
    return D3D_OK;
}

int32_t d3d_ffdev_live_ids(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* inst_ids, int32_t* n_inst,
                           int32_t* zone_ids, int32_t* n_zone, int32_t max_ids, void*) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_live_ids")) return rc;
    SerialCtx cx;
    for (int e = 0; e < B; ++e)
        live_ids(cx, view_of(as_state(*st), slot[e]), inst_ids + (int64_t)e * max_ids, n_inst + e, zone_ids + (int64_t)e * max_ids, n_zone + e, max_ids);
    return D3D_OK;
}

}  // extern "C"
