// ff_plan_kernels.hip -- the memory update's bookkeeping on the GPU: csrc/ff_plan.h run by one workgroup per environment.
// (Reference: Dynam3D_VLN/vlnce_baselines/models/feature_fields.py:362-393, 433-475, 623-691, 694-756, 825/844; the host counterpart
// with the same decisions is ff_state.cpp.)  Integer work on a few KB per environment: the cost is launch + latency, not bandwidth;
// what it buys is that the update never waits for the host between its kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"
#include "ff_plan.h"

namespace {

using namespace ffplan;
constexpr int PLAN_BLOCK = 1024;
constexpr int PLAN_WAVES = PLAN_BLOCK / 64;

struct BlockCtx {
    int* sm;   // LDS [PLAN_WAVES]
    __device__ void sync() const { __syncthreads(); }
    template <class F> __device__ void par(int n, F f) const {
        for (int i = threadIdx.x; i < n; i += PLAN_BLOCK) f(i);
    }
    template <class F> __device__ void one(F f) const {
        __syncthreads();
        if (threadIdx.x == 0) f();
        __syncthreads();
    }
    template <class P, class E> __device__ int compact(int n, int limit, P pred, E emit) const {
        __syncthreads();
        if (limit <= 0 || n <= 0) return 0;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int base = 0;
        for (int c0 = 0; c0 < n; c0 += PLAN_BLOCK) {
            const int i = c0 + threadIdx.x;
            const bool p = i < n && pred(i);
            const unsigned long long bal = __ballot(p);
            if (lane == 0) sm[wv] = __popcll(bal);
            __syncthreads();
            int off = base, tot = 0;
#pragma unroll
            for (int w = 0; w < PLAN_WAVES; ++w) {
                const int c = sm[w];
                if (w < wv) off += c;
                tot += c;
            }
            if (p) {
                const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < limit) emit(i, pos);
            }
            base += tot;
            __syncthreads();
            if (base >= limit) break;
        }
        return base < limit ? base : limit;
    }
    __device__ void atomic_add(int32_t* p, int32_t v) const { atomicAdd(p, v); }
    __device__ void atomic_or(int32_t* p, int32_t v) const { atomicOr(p, v); }
};

__device__ inline State as_state(const d3d_ffdev_state& s) {
    State t;
    t.hdr = s.hdr; t.rows = s.rows; t.inst = s.inst; t.zone = s.zone; t.edges = s.edges; t.scratch = s.scratch;
    t.R = s.R; t.M = s.M; t.Z = s.Z; t.E = s.E; t.W = s.W;
    t.compat_fixed = s.compat_fixed; t.P = s.P; t.K = s.K;
    t.tomb[0] = s.tomb[0]; t.tomb[1] = s.tomb[1]; t.tomb[2] = s.tomb[2];
    return t;
}

__global__ void __launch_bounds__(PLAN_BLOCK) k_ffdev_begin_view(d3d_ffdev_state s, const int32_t* slot, int32_t* k0, int32_t* tree_slots) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    const int e = blockIdx.x;
    const View v = view_of(as_state(s), slot[e]);
    begin_view(cx, v, s.K, k0 + e, tree_slots + e);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_apply_hits(d3d_ffdev_state s, const int32_t* slot, const int32_t* hits, int64_t hits_stride, const int32_t* n_hits, float* inst_pos,
                   float* inst_fts, int64_t m_cap, float* zone_pos, float* zone_fts, int64_t z_cap, int fts_dim) {
    __shared__ int sm[PLAN_WAVES];
    __shared__ int32_t tomb[3];
    BlockCtx cx{sm};
    const int e = blockIdx.x, sl = slot[e];
    if (threadIdx.x < 3) tomb[threadIdx.x] = s.tomb[threadIdx.x];
    const View v = view_of(as_state(s), sl);
    apply_hits(cx, v, hits + (int64_t)e * hits_stride, n_hits[e], tomb, inst_pos + (int64_t)sl * m_cap * 3, inst_fts + (int64_t)sl * m_cap * fts_dim,
               zone_pos + (int64_t)sl * z_cap * 3, zone_fts + (int64_t)sl * z_cap * fts_dim, fts_dim);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_plan_merge(d3d_ffdev_state s, const int32_t* slot, const int32_t* order, const int32_t* tok_seg, const int32_t* seg_off, const int32_t* n_seg,
                   int n_max, int k_max, const int32_t* k0, const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells,
                   int32_t* seg_slot, int32_t* dirty_inst, int32_t* dirty_off, int32_t* dirty_rows, int64_t rows_stride, int32_t* report) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    const int e = blockIdx.x;
    const View v = view_of(as_state(s), slot[e]);
    const int64_t q = (int64_t)e * n_max;
    plan_merge(cx, v, s.compat_fixed, s.P, order + (int64_t)e * s.P, tok_seg + (int64_t)e * s.P, seg_off + (int64_t)e * (n_max + 1), n_seg[e], k0[e], k_max,
               d2 + q * k_max, idx + q * k_max, logits + q * k_max * 2, new_cells + q * 3, seg_slot + q, dirty_inst + q, dirty_off + (int64_t)e * (n_max + 1),
               dirty_rows + (int64_t)e * rows_stride, (int)rows_stride, report + (int64_t)e * V_WORDS);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_flatten_merge(int B, int n_max, const int32_t* slot, const int32_t* dirty_inst, const int32_t* dirty_off, const int32_t* dirty_rows,
                      int64_t rows_stride, const int32_t* report, int32_t* tok_slot, int32_t* tok_row, int64_t tok_cap, int32_t* grp_off,
                      int32_t* grp_slot, int32_t* grp_inst, int32_t* totals) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    flatten_merge(cx, B, n_max, slot, dirty_inst, dirty_off, dirty_rows, rows_stride, report, tok_slot, tok_row, tok_cap, grp_off, grp_slot, grp_inst, totals);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_plan_zones(d3d_ffdev_state s, const int32_t* slot, const int32_t* dirty_inst, const int32_t* merged_cells, const int32_t* new_cells,
                   const int32_t* n_seg, int n_max, int32_t* zone_row, int32_t* zone_mode, int32_t* zone_off, int32_t* zone_mem, int64_t mem_stride,
                   int32_t* report) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    const int e = blockIdx.x;
    const View v = view_of(as_state(s), slot[e]);
    int gbase = 0;                                                 // this environment's first group among the flattened merge groups
    for (int b = 0; b < e; ++b) gbase += report[(int64_t)b * V_WORDS + V_NDIRTY];
    const int64_t q = (int64_t)e * n_max;
    plan_zones(cx, v, s.compat_fixed, dirty_inst + q, report[(int64_t)e * V_WORDS + V_NDIRTY], merged_cells + (int64_t)gbase * 3, new_cells + q * 3, n_seg[e],
               zone_row + q, zone_mode + q, zone_off + (int64_t)e * (n_max + 1), zone_mem + (int64_t)e * mem_stride, (int)mem_stride,
               report + (int64_t)e * V_WORDS);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_flatten_zones(int B, int n_max, const int32_t* slot, const int32_t* zone_row, const int32_t* zone_mode, const int32_t* zone_off,
                      const int32_t* zone_mem, int64_t mem_stride, const int32_t* report, int32_t* tok_slot, int32_t* tok_inst, int64_t tok_cap,
                      int32_t* grp_off, int32_t* grp_mode, int32_t* grp_slot, int32_t* grp_row, int32_t* totals) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    flatten_zones(cx, B, n_max, slot, zone_row, zone_mode, zone_off, zone_mem, mem_stride, report, tok_slot, tok_inst, tok_cap, grp_off, grp_mode, grp_slot,
                  grp_row, totals);
}

__global__ void __launch_bounds__(PLAN_BLOCK)
k_ffdev_live_ids(d3d_ffdev_state s, const int32_t* slot, int32_t* inst_ids, int32_t* n_inst, int32_t* zone_ids, int32_t* n_zone, int max_ids) {
    __shared__ int sm[PLAN_WAVES];
    BlockCtx cx{sm};
    const int e = blockIdx.x;
    const View v = view_of(as_state(s), slot[e]);
    live_ids(cx, v, inst_ids + (int64_t)e * max_ids, n_inst + e, zone_ids + (int64_t)e * max_ids, n_zone + e, max_ids);
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

int32_t d3d_ffdev_begin_view(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* k0, int32_t* tree_slots, void* stream) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_begin_view")) return rc;
    hipLaunchKernelGGL(k_ffdev_begin_view, dim3(B), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, *st, slot, k0, tree_slots);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_apply_hits(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* hits, int64_t hits_stride,
                             const int32_t* n_hits, float* inst_pos, float* inst_fts, int64_t m_cap, float* zone_pos, float* zone_fts,
                             int64_t z_cap, int32_t fts_dim, void* stream) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_apply_hits")) return rc;
    hipLaunchKernelGGL(k_ffdev_apply_hits, dim3(B), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, *st, slot, hits, hits_stride, n_hits, inst_pos, inst_fts,
                       m_cap, zone_pos, zone_fts, z_cap, fts_dim);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_plan_merge(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* order, const int32_t* tok_seg,
                             const int32_t* seg_off, const int32_t* n_seg, int32_t n_max, int32_t k_max, const int32_t* k0,
                             const float* d2, const int32_t* idx, const float* logits, const int32_t* new_cells, int32_t* seg_slot,
                             int32_t* dirty_inst, int32_t* dirty_off, int32_t* dirty_rows, int64_t rows_stride, int32_t* report,
                             void* stream) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_plan_merge")) return rc;
    if (n_max > st->P) {
        d3d_set_error_("d3d_ffdev_plan_merge: more segments than patches");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_ffdev_plan_merge, dim3(B), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, *st, slot, order, tok_seg, seg_off, n_seg, n_max, k_max, k0,
                       d2, idx, logits, new_cells, seg_slot, dirty_inst, dirty_off, dirty_rows, rows_stride, report);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_flatten_merge(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* dirty_inst, const int32_t* dirty_off,
                                const int32_t* dirty_rows, int64_t rows_stride, const int32_t* report, int32_t* tok_slot,
                                int32_t* tok_row, int64_t tok_cap, int32_t* grp_off, int32_t* grp_slot, int32_t* grp_inst,
                                int32_t* totals, void* stream) {
    if (B <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_ffdev_flatten_merge, dim3(1), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, B, n_max, slot, dirty_inst, dirty_off, dirty_rows,
                       rows_stride, report, tok_slot, tok_row, tok_cap, grp_off, grp_slot, grp_inst, totals);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_plan_zones(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, const int32_t* dirty_inst,
                             const int32_t* merged_cells, const int32_t* new_cells, const int32_t* n_seg, int32_t n_max,
                             int32_t* zone_row, int32_t* zone_mode, int32_t* zone_off, int32_t* zone_mem, int64_t mem_stride,
                             int32_t* report, void* stream) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_plan_zones")) return rc;
    hipLaunchKernelGGL(k_ffdev_plan_zones, dim3(B), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, *st, slot, dirty_inst, merged_cells, new_cells, n_seg, n_max,
                       zone_row, zone_mode, zone_off, zone_mem, mem_stride, report);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_flatten_zones(int32_t B, int32_t n_max, const int32_t* slot, const int32_t* zone_row, const int32_t* zone_mode,
                                const int32_t* zone_off, const int32_t* zone_mem, int64_t mem_stride, const int32_t* report,
                                int32_t* tok_slot, int32_t* tok_inst, int64_t tok_cap, int32_t* grp_off, int32_t* grp_mode,
                                int32_t* grp_slot, int32_t* grp_row, int32_t* totals, void* stream) {
    if (B <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_ffdev_flatten_zones, dim3(1), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, B, n_max, slot, zone_row, zone_mode, zone_off, zone_mem,
                       mem_stride, report, tok_slot, tok_inst, tok_cap, grp_off, grp_mode, grp_slot, grp_row, totals);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ffdev_live_ids(const d3d_ffdev_state* st, const int32_t* slot, int32_t B, int32_t* inst_ids, int32_t* n_inst,
                           int32_t* zone_ids, int32_t* n_zone, int32_t max_ids, void* stream) {
    if (B <= 0) return D3D_OK;
    if (int32_t rc = check_state(st, "d3d_ffdev_live_ids")) return rc;
    hipLaunchKernelGGL(k_ffdev_live_ids, dim3(B), dim3(PLAN_BLOCK), 0, (hipStream_t)stream, *st, slot, inst_ids, n_inst, zone_ids, n_zone, max_ids);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
