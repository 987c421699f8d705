// geometry_kernels.hip -- gfx950 kernels for the irregular / geometric half of the Dynam3D step:
// depth preprocessing, closed-form unprojection, frustum culling + tomb-stoning, brute-force KNN
// (torch_kdtree replacement), per-group centroid / geometry reductions, row gathers/scatters and
// the agent-frame query.  Compiled with -ffp-contract=off: every float op rounds once, in the
// order written, which makes these kernels BIT-EXACT against oracle/geometry.py.
//
// All of them are HBM/latency-bound byte movers: they are written for 64-wide
// waves (ballot/popcount compaction, one wave per 768-wide feature row = 3 x 16 B per lane),
// coalesced 16 B accesses on the feature pools and LDS-staged point tiles for the KNN.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int WAVE = 64;
constexpr int FTS = D3D_FTS_DIM;

__device__ __forceinline__ float pymodf(float a, float b) {
    // numpy float32 `%`: fmod, then add the divisor when the signs differ (npy_divmodf)
    float m = fmodf(a, b);
    if (m != 0.0f && ((m < 0.0f) != (b < 0.0f))) m = m + b;
    return m;
}

// ------------------------------------------------------------------------------------------------
// a1 preprocess_depth: block = 32 columns x 8 row groups of one image; column maxima combined through LDS (max is
// order-independent), then the same threads map their rows.  (One thread per column walked H rows alone: 206 us for a
// 224x224 batch of 8, all of it load latency.)
// ------------------------------------------------------------------------------------------------
constexpr int PD_COLS = 32, PD_GROUPS = 8;

__global__ void __launch_bounds__(PD_COLS * PD_GROUPS)
k_preprocess_depth(const float* __restrict__ in, float* __restrict__ out, int H, int W, float range, float lo100) {
    __shared__ float part[PD_GROUPS][PD_COLS];
    const int b = blockIdx.y;
    const int c = threadIdx.x % PD_COLS, g = threadIdx.x / PD_COLS;
    const int w = blockIdx.x * PD_COLS + c;
    const bool live = w < W;
    const float* src = in + (size_t)b * H * W + w;
    float* dst = out + (size_t)b * H * W + w;
    float mx = -INFINITY;
    if (live)
        for (int h = g; h < H; h += PD_GROUPS) mx = fmaxf(mx, src[(size_t)h * W] * 1.0f);
    part[g][c] = mx;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PD_GROUPS; ++j) mx = fmaxf(mx, part[j][c]);
    if (!live) return;
    for (int h = g; h < H; h += PD_GROUPS) {
        float d = src[(size_t)h * W] * 1.0f;
        if (d == 0.0f) d = mx;
        float t = d * range;
        t = t * 100.0f;
        t = lo100 + t;
        dst[(size_t)h * W] = t / 100.0f;
    }
}

// a2+a1: nearest resize to (h,w) then preprocess on the resized image; one block per image
__global__ void k_resize_preprocess(const float* __restrict__ in, float* __restrict__ out, int H, int W, int h, int w,
                                    float range, float lo100) {
    extern __shared__ float tile[];  // h*w resized values, then w column maxima
    const int b = blockIdx.x;
    const double sy = (double)H / (double)h, sx = (double)W / (double)w;
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) {
        const int r = i / w, c = i % w;
        int sr = (int)floor((double)r * sy);
        int sc = (int)floor((double)c * sx);
        sr = sr < H - 1 ? sr : H - 1;
        sc = sc < W - 1 ? sc : W - 1;
        tile[i] = in[((size_t)b * H + sr) * W + sc] * 1.0f;
    }
    __syncthreads();
    float* colmax = tile + h * w;
    for (int c = threadIdx.x; c < w; c += blockDim.x) {
        float mx = -INFINITY;
        for (int r = 0; r < h; ++r) mx = fmaxf(mx, tile[r * w + c]);
        colmax[c] = mx;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) {
        float d = tile[i];
        if (d == 0.0f) d = colmax[i % w];
        float t = d * range;
        t = t * 100.0f;
        t = lo100 + t;
        out[(size_t)b * h * w + i] = t / 100.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// a6 get_patch_segm after the segmenter (VLN-FF:407-420): one block per image.  Only the h*w pixels that the nearest resize
// samples are looked at: for each, the LAST mask that covers it (none -> group 0), then labels -> rank among the labels
// present (torch.unique order).  masks: (total,H,W) u8 {0,1}, image i owns masks [off[i], off[i+1]).
// ------------------------------------------------------------------------------------------------
constexpr int SEGM_MAX_MASKS = 4096;

__global__ void __launch_bounds__(256)
k_patch_segm(const uint8_t* __restrict__ masks, const int32_t* __restrict__ off, int H, int W, int h, int w,
             int32_t* __restrict__ segm, int32_t* __restrict__ n_seg) {
    __shared__ int32_t rank[SEGM_MAX_MASKS];          // present flag, then exclusive rank
    __shared__ int32_t wsum[4];
    extern __shared__ int32_t label[];                // h*w
    const int img = blockIdx.x, tid = threadIdx.x;
    const int m0 = off[img], nm = off[img + 1] - m0;
    for (int g = tid; g < nm; g += 256) rank[g] = 0;
    if (nm == 0 && tid == 0) rank[0] = 0;
    __syncthreads();
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;        // ATen nearest: float32 scale
    const size_t plane = (size_t)H * W;
    for (int p = tid; p < h * w; p += 256) {
        const int r = p / w, c = p % w;
        int sr = (int)floorf((float)r * sy), sc = (int)floorf((float)c * sx);
        sr = sr < H - 1 ? sr : H - 1;
        sc = sc < W - 1 ? sc : W - 1;
        const uint8_t* px = masks + (size_t)m0 * plane + (size_t)sr * W + sc;
        int lab = 0;
        for (int g = nm - 1; g > 0; --g)
            if (px[(size_t)g * plane] == 1) {
                lab = g;
                break;
            }
        label[p] = lab;
        rank[lab] = 1;                                  // benign race: every writer stores 1
    }
    __syncthreads();
    // exclusive scan of the present flags (nm <= 4096: 16 flags per thread, wave scan, 4 wave sums)
    constexpr int PER = SEGM_MAX_MASKS / 256;
    const int lim = nm > 0 ? nm : 1;
    int loc[PER], tot = 0;
    for (int j = 0; j < PER; ++j) {
        const int g = tid * PER + j;
        loc[j] = tot;
        tot += (g < lim) ? rank[g] : 0;
    }
    int incl = tot;
    for (int o = 1; o < WAVE; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & (WAVE - 1)) >= o) incl += v;
    }
    if ((tid & (WAVE - 1)) == WAVE - 1) wsum[tid >> 6] = incl;
    __syncthreads();
    int base = incl - tot;
    for (int wv = 0; wv < (tid >> 6); ++wv) base += wsum[wv];
    __syncthreads();
    for (int j = 0; j < PER; ++j) {
        const int g = tid * PER + j;
        if (g < lim) rank[g] = base + loc[j];
    }
    if (tid == 255) n_seg[img] = base + tot;
    __syncthreads();
    for (int p = tid; p < h * w; p += 256) segm[(size_t)img * h * w + p] = rank[label[p]];
}

// ------------------------------------------------------------------------------------------------
// a5 unprojection + append
// ------------------------------------------------------------------------------------------------
__global__ void k_unproject_append(const float* __restrict__ depth24, const d3d_pose* __restrict__ pose,
                                   const int32_t* __restrict__ slot, const int32_t* __restrict__ row_base, int P, int W,
                                   const float* __restrict__ tan_xy, const float* __restrict__ tan_z,
                                   const float* __restrict__ dir0, float th, float* __restrict__ rows_pos,
                                   float* __restrict__ rows_dir, float* __restrict__ rows_scale, int64_t n_cap) {
    const int e = blockIdx.x;
    const d3d_pose ps = pose[e];
    const int64_t base = (int64_t)slot[e] * n_cap + row_base[e];
    const float two_pi = 6.2831855f;  // float32(2*pi)
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const float d = depth24[(size_t)e * P + p];
        const float dx = d * tan_xy[p];
        const float dz = d * tan_z[p];
        float sc = d * th;
        sc = sc * 2.0f;
        sc = sc / (float)W;
        const float dir = pymodf(dir0[p] + ps.heading, two_pi);
        const float a = dx * ps.cos_h, bq = d * ps.sin_h;
        const float rel_x = a - bq;
        const float c = dx * ps.sin_h, dq = d * ps.cos_h;
        const float rel_y = c + dq;
        const int64_t r = base + p;
        rows_pos[r * 3 + 0] = rel_x + ps.cam[0];
        rows_pos[r * 3 + 1] = rel_y + ps.cam[1];
        rows_pos[r * 3 + 2] = dz + ps.cam[2];
        rows_dir[r] = dir;
        rows_scale[r] = sc;
    }
}

// Intrinsics branch of update_feature_fields (PRE-FF:81-94, 905-916): project_depth_to_3d (Open3D create_from_depth_image
// restated, see oracle/geometry.py::project_depth_to_3d) + nearest (h,w) sampling + world transform + get_heading_angle.
// One block per (env) view.  Pass 1: image maximum (zero pixels take it, PRE-FF:82) and validity of EVERY pixel (one
// invalid pixel makes the reference fall back to all-zero points).  Pass 2: the h*w sampled pixels in double.
__global__ void __launch_bounds__(256)
k_unproject_pinhole_append(const float* __restrict__ depth, int Hd, int Wd, const d3d_pinhole_unproject* __restrict__ cams,
                           const int32_t* __restrict__ slot, const int32_t* __restrict__ row_base, int h, int w, int input_width,
                           float* __restrict__ rows_pos, float* __restrict__ rows_dir, float* __restrict__ rows_scale, int64_t n_cap) {
    __shared__ float red[256];
    __shared__ int bad[256];
    const int e = blockIdx.x, tid = threadIdx.x;
    const d3d_pinhole_unproject cm = cams[e];
    const float* img = depth + (size_t)e * Hd * Wd;
    const int n = Hd * Wd;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, img[i]);
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    mx = red[0];
    auto metres = [&](float raw) -> float {            // uint16 cast, / depth_scale, >= trunc -> 0 (Open3D ConvertDepthToFloatImage)
        if (raw == 0.0f) raw = mx;
        const float f = (float)(uint16_t)raw / cm.depth_scale;
        return f >= cm.depth_trunc ? 0.0f : f;
    };
    int nb = 0;
    for (int i = tid; i < n; i += 256) nb |= !(metres(img[i]) > 0.0f);
    bad[tid] = nb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) bad[tid] |= bad[tid + o];
        __syncthreads();
    }
    const bool all_zero = bad[0] != 0;
    const float sy = (float)Hd / (float)h, sx = (float)Wd / (float)w;      // ATen nearest: float32 scale
    const int64_t base = (int64_t)slot[e] * n_cap + row_base[e];
    for (int p = tid; p < h * w; p += 256) {
        const int r = p / w, c = p % w;
        int sr = (int)floorf((float)r * sy), sc = (int)floorf((float)c * sx);
        sr = sr < Hd - 1 ? sr : Hd - 1;
        sc = sc < Wd - 1 ? sc : Wd - 1;
        double x = 0.0, y = 0.0, z = 0.0;
        if (!all_zero) {
            z = (double)metres(img[(size_t)sr * Wd + sc]);
            x = ((double)sc - cm.cx) * z / cm.fx;
            y = ((double)sr - cm.cy) * z / cm.fy;
        }
        const float xf = (float)x, yf = (float)y, zf = (float)z;            // points.astype(np.float32)  PRE-FF:907
        float s = zf * cm.scale_tan;
        s = s * 2.0f;
        s = s / (float)input_width;
        const double wx = (cm.R[0] * (double)xf + cm.R[1] * (double)yf + cm.R[2] * (double)zf) + cm.T[0];
        const double wy = (cm.R[3] * (double)xf + cm.R[4] * (double)yf + cm.R[5] * (double)zf) + cm.T[1];
        const double wz = (cm.R[6] * (double)xf + cm.R[7] * (double)yf + cm.R[8] * (double)zf) + cm.T[2];
        double xy = sqrt(wx * wx + wy * wy);                                  // get_heading_angle  PRE-FF:378-387
        if (xy < 1e-4) xy = 1e-4;
        double ang = -asin(wx / xy);
        if (wy < 0.0) ang = ang - 3.141592653589793;
        const int64_t row = base + p;
        rows_pos[row * 3 + 0] = (float)wx;
        rows_pos[row * 3 + 1] = (float)wy;
        rows_pos[row * 3 + 2] = (float)wz;
        rows_dir[row] = (float)ang;
        rows_scale[row] = s;
    }
}

template <bool F16IN>
__global__ void k_append_fts(const void* __restrict__ grid, const int32_t* __restrict__ slot,
                             const int32_t* __restrict__ row_base, int n_env, int P, uint16_t* __restrict__ rows_fts,
                             int64_t n_cap) {
    // one wave per row: 768 values = 64 lanes x 12 (f32 in: 3 x float4 -> 3 x 8 B of halves)
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    const int e = wave / P, p = wave % P;
    if (e >= n_env) return;
    const int64_t r = (int64_t)slot[e] * n_cap + row_base[e] + p;
    uint16_t* dst = rows_fts + r * FTS;
    if (F16IN) {
        const uint2* src = reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(grid) + ((size_t)e * P + p) * FTS);
        uint2* d2 = reinterpret_cast<uint2*>(dst);
        for (int i = lane; i < FTS / 4; i += WAVE) d2[i] = src[i];
    } else {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(grid) + ((size_t)e * P + p) * FTS);
        uint2* d2 = reinterpret_cast<uint2*>(dst);
        for (int i = lane; i < FTS / 4; i += WAVE) {
            const float4 v = src[i];
            const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
            uint2 o;
            o.x = *reinterpret_cast<const uint32_t*>(&a);
            o.y = *reinterpret_cast<const uint32_t*>(&b);
            d2[i] = o;
        }
    }
}

// a13
__global__ void k_patch_3d_info(const float* __restrict__ depth24, int N, int P, int W, const float* __restrict__ tan_xy,
                                const float* __restrict__ tan_z, const float* __restrict__ dir0, float th,
                                float* __restrict__ rel_x, float* __restrict__ rel_y, float* __restrict__ rel_z,
                                float* __restrict__ dir, float* __restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * P) return;
    const int p = (int)(i % P);
    const float d = depth24[i];
    rel_x[i] = d * tan_xy[p];
    rel_y[i] = d;
    rel_z[i] = d * tan_z[p];
    float sc = d * th;
    sc = sc * 2.0f;
    scale[i] = sc / (float)W;
    dir[i] = pymodf(dir0[p], 6.2831855f);
}

// ------------------------------------------------------------------------------------------------
// a4 frustum test (shared by the mutating cull and the pure mask kernel)
// ------------------------------------------------------------------------------------------------
struct FrustumCam {
    float fx, fy, cx, cy, near_, far_, slack;
    int Hd, Wd;
};

__device__ __forceinline__ bool frustum_hit(float x, float y, float z, const d3d_pose& ps, const FrustumCam& cm,
                                            const float* __restrict__ depth) {
    const float px = x - ps.cam[0], py = y - ps.cam[1], pz = z - ps.cam[2];
    const float a = px * ps.cos_nh, b = py * ps.sin_nh;
    const float rx = a - b;
    const float c = px * ps.sin_nh, d = py * ps.cos_nh;
    const float ry = c + d;
    const float X = rx, Y = -pz, Z = ry;  // (rel_x, -rel_z, rel_y)  VLN-FF:102
    const float u1 = cm.fx * X, u2 = cm.cx * Z;
    const float uh = u1 + u2;
    const float v1 = cm.fy * Y, v2 = cm.cy * Z;
    const float vh = v1 + v2;
    const float uf = uh / Z, vf = vh / Z;
    // trunc(uf) in [0, Wd-1]  <=>  -1 < uf < Wd ; NaN/inf fail (== INT64_MIN path of the reference)
    const bool in_img = (uf > -1.0f) && (uf < (float)cm.Wd) && (vf > -1.0f) && (vf < (float)cm.Hd);
    if (!(in_img && Z >= cm.near_ && Z <= cm.far_)) return false;
    const int u = (int)uf, v = (int)vf;
    const float cd = depth[(size_t)v * cm.Wd + u] + cm.slack;
    return Z < cd;
}

// Intrinsics / extrinsics cull: get_frustum_mask (PRE-FF:98-118).  The two einsums are evaluated as ATen's CPU matmul does
// for these shapes -- acc = m0*a, then one fused multiply-add per further term, in column order -- which oracle/geometry.py::
// frustum_mask_pinhole matched bit for bit against the reference on every intermediate value; cm.fx..cy are unused here.
__device__ __forceinline__ bool frustum_hit(float x, float y, float z, const d3d_pinhole_view& pv, const FrustumCam& cm,
                                            const float* __restrict__ depth) {
    const float* V = pv.view;
    const float X = fmaf(V[3], 1.0f, fmaf(V[2], z, fmaf(V[1], y, V[0] * x)));
    const float Y = fmaf(V[7], 1.0f, fmaf(V[6], z, fmaf(V[5], y, V[4] * x)));
    const float Z = fmaf(V[11], 1.0f, fmaf(V[10], z, fmaf(V[9], y, V[8] * x)));
    const float* K = pv.K;
    const float uh = fmaf(K[2], Z, fmaf(K[1], Y, K[0] * X));
    const float vh = fmaf(K[5], Z, fmaf(K[4], Y, K[3] * X));
    const float zh = fmaf(K[8], Z, fmaf(K[7], Y, K[6] * X));
    const float uf = uh / zh, vf = vh / zh;
    const bool in_img = (uf > -1.0f) && (uf < (float)cm.Wd) && (vf > -1.0f) && (vf < (float)cm.Hd);
    if (!(in_img && Z >= cm.near_ && Z <= cm.far_)) return false;
    const int u = (int)uf, v = (int)vf;
    const float cd = depth[(size_t)v * cm.Wd + u] + cm.slack;
    return Z < cd;
}

template <class POSE>
__global__ void k_frustum_mask(const float* __restrict__ pts, int64_t n, const float* __restrict__ depth, POSE ps,
                               FrustumCam cm, uint8_t* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mask[i] = frustum_hit(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], ps, cm, depth) ? 1 : 0;
}

template <class POSE>
__global__ void k_frustum_cull(float* __restrict__ rows_pos, uint16_t* __restrict__ rows_fts, float* __restrict__ rows_dir,
                               float* __restrict__ rows_scale, int64_t n_cap, const int32_t* __restrict__ slot,
                               const int32_t* __restrict__ n_rows, const float* __restrict__ depth,
                               const POSE* __restrict__ pose, FrustumCam cm, int32_t* __restrict__ hits,
                               int32_t* __restrict__ n_hits, int hit_cap, uint8_t* __restrict__ mask) {
    const int e = blockIdx.y;
    const int nr = n_rows[e];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    if ((r - lane) >= nr) return;  // whole wave out of range (wave-uniform)
    const int64_t base = (int64_t)slot[e] * n_cap;
    bool hit = false;
    if (r < nr) {
        const float* p = rows_pos + (base + r) * 3;
        hit = frustum_hit(p[0], p[1], p[2], pose[e], cm, depth + (size_t)e * cm.Hd * cm.Wd);
        if (mask) mask[(size_t)e * n_cap + r] = hit ? 1 : 0;
    }
    const unsigned long long bal = __ballot(hit);
    if (bal == 0ull) return;
    int wbase = 0;
    if (lane == 0) wbase = atomicAdd(&n_hits[e], __popcll(bal));
    wbase = __shfl(wbase, 0);
    if (hit) {
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (wbase + rank < hit_cap) hits[(size_t)e * hit_cap + wbase + rank] = r;
        float* p = rows_pos + (base + r) * 3;
        p[0] = D3D_TOMBSTONE;
        p[1] = D3D_TOMBSTONE;
        p[2] = D3D_TOMBSTONE;
        rows_dir[base + r] = 0.0f;
        rows_scale[base + r] = 0.0f;
    }
    // cooperative zeroing of the 1536-byte feature rows of this wave's hits (16 B per lane)
    unsigned long long rem = bal;
    const int r0 = r - lane;
    while (rem) {
        const int l = __ffsll((long long)rem) - 1;
        rem &= rem - 1;
        uint4* row = reinterpret_cast<uint4*>(rows_fts + (base + r0 + l) * FTS);
        const uint4 z = make_uint4(0, 0, 0, 0);
        row[lane] = z;
        if (lane < (FTS * 2 / 16 - WAVE)) row[WAVE + lane] = z;
    }
}

// ------------------------------------------------------------------------------------------------
// a8 brute-force KNN: one thread per query, points staged through LDS, top-k in registers
// ------------------------------------------------------------------------------------------------
constexpr int KNN_TILE = 1024;
constexpr int KNN_BLOCK = 256;

template <int K>
__global__ void __launch_bounds__(KNN_BLOCK)
k_knn(const float* __restrict__ points, int64_t point_stride, const int32_t* __restrict__ n_points,
      const float* __restrict__ queries, int64_t query_stride, const int32_t* __restrict__ n_queries,
      const int32_t* __restrict__ kk, int max_queries, float* __restrict__ d2_out, int32_t* __restrict__ idx_out) {
    __shared__ float tile[KNN_TILE * 3];
    const int b = blockIdx.y;
    const int nq = n_queries[b], np = n_points[b];
    if ((int)(blockIdx.x * KNN_BLOCK) >= nq) return;
    const int k = kk[b];
    const int q = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const bool active = q < nq;
    const float* P = points + (size_t)b * point_stride;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const float* Q = queries + (size_t)b * query_stride + (size_t)q * 3;
        qx = Q[0];
        qy = Q[1];
        qz = Q[2];
    }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = INFINITY;
        bi[j] = -1;
    }
    for (int t0 = 0; t0 < np; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, np - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += KNN_BLOCK) tile[i] = P[(size_t)t0 * 3 + i];
        __syncthreads();
        if (active) {
            for (int i = 0; i < cnt; ++i) {
                const float dx = qx - tile[i * 3], dy = qy - tile[i * 3 + 1], dz = qz - tile[i * 3 + 2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                if (d < bd[K - 1]) {  // strict: equal distances keep the earlier (lower) index
                    bd[K - 1] = d;
                    bi[K - 1] = t0 + i;
#pragma unroll
                    for (int j = K - 1; j > 0; --j) {
                        if (bd[j] < bd[j - 1]) {
                            const float td = bd[j];
                            bd[j] = bd[j - 1];
                            bd[j - 1] = td;
                            const int ti = bi[j];
                            bi[j] = bi[j - 1];
                            bi[j - 1] = ti;
                        }
                    }
                }
            }
        }
    }
    if (active) {
        const size_t o = ((size_t)b * max_queries + q) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < k) {
                d2_out[o + j] = bd[j];
                idx_out[o + j] = bi[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// KNN over LARGE point sets (the Pretrain GT instance cloud, PRE-FF:977-983: 1e5-1e6 points, k = 1, 576 queries per view): k_knn gives every
// workgroup ALL points of its environment, so 4 608 queries x 2e5 points ran on 24 workgroups (6 ms).  Here blockIdx.z cuts the point range
// into `n_chunks` pieces; every (query block, chunk) workgroup writes its own top-k -- same d^2 expression, same strict-less insertion --
// and k_knn_merge folds the chunks in ascending chunk order with the same insertion, so equal distances still go to the lower index:
// bit-identical to k_knn.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(KNN_BLOCK)
k_knn_chunk(const float* __restrict__ points, int64_t point_stride, const int32_t* __restrict__ n_points, const float* __restrict__ queries,
            int64_t query_stride, const int32_t* __restrict__ n_queries, int max_queries, int n_chunks, float* __restrict__ ws_d2,
            int32_t* __restrict__ ws_idx) {
    __shared__ float tile[KNN_TILE * 3];
    const int b = blockIdx.y, ch = blockIdx.z;
    const int nq = n_queries[b], np = n_points[b];
    if ((int)(blockIdx.x * KNN_BLOCK) >= nq) return;
    const int per = ((np + n_chunks - 1) / n_chunks + KNN_TILE - 1) / KNN_TILE * KNN_TILE;       // whole tiles per chunk
    const int p0 = ch * per, p1 = min(np, p0 + per);
    const int q = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const bool active = q < nq;
    const float* P = points + (size_t)b * point_stride;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const float* Q = queries + (size_t)b * query_stride + (size_t)q * 3;
        qx = Q[0];
        qy = Q[1];
        qz = Q[2];
    }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = INFINITY;
        bi[j] = -1;
    }
    for (int t0 = p0; t0 < p1; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, p1 - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 3; i += KNN_BLOCK) tile[i] = P[(size_t)t0 * 3 + i];
        __syncthreads();
        if (active) {
            for (int i = 0; i < cnt; ++i) {
                const float dx = qx - tile[i * 3], dy = qy - tile[i * 3 + 1], dz = qz - tile[i * 3 + 2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                if (d < bd[K - 1]) {
                    bd[K - 1] = d;
                    bi[K - 1] = t0 + i;
#pragma unroll
                    for (int j = K - 1; j > 0; --j) {
                        if (bd[j] < bd[j - 1]) {
                            const float td = bd[j];
                            bd[j] = bd[j - 1];
                            bd[j - 1] = td;
                            const int ti = bi[j];
                            bi[j] = bi[j - 1];
                            bi[j - 1] = ti;
                        }
                    }
                }
            }
        }
    }
    if (active) {
        const size_t o = (((size_t)b * max_queries + q) * n_chunks + ch) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            ws_d2[o + j] = bd[j];
            ws_idx[o + j] = bi[j];
        }
    }
}

template <int K>
__global__ void __launch_bounds__(KNN_BLOCK)
k_knn_merge(const float* __restrict__ ws_d2, const int32_t* __restrict__ ws_idx, const int32_t* __restrict__ n_queries, const int32_t* __restrict__ kk,
            int max_queries, int n_chunks, float* __restrict__ d2_out, int32_t* __restrict__ idx_out) {
    const int b = blockIdx.y, q = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (q >= n_queries[b]) return;
    const int k = kk[b];
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = INFINITY;
        bi[j] = -1;
    }
    const size_t base = ((size_t)b * max_queries + q) * n_chunks * K;
    for (int c = 0; c < n_chunks * K; ++c) {              // chunk-major, each chunk's list ascending: ties keep the lower chunk = lower index
        const float d = ws_d2[base + c];
        if (d < bd[K - 1]) {
            bd[K - 1] = d;
            bi[K - 1] = ws_idx[base + c];
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                if (bd[j] < bd[j - 1]) {
                    const float td = bd[j];
                    bd[j] = bd[j - 1];
                    bd[j - 1] = td;
                    const int ti = bi[j];
                    bi[j] = bi[j - 1];
                    bi[j - 1] = ti;
                }
            }
        }
    }
    const size_t o = ((size_t)b * max_queries + q) * K;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            d2_out[o + j] = bd[j];
            idx_out[o + j] = bi[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a21 radius-limited KNN (the novel-view renderer, PRE-FF:540-566: neighbours at >= 1 m are discarded right behind the query, so only
// neighbours INSIDE the radius have to be exact).  Same thread-per-query top-k as k_knn, but a workgroup first boxes its 256 queries
// (consecutive depth samples of one ray: a thin 5 m segment) and, tile by tile, compacts the points that lie inside the box grown by the
// radius into LDS -- in index order, so ties still go to the lower index -- and only those are scored.  A point outside the grown box is
// farther than `radius` from every query of the workgroup.  Every neighbour with d^2 < radius^2 comes out exactly as k_knn reports it
// (same d^2 expression, same order); slots beyond the radius hold either a farther point or (inf, -1).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(KNN_BLOCK)
k_knn_radius(const float* __restrict__ points, int64_t point_stride, const int32_t* __restrict__ n_points,
             const float* __restrict__ queries, int64_t query_stride, const int32_t* __restrict__ n_queries,
             const int32_t* __restrict__ kk, int max_queries, float radius, float* __restrict__ d2_out, int32_t* __restrict__ idx_out) {
    __shared__ float tile[KNN_TILE * 3];
    __shared__ int tidx[KNN_TILE];
    __shared__ float box[6][KNN_BLOCK / WAVE];
    __shared__ int wcount[KNN_TILE / WAVE];
    const int b = blockIdx.y;
    const int nq = n_queries[b], np = n_points[b];
    if ((int)(blockIdx.x * KNN_BLOCK) >= nq) return;
    const int k = kk[b];
    const int q = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const bool active = q < nq;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const float* P = points + (size_t)b * point_stride;
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const float* Q = queries + (size_t)b * query_stride + (size_t)q * 3;
        qx = Q[0];
        qy = Q[1];
        qz = Q[2];
    }
    // bounding box of the workgroup's queries
    float lo[3] = {active ? qx : INFINITY, active ? qy : INFINITY, active ? qz : INFINITY};
    float hi[3] = {active ? qx : -INFINITY, active ? qy : -INFINITY, active ? qz : -INFINITY};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
        if (lane == 0) {
            box[a][wave] = lo[a];
            box[3 + a][wave] = hi[a];
        }
    }
    __syncthreads();
    const float grow = radius * 1.0001f + 1e-4f;          // conservative: rounding of the box arithmetic never drops a point inside the radius
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(fminf(box[a][0], box[a][1]), fminf(box[a][2], box[a][3])) - grow;
        hi[a] = fmaxf(fmaxf(box[3 + a][0], box[3 + a][1]), fmaxf(box[3 + a][2], box[3 + a][3])) + grow;
    }
    float bd[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        bd[j] = INFINITY;
        bi[j] = -1;
    }
    for (int t0 = 0; t0 < np; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, np - t0);
        // ---- compaction of the tile's points that fall into the grown box, in index order: point i = j * 256 + thread ----
        float px[KNN_TILE / KNN_BLOCK], py[KNN_TILE / KNN_BLOCK], pz[KNN_TILE / KNN_BLOCK];
        unsigned long long bal[KNN_TILE / KNN_BLOCK];
        __syncthreads();                                    // the previous tile's list is no longer read
#pragma unroll
        for (int j = 0; j < KNN_TILE / KNN_BLOCK; ++j) {
            const int i = j * KNN_BLOCK + threadIdx.x;
            bool in = false;
            if (i < cnt) {
                const float* pp = P + (size_t)(t0 + i) * 3;
                px[j] = pp[0];
                py[j] = pp[1];
                pz[j] = pp[2];
                in = px[j] >= lo[0] && px[j] <= hi[0] && py[j] >= lo[1] && py[j] <= hi[1] && pz[j] >= lo[2] && pz[j] <= hi[2];
            }
            bal[j] = __ballot(in);
            if (lane == 0) wcount[j * (KNN_BLOCK / WAVE) + wave] = __popcll(bal[j]);
        }
        __syncthreads();
        int n_in = 0;
#pragma unroll
        for (int j = 0; j < KNN_TILE / KNN_BLOCK; ++j) {
            int base = 0;
            for (int w = 0; w < j * (KNN_BLOCK / WAVE) + wave; ++w) base += wcount[w];
            if ((bal[j] >> lane) & 1ull) {
                const int o = base + __popcll(bal[j] & ((1ull << lane) - 1ull));
                tile[o * 3] = px[j];
                tile[o * 3 + 1] = py[j];
                tile[o * 3 + 2] = pz[j];
                tidx[o] = t0 + j * KNN_BLOCK + threadIdx.x;
            }
        }
        for (int w = 0; w < KNN_TILE / WAVE; ++w) n_in += wcount[w];
        __syncthreads();
        if (active) {
            for (int i = 0; i < n_in; ++i) {
                const float dx = qx - tile[i * 3], dy = qy - tile[i * 3 + 1], dz = qz - tile[i * 3 + 2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float s = xx + yy;
                const float d = s + zz;
                if (d < bd[K - 1]) {  // strict: equal distances keep the earlier (lower) index
                    bd[K - 1] = d;
                    bi[K - 1] = tidx[i];
#pragma unroll
                    for (int j = K - 1; j > 0; --j) {
                        if (bd[j] < bd[j - 1]) {
                            const float td = bd[j];
                            bd[j] = bd[j - 1];
                            bd[j - 1] = td;
                            const int ti = bi[j];
                            bi[j] = bi[j - 1];
                            bi[j - 1] = ti;
                        }
                    }
                }
            }
        }
    }
    if (active) {
        const size_t o = ((size_t)b * max_queries + q) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j < k) {
                d2_out[o + j] = bd[j];
                idx_out[o + j] = bi[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a7/a10 group statistics (7-vector); one block per group, members staged through LDS in chunks,
// centroid = sequential float64 sum (deterministic, == oracle mean_rows_f64)
// ------------------------------------------------------------------------------------------------
constexpr int GS_CHUNK = 1024;

__global__ void __launch_bounds__(256)
k_group_stats7(const float* __restrict__ rows_pos, const float* __restrict__ rows_dir, const float* __restrict__ rows_scale,
               int64_t n_cap, const int32_t* __restrict__ tok_slot, const int32_t* __restrict__ tok_row,
               const int32_t* __restrict__ grp_off, float cell_x, float cell_y, float cell_z, float* __restrict__ centroid,
               int32_t* __restrict__ cell, float* __restrict__ geom, float* __restrict__ inst_pos,
               const int32_t* __restrict__ grp_slot, const int32_t* __restrict__ grp_inst, int64_t m_cap) {
    __shared__ float sp[GS_CHUNK * 3];
    __shared__ double acc[3];
    __shared__ float cen[3];
    const int g = blockIdx.x;
    const int t0 = grp_off[g], t1 = grp_off[g + 1];
    const int n = t1 - t0;
    if (threadIdx.x < 3) acc[threadIdx.x] = 0.0;
    for (int c0 = 0; c0 < n; c0 += GS_CHUNK) {
        const int cnt = min(GS_CHUNK, n - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int t = t0 + c0 + i;
            const float* p = rows_pos + ((int64_t)tok_slot[t] * n_cap + tok_row[t]) * 3;
            sp[i * 3] = p[0];
            sp[i * 3 + 1] = p[1];
            sp[i * 3 + 2] = p[2];
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double a = acc[threadIdx.x];
            for (int i = 0; i < cnt; ++i) a = a + (double)sp[i * 3 + threadIdx.x];
            acc[threadIdx.x] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float c = (float)(acc[threadIdx.x] / (double)n);
        cen[threadIdx.x] = c;
        centroid[(size_t)g * 3 + threadIdx.x] = c;
        const float L = threadIdx.x == 0 ? cell_x : (threadIdx.x == 1 ? cell_y : cell_z);
        cell[(size_t)g * 3 + threadIdx.x] = (int32_t)floorf(c / L);
        if (inst_pos && grp_inst[g] >= 0) inst_pos[((int64_t)grp_slot[g] * m_cap + grp_inst[g]) * 3 + threadIdx.x] = c;
    }
    __syncthreads();
    const float cx = cen[0], cy = cen[1], cz = cen[2];
    for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const int64_t r = (int64_t)tok_slot[t] * n_cap + tok_row[t];
        const float x = rows_pos[r * 3], y = rows_pos[r * 3 + 1], z = rows_pos[r * 3 + 2];
        const float xx = x * x, yy = y * y, zz = z * z;
        const float s = xx + yy;
        const float d = rows_dir[r];
        float* o = geom + (size_t)t * 7;
        o[0] = x - cx;
        o[1] = y - cy;
        o[2] = z - cz;
        o[3] = sqrtf(s + zz);
        o[4] = sinf(d);
        o[5] = cosf(d);
        o[6] = rows_scale[r];
    }
}

// a11 zone statistics (4-vector); groups are small (instances in one 2 m cell): one wave per group
__global__ void k_group_stats4(const float* __restrict__ inst_pos, int64_t m_cap, const int32_t* __restrict__ tok_slot,
                               const int32_t* __restrict__ tok_inst, const int32_t* __restrict__ grp_off,
                               const int32_t* __restrict__ grp_mode, const int32_t* __restrict__ grp_slot,
                               const int32_t* __restrict__ grp_zone_row, int G, float cell_x, float cell_y, float cell_z,
                               float* __restrict__ geom, float* __restrict__ zone_pos, int64_t z_cap) {
    const int g = blockIdx.x;
    if (g >= G) return;
    const int t0 = grp_off[g], t1 = grp_off[g + 1], mode = grp_mode[g];
    const int lane = threadIdx.x;
    __shared__ float cen[3];
    auto member_pos = [&](int t, int a) -> float {
        const float p = inst_pos[((int64_t)tok_slot[t] * m_cap + tok_inst[t]) * 3 + a];
        if (mode == 0) return p;
        const float L = a == 0 ? cell_x : (a == 1 ? cell_y : cell_z);
        const float q = p / L;
        const float f = floorf(q) * L;
        return f + (L * 0.5f);
    };
    if (lane < 3) {
        double a = 0.0;
        for (int t = t0; t < t1; ++t) a = a + (double)member_pos(t, lane);
        const float c = (float)(a / (double)(t1 - t0));  // empty group -> 0/0 = NaN (mean of empty)
        cen[lane] = c;
        zone_pos[((int64_t)grp_slot[g] * z_cap + grp_zone_row[g]) * 3 + lane] = c;
    }
    __syncthreads();
    for (int t = t0 + lane; t < t1; t += blockDim.x) {
        const float x = member_pos(t, 0), y = member_pos(t, 1), z = member_pos(t, 2);
        const float xx = x * x, yy = y * y, zz = z * z;
        const float s = xx + yy;
        float* o = geom + (size_t)t * 4;
        o[0] = x - cen[0];
        o[1] = y - cen[1];
        o[2] = z - cen[2];
        o[3] = sqrtf(s + zz);
    }
}

// ------------------------------------------------------------------------------------------------
// row movers: one wave per 768-wide row
// ------------------------------------------------------------------------------------------------
__global__ void k_gather_fts(const uint16_t* __restrict__ rows_fts, int64_t n_cap, const int32_t* __restrict__ tok_slot,
                             const int32_t* __restrict__ tok_row, int T, float* __restrict__ out) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (t >= T) return;
    const uint2* src = reinterpret_cast<const uint2*>(rows_fts + ((int64_t)tok_slot[t] * n_cap + tok_row[t]) * FTS);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)t * FTS);
    for (int i = lane; i < FTS / 4; i += WAVE) {
        const uint2 v = src[i];
        const __half2 a = *reinterpret_cast<const __half2*>(&v.x), b = *reinterpret_cast<const __half2*>(&v.y);
        const float2 fa = __half22float2(a), fb = __half22float2(b);
        dst[i] = make_float4(fa.x, fa.y, fb.x, fb.y);
    }
}

__global__ void k_gather_rows(const float* __restrict__ pool, int64_t cap, int D, const int32_t* __restrict__ slot,
                              const int32_t* __restrict__ row, int T, float* __restrict__ out) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (t >= T) return;
    const float* src = pool + ((int64_t)slot[t] * cap + row[t]) * D;
    float* dst = out + (size_t)t * D;
    if ((D & 3) == 0) {
        for (int i = lane; i < D / 4; i += WAVE) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = lane; i < D; i += WAVE) dst[i] = src[i];
    }
}

__global__ void k_scatter_rows(float* __restrict__ pool, int64_t cap, int D, const int32_t* __restrict__ slot,
                               const int32_t* __restrict__ row, int T, const float* __restrict__ src,
                               const int32_t* __restrict__ src_row) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (t >= T || row[t] < 0) return;                          // row -1: nothing to store for this source row (device-planned scatters)
    float* dst = pool + ((int64_t)slot[t] * cap + row[t]) * D;
    const float* s = src + (size_t)(src_row ? src_row[t] : t) * D;
    if ((D & 3) == 0) {
        for (int i = lane; i < D / 4; i += WAVE) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(s)[i];
    } else {
        for (int i = lane; i < D; i += WAVE) dst[i] = s[i];
    }
}

__global__ void k_fill_rows(float* __restrict__ pool, int64_t cap, int D, const int32_t* __restrict__ slot,
                            const int32_t* __restrict__ row, int T, float value) {
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (t >= T) return;
    float* dst = pool + ((int64_t)slot[t] * cap + row[t]) * D;
    for (int i = lane; i < D; i += WAVE) dst[i] = value;
}

// a9 merge-discriminator input rows
__global__ void k_merge_input(const float* __restrict__ inst_fts, const float* __restrict__ inst_pos, int64_t m_cap,
                              const float* __restrict__ new_fts, const float* __restrict__ new_pos,
                              const int32_t* __restrict__ pair_slot, const int32_t* __restrict__ pair_inst,
                              const int32_t* __restrict__ pair_new, int R, float* __restrict__ out) {
    const int r = blockIdx.x;
    if (r >= R) return;
    const int64_t ii = (int64_t)pair_slot[r] * m_cap + pair_inst[r];
    const int nn = pair_new[r];
    float* o = out + (size_t)r * (2 * FTS + 3);
    const float* a = inst_fts + ii * FTS;
    const float* b = new_fts + (size_t)nn * FTS;
    for (int i = threadIdx.x; i < FTS; i += blockDim.x) {
        o[i] = a[i];
        o[FTS + i] = b[i];
    }
    if (threadIdx.x < 3) o[2 * FTS + threadIdx.x] = new_pos[(size_t)nn * 3 + threadIdx.x] - inst_pos[ii * 3 + threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// a12 agent-frame transform + radius filter + ORDERED compaction; one block (256) per env
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_agent_frame_compact(const float* __restrict__ pool_pos, const float* __restrict__ pool_fts, int64_t cap,
                      const int32_t* __restrict__ slot, const int32_t* __restrict__ ids, const int32_t* __restrict__ n_ids,
                      int max_ids, const d3d_pose* __restrict__ pose, float radius, float* __restrict__ rel,
                      float* __restrict__ fts, int32_t* __restrict__ kept_ids, int32_t* __restrict__ count) {
    __shared__ int wave_cnt[4];
    __shared__ int run_base;
    const int e = blockIdx.x;
    const int n = n_ids[e];
    const d3d_pose ps = pose[e];
    const int64_t base = (int64_t)slot[e] * cap;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    if (threadIdx.x == 0) run_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        bool keep = false;
        float rx = 0, ry = 0, pz = 0;
        int id = -1;
        if (i < n) {
            id = ids[(size_t)e * max_ids + i];
            const float* p = pool_pos + (base + id) * 3;
            const float px = p[0] - ps.cam[0], py = p[1] - ps.cam[1];
            pz = p[2] - ps.cam[2];
            const float a = px * ps.cos_nh, b = py * ps.sin_nh;
            rx = a - b;
            const float c = px * ps.sin_nh, d = py * ps.cos_nh;
            ry = c + d;
            const float xx = rx * rx, yy = ry * ry, zz = pz * pz;
            const float s = xx + yy;
            keep = sqrtf(s + zz) <= radius;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = run_base;
        for (int w = 0; w < wv; ++w) off += wave_cnt[w];
        if (keep) {
            const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
            float* r = rel + ((size_t)e * max_ids + o) * 3;
            r[0] = rx;
            r[1] = ry;
            r[2] = pz;
            kept_ids[(size_t)e * max_ids + o] = id;
        }
        __syncthreads();
        if (threadIdx.x == 0) run_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int kept = run_base;
    if (threadIdx.x == 0) count[e] = kept;
    // gather the survivors' feature rows (one wave per row)
    for (int o = wv; o < kept; o += 4) {
        const int id = kept_ids[(size_t)e * max_ids + o];
        const float4* src = reinterpret_cast<const float4*>(pool_fts + (base + id) * FTS);
        float4* dst = reinterpret_cast<float4*>(fts + ((size_t)e * max_ids + o) * FTS);
        for (int i = lane; i < FTS / 4; i += WAVE) dst[i] = src[i];
    }
}

}  // namespace

// ================================================================================================
// C ABI launchers
// ================================================================================================
extern "C" {

int32_t d3d_device_info(int32_t* n_cu, int32_t* wave_size, int32_t* lds_bytes) {
    int dev = 0;
    D3D_HIP(hipGetDevice(&dev));
    hipDeviceProp_t pr;
    D3D_HIP(hipGetDeviceProperties(&pr, dev));
    if (n_cu) *n_cu = pr.multiProcessorCount;
    if (wave_size) *wave_size = pr.warpSize;
    if (lds_bytes) *lds_bytes = (int32_t)pr.sharedMemPerBlock;
    return D3D_OK;
}

int32_t d3d_preprocess_depth(const float* depth, float* out, int32_t B, int32_t H, int32_t W, float lo, float hi, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return D3D_OK;
    const float range = (float)((double)hi - (double)lo), lo100 = (float)((double)lo * 100.0);
    dim3 grid((W + PD_COLS - 1) / PD_COLS, B);
    hipLaunchKernelGGL(k_preprocess_depth, grid, dim3(PD_COLS * PD_GROUPS), 0, (hipStream_t)stream, depth, out, H, W, range, lo100);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_resize_nearest_preprocess(const float* depth, float* out, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w,
                                      float lo, float hi, void* stream) {
    if (B <= 0) return D3D_OK;
    const float range = (float)((double)hi - (double)lo), lo100 = (float)((double)lo * 100.0);
    const size_t sh = (size_t)(h * w + w) * sizeof(float);
    hipLaunchKernelGGL(k_resize_preprocess, dim3(B), dim3(256), sh, (hipStream_t)stream, depth, out, H, W, h, w, range, lo100);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_unproject_append(const float* depth24, const d3d_pose* pose, const int32_t* slot, const int32_t* row_base,
                             int32_t n_env, int32_t P, int32_t W, const float* tan_xy, const float* tan_z, const float* dir0,
                             float th, float* rows_pos, float* rows_dir, float* rows_scale, int64_t n_cap, void* stream) {
    if (n_env <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_unproject_append, dim3(n_env), dim3(576), 0, (hipStream_t)stream, depth24, pose, slot, row_base, P, W,
                       tan_xy, tan_z, dir0, th, rows_pos, rows_dir, rows_scale, n_cap);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_patch_segm_from_masks(const uint8_t* masks, const int32_t* mask_off, int32_t n_img, int32_t max_masks, int32_t H, int32_t W,
                                  int32_t h, int32_t w, int32_t* segm, int32_t* n_seg, void* stream) {
    if (n_img <= 0) return D3D_OK;
    if (max_masks > SEGM_MAX_MASKS || h * w > 4096 || H <= 0 || W <= 0) {
        d3d_set_error_("d3d_patch_segm_from_masks: at most 4096 masks per image and 4096 patches");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_patch_segm, dim3(n_img), dim3(256), (size_t)h * w * sizeof(int32_t), (hipStream_t)stream, masks, mask_off, H, W, h, w,
                       segm, n_seg);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_unproject_pinhole_append(const float* depth, int32_t Hd, int32_t Wd, const d3d_pinhole_unproject* cams, const int32_t* slot,
                                     const int32_t* row_base, int32_t n_env, int32_t h, int32_t w, int32_t input_width, float* rows_pos,
                                     float* rows_dir, float* rows_scale, int64_t n_cap, void* stream) {
    if (n_env <= 0) return D3D_OK;
    if (Hd <= 0 || Wd <= 0 || h <= 0 || w <= 0) {
        d3d_set_error_("d3d_unproject_pinhole_append: empty image or grid");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_unproject_pinhole_append, dim3(n_env), dim3(256), 0, (hipStream_t)stream, depth, Hd, Wd, cams, slot, row_base, h, w,
                       input_width, rows_pos, rows_dir, rows_scale, n_cap);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_append_fts(const void* grid, int32_t grid_is_f16, const int32_t* slot, const int32_t* row_base, int32_t n_env,
                       int32_t P, uint16_t* rows_fts, int64_t n_cap, void* stream) {
    if (n_env <= 0) return D3D_OK;
    const int waves = n_env * P;
    const dim3 grid1((waves * 64 + 255) / 256);
    if (grid_is_f16)
        hipLaunchKernelGGL(k_append_fts<true>, grid1, dim3(256), 0, (hipStream_t)stream, grid, slot, row_base, n_env, P, rows_fts, n_cap);
    else
        hipLaunchKernelGGL(k_append_fts<false>, grid1, dim3(256), 0, (hipStream_t)stream, grid, slot, row_base, n_env, P, rows_fts, n_cap);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_patch_3d_info(const float* depth24, int32_t N, int32_t P, int32_t W, const float* tan_xy, const float* tan_z,
                          const float* dir0, float th, float* rel_x, float* rel_y, float* rel_z, float* dir, float* scale,
                          void* stream) {
    if (N <= 0) return D3D_OK;
    const size_t n = (size_t)N * P;
    hipLaunchKernelGGL(k_patch_3d_info, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, depth24, N, P, W, tan_xy, tan_z,
                       dir0, th, rel_x, rel_y, rel_z, dir, scale);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_frustum_cull(float* rows_pos, uint16_t* rows_fts, float* rows_dir, float* rows_scale, int64_t n_cap,
                         const int32_t* slot, const int32_t* n_rows, int32_t n_env, int32_t max_rows, const float* depth,
                         int32_t Hd, int32_t Wd, const d3d_pose* pose, float fx, float fy, float cx, float cy, float near_,
                         float far_, float slack, int32_t* hits, int32_t* n_hits, int32_t hit_cap, uint8_t* mask, void* stream) {
    if (n_env <= 0 || max_rows <= 0) return D3D_OK;
    FrustumCam cm{fx, fy, cx, cy, near_, far_, slack, Hd, Wd};
    dim3 grid((max_rows + 255) / 256, n_env);
    hipLaunchKernelGGL(k_frustum_cull<d3d_pose>, grid, dim3(256), 0, (hipStream_t)stream, rows_pos, rows_fts, rows_dir, rows_scale, n_cap,
                       slot, n_rows, depth, pose, cm, hits, n_hits, hit_cap, mask);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_frustum_cull_pinhole(float* rows_pos, uint16_t* rows_fts, float* rows_dir, float* rows_scale, int64_t n_cap,
                                 const int32_t* slot, const int32_t* n_rows, int32_t n_env, int32_t max_rows, const float* depth,
                                 int32_t Hd, int32_t Wd, const d3d_pinhole_view* views, float near_, float far_, float slack,
                                 int32_t* hits, int32_t* n_hits, int32_t hit_cap, uint8_t* mask, void* stream) {
    if (n_env <= 0 || max_rows <= 0) return D3D_OK;
    FrustumCam cm{0.f, 0.f, 0.f, 0.f, near_, far_, slack, Hd, Wd};
    dim3 grid((max_rows + 255) / 256, n_env);
    hipLaunchKernelGGL(k_frustum_cull<d3d_pinhole_view>, grid, dim3(256), 0, (hipStream_t)stream, rows_pos, rows_fts, rows_dir,
                       rows_scale, n_cap, slot, n_rows, depth, views, cm, hits, n_hits, hit_cap, mask);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_frustum_mask_pinhole(const float* points, int64_t n, const float* depth, int32_t Hd, int32_t Wd,
                                 const d3d_pinhole_view* view_h, float near_, float far_, float slack, uint8_t* mask, void* stream) {
    if (n <= 0) return D3D_OK;
    FrustumCam cm{0.f, 0.f, 0.f, 0.f, near_, far_, slack, Hd, Wd};
    hipLaunchKernelGGL(k_frustum_mask<d3d_pinhole_view>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, n,
                       depth, *view_h, cm, mask);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_frustum_mask(const float* points, int64_t n, const float* depth, int32_t Hd, int32_t Wd, const d3d_pose* pose_h,
                         float fx, float fy, float cx, float cy, float near_, float far_, float slack, uint8_t* mask, void* stream) {
    if (n <= 0) return D3D_OK;
    FrustumCam cm{fx, fy, cx, cy, near_, far_, slack, Hd, Wd};
    hipLaunchKernelGGL(k_frustum_mask<d3d_pose>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, n, depth,
                       *pose_h, cm, mask);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_knn(const float* points, int64_t point_stride, const int32_t* n_points, const float* queries, int64_t query_stride,
                const int32_t* n_queries, const int32_t* k, int32_t n_batch, int32_t max_queries, int32_t k_max, float* d2,
                int32_t* idx, void* stream) {
    if (n_batch <= 0 || max_queries <= 0) return D3D_OK;
    dim3 grid((max_queries + KNN_BLOCK - 1) / KNN_BLOCK, n_batch);
#define D3D_KNN_CASE(KK)                                                                                                   \
    case KK:                                                                                                               \
        hipLaunchKernelGGL(k_knn<KK>, grid, dim3(KNN_BLOCK), 0, (hipStream_t)stream, points, point_stride, n_points, queries, \
                           query_stride, n_queries, k, max_queries, d2, idx);                                             \
        break;
    switch (k_max) {
        D3D_KNN_CASE(1)
        D3D_KNN_CASE(2)
        D3D_KNN_CASE(4)
        D3D_KNN_CASE(8)
        default:
            d3d_set_error_("d3d_knn: k_max must be 1, 2, 4 or 8");
            return D3D_EINVAL;
    }
#undef D3D_KNN_CASE
    D3D_LAUNCH_CHECK();
}

int32_t d3d_knn_chunked(const float* points, int64_t point_stride, const int32_t* n_points, const float* queries, int64_t query_stride,
                        const int32_t* n_queries, const int32_t* k, int32_t n_batch, int32_t max_queries, int32_t k_max, int32_t n_chunks,
                        float* ws_d2, int32_t* ws_idx, float* d2, int32_t* idx, void* stream) {
    if (n_batch <= 0 || max_queries <= 0) return D3D_OK;
    if (n_chunks < 1 || n_chunks > 4096 || !ws_d2 || !ws_idx) {
        d3d_set_error_("d3d_knn_chunked: 1 <= n_chunks <= 4096 and workspaces of n_batch * max_queries * n_chunks * k_max elements");
        return D3D_EINVAL;
    }
    dim3 grid((max_queries + KNN_BLOCK - 1) / KNN_BLOCK, n_batch, n_chunks), grid_m((max_queries + KNN_BLOCK - 1) / KNN_BLOCK, n_batch);
#define D3D_KNNC_CASE(KK)                                                                                                              \
    case KK:                                                                                                                          \
        hipLaunchKernelGGL(k_knn_chunk<KK>, grid, dim3(KNN_BLOCK), 0, (hipStream_t)stream, points, point_stride, n_points, queries,    \
                           query_stride, n_queries, max_queries, n_chunks, ws_d2, ws_idx);                                             \
        hipLaunchKernelGGL(k_knn_merge<KK>, grid_m, dim3(KNN_BLOCK), 0, (hipStream_t)stream, ws_d2, ws_idx, n_queries, k, max_queries, \
                           n_chunks, d2, idx);                                                                                         \
        break;
    switch (k_max) {
        D3D_KNNC_CASE(1)
        D3D_KNNC_CASE(2)
        D3D_KNNC_CASE(4)
        D3D_KNNC_CASE(8)
        default:
            d3d_set_error_("d3d_knn_chunked: k_max must be 1, 2, 4 or 8");
            return D3D_EINVAL;
    }
#undef D3D_KNNC_CASE
    D3D_LAUNCH_CHECK();
}

int32_t d3d_knn_radius(const float* points, int64_t point_stride, const int32_t* n_points, const float* queries, int64_t query_stride,
                       const int32_t* n_queries, const int32_t* k, int32_t n_batch, int32_t max_queries, int32_t k_max, float radius, float* d2,
                       int32_t* idx, void* stream) {
    if (n_batch <= 0 || max_queries <= 0) return D3D_OK;
    if (!(radius > 0.f)) {
        d3d_set_error_("d3d_knn_radius: radius must be positive");
        return D3D_EINVAL;
    }
    dim3 grid((max_queries + KNN_BLOCK - 1) / KNN_BLOCK, n_batch);
#define D3D_KNNR_CASE(KK)                                                                                                         \
    case KK:                                                                                                                      \
        hipLaunchKernelGGL(k_knn_radius<KK>, grid, dim3(KNN_BLOCK), 0, (hipStream_t)stream, points, point_stride, n_points, queries, \
                           query_stride, n_queries, k, max_queries, radius, d2, idx);                                             \
        break;
    switch (k_max) {
        D3D_KNNR_CASE(1)
        D3D_KNNR_CASE(2)
        D3D_KNNR_CASE(4)
        D3D_KNNR_CASE(8)
        default:
            d3d_set_error_("d3d_knn_radius: k_max must be 1, 2, 4 or 8");
            return D3D_EINVAL;
    }
#undef D3D_KNNR_CASE
    D3D_LAUNCH_CHECK();
}

int32_t d3d_group_stats7(const float* rows_pos, const float* rows_dir, const float* rows_scale, int64_t n_cap,
                         const int32_t* tok_slot, const int32_t* tok_row, const int32_t* grp_off, int32_t G, int32_t T,
                         float cell_x, float cell_y, float cell_z, float* centroid, int32_t* cell, float* geom, float* inst_pos,
                         const int32_t* grp_slot, const int32_t* grp_inst, int64_t m_cap, void* stream) {
    (void)T;
    if (G <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_group_stats7, dim3(G), dim3(256), 0, (hipStream_t)stream, rows_pos, rows_dir, rows_scale, n_cap, tok_slot,
                       tok_row, grp_off, cell_x, cell_y, cell_z, centroid, cell, geom, inst_pos, grp_slot, grp_inst, m_cap);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_group_stats4(const float* inst_pos, int64_t m_cap, const int32_t* tok_slot, const int32_t* tok_inst,
                         const int32_t* grp_off, const int32_t* grp_mode, const int32_t* grp_slot, const int32_t* grp_zone_row,
                         int32_t G, int32_t T, float cell_x, float cell_y, float cell_z, float* geom, float* zone_pos,
                         int64_t z_cap, void* stream) {
    (void)T;
    if (G <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_group_stats4, dim3(G), dim3(64), 0, (hipStream_t)stream, inst_pos, m_cap, tok_slot, tok_inst, grp_off,
                       grp_mode, grp_slot, grp_zone_row, G, cell_x, cell_y, cell_z, geom, zone_pos, z_cap);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_gather_fts(const uint16_t* rows_fts, int64_t n_cap, const int32_t* tok_slot, const int32_t* tok_row, int32_t T,
                       float* out, void* stream) {
    if (T <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_gather_fts, dim3((T * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows_fts, n_cap, tok_slot, tok_row, T, out);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_gather_rows_f32(const float* pool, int64_t cap, int32_t D, const int32_t* slot, const int32_t* row, int32_t T,
                            float* out, void* stream) {
    if (T <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3((T * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool, cap, D, slot, row, T, out);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_scatter_rows_f32(float* pool, int64_t cap, int32_t D, const int32_t* slot, const int32_t* row, int32_t T,
                             const float* src, const int32_t* src_row, void* stream) {
    if (T <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3((T * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool, cap, D, slot, row, T, src, src_row);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_fill_rows_f32(float* pool, int64_t cap, int32_t D, const int32_t* slot, const int32_t* row, int32_t T, float value,
                          void* stream) {
    if (T <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_fill_rows, dim3((T * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool, cap, D, slot, row, T, value);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_merge_input(const float* inst_fts, const float* inst_pos, int64_t m_cap, const float* new_fts, const float* new_pos,
                        const int32_t* pair_slot, const int32_t* pair_inst, const int32_t* pair_new, int32_t R, float* out,
                        void* stream) {
    if (R <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_merge_input, dim3(R), dim3(256), 0, (hipStream_t)stream, inst_fts, inst_pos, m_cap, new_fts, new_pos,
                       pair_slot, pair_inst, pair_new, R, out);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_agent_frame_compact(const float* pool_pos, const float* pool_fts, int64_t cap, const int32_t* slot,
                                const int32_t* ids, const int32_t* n_ids, int32_t n_env, int32_t max_ids, const d3d_pose* pose,
                                float radius, float* rel, float* fts, int32_t* kept_ids, int32_t* count, void* stream) {
    if (n_env <= 0) return D3D_OK;
    hipLaunchKernelGGL(k_agent_frame_compact, dim3(n_env), dim3(256), 0, (hipStream_t)stream, pool_pos, pool_fts, cap, slot, ids,
                       n_ids, max_ids, pose, radius, rel, fts, kept_ids, count);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
