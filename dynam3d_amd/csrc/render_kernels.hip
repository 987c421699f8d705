// render_kernels.hip -- novel-view feature rendering of the Pretrain Feature_Fields:
//   d3d_rays_habitat     12x12 ray grid x 501 depths, float64 like the reference's numpy, rounded once  (PRE-FF:408-422, 524-530)
//   d3d_ray_topk         sqrt / radius mask / importance = 1/sum(dist) / top-8 per ray, one wave per ray (PRE-FF:543-556)
//   d3d_render_embed     gather 4 neighbour features + 6-d relative geometry -> Linear(6,768)+LN -> fp16 add     (PRE-FF:586-616, 481)
//   d3d_composite        softplus density, alpha compositing over the 8 samples, L2-normalised feature, depth (PRE-FF:446-474)
// The k=4 KNN over the 72 144 ray samples is d3d_knn (geometry_kernels.hip); the 768-wide MLPs are d3d_gemm_nt launches.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int WAVE = 64;
constexpr int FTS = D3D_FTS_DIM;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pose64: (cx, cy, cz, cos h, sin h) in double, as Python computes them
__global__ void k_rays_habitat(const double* __restrict__ rel_y, const float* __restrict__ tan_xy, const float* __restrict__ tan_z,
                               const double* __restrict__ pose64, int R, int N, float* __restrict__ ray) {
    const int e = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * N) return;
    const int r = (int)(i / N), n = (int)(i % N);
    const double* ps = pose64 + e * 5;
    const double y = rel_y[n];
    const double x = y * (double)tan_xy[r], z = y * (double)tan_z[r];
    const double xc = x * ps[3], ys = y * ps[4], xs = x * ps[4], yc = y * ps[3];
    float* o = ray + ((int64_t)e * R * N + i) * 3;
    o[0] = (float)((xc - ys) + ps[0]);
    o[1] = (float)((xs + yc) + ps[1]);
    o[2] = (float)(z + ps[2]);
}

// get_rays (PRE-FF:390-405) + world transform of the intrinsics mode (PRE-FF:534): Open3D unprojects N constant-depth images
// in double, then ray = R @ rel + T in double, rounded to float32 once.  z (N) f64 = float32(near + spacing*(i+1)) widened;
// cam (n_env,16) f64 = fx, fy, cx, cy, R[9] row-major, T[3].  Ray r = pixel (row r / W, col r % W).
__global__ void k_rays_pinhole(const double* __restrict__ zs, const double* __restrict__ cam, int R, int N, int W,
                               float* __restrict__ ray) {
    const int e = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * N) return;
    const int r = (int)(i / N), n = (int)(i % N);
    const double* c = cam + e * 16;
    const double z = zs[n];
    const double x = ((double)(r % W) - c[2]) * z / c[0];
    const double y = ((double)(r / W) - c[3]) * z / c[1];
    const double* Rm = c + 4;
    const double* T = c + 13;
    float* o = ray + ((int64_t)e * R * N + i) * 3;
    o[0] = (float)(((Rm[0] * x + Rm[1] * y) + Rm[2] * z) + T[0]);
    o[1] = (float)(((Rm[3] * x + Rm[4] * y) + Rm[5] * z) + T[1]);
    o[2] = (float)(((Rm[6] * x + Rm[7] * y) + Rm[8] * z) + T[2]);
}

// one wave per ray.  d2/idx: (n_rays*N, KK).  Importance = 1/sum_j min(sqrt(d2_j) >= radius ? radius : sqrt(d2_j)).
// top-n_imp by (importance desc, sample index asc).  Writes topk (n_imp), masked neighbour ids of the chosen samples
// (n_imp, KK) and the number of samples that have at least one neighbour inside the radius.
template <int KK>
__global__ void __launch_bounds__(256)
k_ray_topk(const float* __restrict__ d2, const int32_t* __restrict__ idx, int n_rays, int N, float radius, int n_imp,
           int32_t* __restrict__ topk, int32_t* __restrict__ sidx, int32_t* __restrict__ n_ranked) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    constexpr int PER = 8;                                   // samples per lane (N <= 512)
    float dens[PER];
    int ranked = 0;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const int n = s * WAVE + lane;
        dens[s] = -INFINITY;
        if (n < N) {
            const float* dd = d2 + ((int64_t)ray * N + n) * KK;
            float tmp = 0.f;
            bool any = false;
#pragma unroll
            for (int j = 0; j < KK; ++j) {
                float d = sqrtf(dd[j]);
                if (d >= radius) d = radius; else any = true;
                tmp = j == 0 ? d : tmp + d;
            }
            dens[s] = 1.0f / tmp;
            ranked += any ? 1 : 0;
        }
    }
    ranked = (int)wsum((float)ranked);
    if (lane == 0) n_ranked[ray] = ranked;
    for (int t = 0; t < n_imp; ++t) {
        // lane-local best (lowest sample index wins ties: s ascending => n ascending for a fixed lane)
        float bv = -INFINITY;
        int bn = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < PER; ++s) {
            const int n = s * WAVE + lane;
            if (dens[s] > bv) { bv = dens[s]; bn = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int on = __shfl_xor(bn, o);
            if (ov > bv || (ov == bv && on < bn)) { bv = ov; bn = on; }
        }
        if (lane == 0) topk[(int64_t)ray * n_imp + t] = bn;
        if ((bn & (WAVE - 1)) == lane) {
#pragma unroll
            for (int s = 0; s < PER; ++s)
                if (s == (bn >> 6)) dens[s] = -INFINITY;
        }
        if (lane < KK) {
            const int64_t q = ((int64_t)ray * N + bn) * KK + lane;
            sidx[((int64_t)ray * n_imp + t) * KK + lane] = sqrtf(d2[q]) >= radius ? -1 : idx[q];
        }
    }
}

// One wave per (sample, neighbour): 6-vector (quirk R1), Linear(6,768)+bias, LayerNorm(eps), fp16; + fp16 feature; fp16 out.
__global__ void __launch_bounds__(256)
k_render_embed(const float* __restrict__ rows_pos, const float* __restrict__ rows_dir, const float* __restrict__ rows_scale,
               const uint16_t* __restrict__ rows_fts, int64_t n_cap, const int32_t* __restrict__ ray_slot,
               const float* __restrict__ ray_xyz, const int32_t* __restrict__ topk, const int32_t* __restrict__ sidx,
               const float* __restrict__ pose /* (n_rays-wise env pose: cos(-h), sin(-h), heading) via ray_env */,
               const int32_t* __restrict__ ray_env, const float* __restrict__ rel_direction, int R, int N, int n_imp, int KK,
               float far_, const float* __restrict__ w6, const float* __restrict__ b6, const float* __restrict__ ln_w,
               const float* __restrict__ ln_b, float eps, int64_t n_items, uint16_t* __restrict__ s16, float* __restrict__ geom6,
               float* __restrict__ sample_xyz) {
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // (ray, t, j)
    const int lane = threadIdx.x & 63;
    if (item >= n_items) return;
    const int j = (int)(item % KK);
    const int64_t st = item / KK;                                        // ray * n_imp + t
    const int64_t ray = st / n_imp;
    const int e = ray_env[ray];
    const float c = pose[e * 3], s = pose[e * 3 + 1], heading = pose[e * 3 + 2];
    const int n = topk[st];
    const float* sp = ray_xyz + ((int64_t)ray * N + n) * 3;
    const int id = sidx[st * KK + j];
    float g[6];
    const uint16_t* frow = nullptr;
    if (id >= 0) {
        const int64_t r = (int64_t)ray_slot[ray] * n_cap + id;
        const float dx = rows_pos[r * 3] - sp[0], dy = rows_pos[r * 3 + 1] - sp[1], dz = rows_pos[r * 3 + 2] - sp[2];
        const float a = dx * c, b = dy * s;
        const float xr = a - b;
        const float a2 = xr * s, b2 = dy * c;                            // quirk R1: rotated x feeds y'
        g[0] = xr; g[1] = a2 + b2; g[2] = dz;
        const float pd = rows_dir[r] - heading;
        const float ang = pd - rel_direction[ray % R];
        g[3] = sinf(ang); g[4] = cosf(ang); g[5] = rows_scale[r];
        frow = rows_fts + r * FTS;
    } else {
        g[0] = far_; g[1] = far_; g[2] = far_; g[3] = 0.f; g[4] = 0.f; g[5] = 0.f;
    }
    if (j == 0 && lane < 3 && sample_xyz) sample_xyz[st * 3 + lane] = sp[lane];
    if (geom6 && lane < 6) geom6[item * 6 + lane] = g[lane];
    float y[12];
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int ch = q * WAVE + lane;
        const float* w = w6 + ch * 6;
        float acc = b6[ch];
#pragma unroll
        for (int d = 0; d < 6; ++d) acc += w[d] * g[d];
        y[q] = acc;
        sum += acc;
    }
    const float mean = wsum(sum) / (float)FTS;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const float d = y[q] - mean;
        var += d * d;
    }
    const float rstd = rsqrtf(wsum(var) / (float)FTS + eps);
    uint16_t* out = s16 + st * (int64_t)(KK * FTS) + j * FTS;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const int ch = q * WAVE + lane;
        const __half gh = __float2half_rn((y[q] - mean) * rstd * ln_w[ch] + ln_b[ch]);
        const __half eh = frow ? *reinterpret_cast<const __half*>(frow + ch) : __float2half_rn(0.f);
        const __half r = __float2half_rn(__half2float(eh) + __half2float(gh));      // fp16 add (PRE-FF:481-483)
        out[ch] = *reinterpret_cast<const uint16_t*>(&r);
    }
}

// one wave per ray: the 8 samples sorted by sample index, transmittance over them only (empty bins have alpha == 0 and
// contribute exactly 1.0f to the cumprod), weighted feature sum, L2 normalisation, expected depth.
__global__ void __launch_bounds__(256)
k_composite(const uint16_t* __restrict__ feat, int64_t ldf, const uint16_t* __restrict__ dens, int64_t ldd, const float* __restrict__ rel_dist /*N, fp16-rounded*/,
            const int32_t* __restrict__ topk, int n_rays, int N, int n_imp, float* __restrict__ fmap, float* __restrict__ depth) {
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    // every lane evaluates the (<= 16) sample weights redundantly
    float w[16];
    float wtot = 0.f, dsum = 0.f;
    for (int t = 0; t < n_imp; ++t) {
        const int n = topk[(int64_t)ray * n_imp + t];
        const float sig = __half2float(*reinterpret_cast<const __half*>(dens + ((int64_t)ray * n_imp + t) * ldd));
        const float sp = sig > 20.f ? sig : log1pf(expf(sig));                           // F.softplus (threshold 20)
        const float dist = n + 1 < N ? fabsf(rel_dist[n + 1] - rel_dist[n]) : 1e10f;
        const float alpha = 1.0f - expf(-fmaxf(sp, 0.f) * dist);
        float T = 1.0f;
        for (int u = 0; u < n_imp; ++u) {
            const int m = topk[(int64_t)ray * n_imp + u];
            if (m < n) {
                const float sg = __half2float(*reinterpret_cast<const __half*>(dens + ((int64_t)ray * n_imp + u) * ldd));
                const float su = sg > 20.f ? sg : log1pf(expf(sg));
                const float du = m + 1 < N ? fabsf(rel_dist[m + 1] - rel_dist[m]) : 1e10f;
                T *= (1.0f - (1.0f - expf(-fmaxf(su, 0.f) * du))) + 1e-10f;
            }
        }
        w[t] = alpha * T;
        wtot += w[t];
        dsum += w[t] * rel_dist[n];
    }
    float acc[12];
    float nrm = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    for (int t = 0; t < n_imp; ++t) {
        const uint16_t* f = feat + ((int64_t)ray * n_imp + t) * ldf;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] += w[t] * __half2float(*reinterpret_cast<const __half*>(f + q * WAVE + lane));
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) nrm += acc[q] * acc[q];
    const float inv = 1.0f / fmaxf(sqrtf(wsum(nrm)), 1e-7f);
#pragma unroll
    for (int q = 0; q < 12; ++q) fmap[(int64_t)ray * FTS + q * WAVE + lane] = acc[q] * inv;
    if (lane == 0) depth[ray] = dsum / fmaxf(wtot, 1e-7f);
}

}  // namespace

extern "C" {

int32_t d3d_rays_habitat(const double* rel_y, const float* tan_xy, const float* tan_z, const double* pose64, int32_t n_env, int32_t R,
                         int32_t N, float* ray_xyz, void* stream) {
    if (n_env <= 0) return D3D_OK;
    dim3 grid((unsigned)(((int64_t)R * N + 255) / 256), n_env);
    hipLaunchKernelGGL(k_rays_habitat, grid, dim3(256), 0, (hipStream_t)stream, rel_y, tan_xy, tan_z, pose64, R, N, ray_xyz);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_rays_pinhole(const double* z, const double* cam16, int32_t n_env, int32_t view_h, int32_t view_w, int32_t N, float* ray_xyz,
                         void* stream) {
    if (n_env <= 0) return D3D_OK;
    const int R = view_h * view_w;
    dim3 grid((unsigned)(((int64_t)R * N + 255) / 256), n_env);
    hipLaunchKernelGGL(k_rays_pinhole, grid, dim3(256), 0, (hipStream_t)stream, z, cam16, R, N, view_w, ray_xyz);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_ray_topk(const float* d2, const int32_t* idx, int32_t n_rays, int32_t N, int32_t k, float radius, int32_t n_imp,
                     int32_t* topk, int32_t* sidx, int32_t* n_ranked, void* stream) {
    if (n_rays <= 0) return D3D_OK;
    if (N > 512 || n_imp > 16 || (k != 4 && k != 2 && k != 1 && k != 8)) {
        d3d_set_error_("d3d_ray_topk: need N <= 512, n_imp <= 16, k in {1,2,4,8}");
        return D3D_EINVAL;
    }
    dim3 grid((n_rays + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL(k_ray_topk<1>, grid, block, 0, s, d2, idx, n_rays, N, radius, n_imp, topk, sidx, n_ranked); break;
        case 2: hipLaunchKernelGGL(k_ray_topk<2>, grid, block, 0, s, d2, idx, n_rays, N, radius, n_imp, topk, sidx, n_ranked); break;
        case 4: hipLaunchKernelGGL(k_ray_topk<4>, grid, block, 0, s, d2, idx, n_rays, N, radius, n_imp, topk, sidx, n_ranked); break;
        default: hipLaunchKernelGGL(k_ray_topk<8>, grid, block, 0, s, d2, idx, n_rays, N, radius, n_imp, topk, sidx, n_ranked); break;
    }
    D3D_LAUNCH_CHECK();
}

int32_t d3d_render_embed(const float* rows_pos, const float* rows_dir, const float* rows_scale, const uint16_t* rows_fts, int64_t n_cap,
                         const int32_t* ray_slot, const int32_t* ray_env, const float* ray_xyz, const int32_t* topk, const int32_t* sidx,
                         const float* pose3, const float* rel_direction, int32_t n_rays, int32_t R, int32_t N, int32_t n_imp, int32_t k,
                         float far_, const float* w6, const float* b6, const float* ln_w, const float* ln_b, float eps, uint16_t* s16,
                         float* geom6, float* sample_xyz, void* stream) {
    if (n_rays <= 0) return D3D_OK;
    const int64_t items = (int64_t)n_rays * n_imp * k;
    hipLaunchKernelGGL(k_render_embed, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows_pos, rows_dir, rows_scale,
                       rows_fts, n_cap, ray_slot, ray_xyz, topk, sidx, pose3, ray_env, rel_direction, R, N, n_imp, k, far_, w6, b6, ln_w,
                       ln_b, eps, items, s16, geom6, sample_xyz);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_composite(const void* feat16, int64_t ldf, const void* dens16, int64_t ldd, const float* rel_dist, const int32_t* topk,
                      int32_t n_rays, int32_t N, int32_t n_imp, float* feature_map, float* depth, void* stream) {
    if (n_rays <= 0) return D3D_OK;
    if (n_imp > 16) {
        d3d_set_error_("d3d_composite: n_imp <= 16");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_composite, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)feat16, ldf, (const uint16_t*)dens16,
                       ldd, rel_dist, topk, n_rays, N, n_imp, feature_map, depth);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
