// train_kernels.hip -- the small kernels around the GEMMs of the tcnn-MLP backward pass (first slice):
//
//   d3d_transpose_pad16   (R, C) 16-bit row-major -> (C, Rp) with Rp >= R a multiple of 64, columns [R, Rp) zero.  The weight
//                         gradient dW[N,K] = sum_m dz[m,n] h[m,k] is the NT GEMM  dz^T[N,Mp] . (h^T[K,Mp])^T  (both operands
//                         contiguous along the reduction dimension m), so dz and h are transposed once, through LDS.
//   d3d_lrelu_bwd         dz = dy * (y > 0 ? 1 : 0.01): the LeakyReLU(0.01) of a layer whose OUTPUT y is kept (sign(y) = sign(z)).
//                         (Between layers the same factor is the GEMM epilogue 8 of the data-gradient GEMM.)
//
// Round 4 (backward of a7 / a11 / a22 on the device instead of PyTorch autograd expressions):
//   d3d_layer_norm_bwd_f32   dx of y = [gelu](LayerNorm(x) * w + b) and per-workgroup partial sums of dw / db (summed by the caller)
//   d3d_set_attention_bwd    dq, dk, dv of the packed variable-length set attention (d3d_set_attention), two launches:
//                            per query row (log-sum-exp, D = dO . O, dq), then per key row (dk, dv); fixed summation order
//   d3d_composite_bwd        d(sample features), d(sample densities) of the alpha compositing + L2 normalisation (d3d_composite)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

// 64 x 64 tile per 256-thread workgroup through a padded LDS tile (row pitch 66 elements: conflict-free column reads).
__global__ void __launch_bounds__(256)
k_transpose_pad16(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C, int64_t ld_in, int Rp) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < C) ? in[(int64_t)r * ld_in + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(int64_t)c * Rp + r] = tile[tx][ty * 16 + i];
    }
}

template <bool BF16>
__global__ void k_lrelu_bwd(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y, uint16_t* __restrict__ dz, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint4 a = reinterpret_cast<const uint4*>(dy)[i], b = reinterpret_cast<const uint4*>(y)[i];
    const uint16_t* ah = reinterpret_cast<const uint16_t*>(&a);
    const uint16_t* bh = reinterpret_cast<const uint16_t*>(&b);
    uint16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float g, v;
        if constexpr (BF16) {
            g = __uint_as_float((uint32_t)ah[j] << 16);
            v = __uint_as_float((uint32_t)bh[j] << 16);
        } else {
            g = __half2float(*reinterpret_cast<const __half*>(&ah[j]));
            v = __half2float(*reinterpret_cast<const __half*>(&bh[j]));
        }
        const float r = v > 0.f ? g : 0.01f * g;
        if constexpr (BF16) {
            uint32_t u = __float_as_uint(r);
            u += 0x7fffu + ((u >> 16) & 1u);
            o[j] = (uint16_t)(u >> 16);
        } else {
            const __half h = __float2half_rn(r);
            o[j] = *reinterpret_cast<const uint16_t*>(&h);
        }
    }
    reinterpret_cast<uint4*>(dz)[i] = *reinterpret_cast<const uint4*>(o);
}

// ---- LayerNorm backward ---------------------------------------------------------------------------------------------------------
// One wave per row at a time, LN_ROWS rows per workgroup (4 waves x LN_ROWS / 4 rows each).  Per row: recompute mean / rstd (the forward's
// two-pass arithmetic), z = xhat * w + b, dz = dy * gelu'(z) (or dy), dxhat = dz * w, dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)).
// dw += dz * xhat, db += dz: every wave keeps its column sums in registers, the four waves are added in wave order through LDS and the
// workgroup writes ONE partial row (deterministic; the caller sums the partial rows).
constexpr int LN_ROWS = 64;

__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
    return cdf + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}

template <int NCH>
__global__ void __launch_bounds__(256)
k_layer_norm_bwd_f32(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ dy,
                     float* __restrict__ dx, float* __restrict__ dw_part, float* __restrict__ db_part, int rows, int D, int64_t ldx, int64_t lddy,
                     int64_t lddx, float eps, int gelu) {
    __shared__ float red[2][NCH * 256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float gw[NCH][4], gb[NCH][4], ww[NCH][4], bb[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gw[c][j] = gb[c][j] = 0.f;
            ww[c][j] = off + j < D ? w[off + j] : 0.f;
            bb[c][j] = off + j < D ? b[off + j] : 0.f;
        }
    }
    for (int r = wave; r < LN_ROWS; r += 4) {
        const int row = blockIdx.x * LN_ROWS + r;
        if (row >= rows) break;
        float v[NCH][4], g[NCH][4];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = c * 256 + lane * 4;
            if (off < D) {
                const float4 a = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + off);
                const float4 d = *reinterpret_cast<const float4*>(dy + (int64_t)row * lddy + off);
                v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
                g[c][0] = d.x; g[c][1] = d.y; g[c][2] = d.z; g[c][3] = d.w;
                s += (a.x + a.y) + (a.z + a.w);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[c][j] = g[c][j] = 0.f;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c * 256 + lane * 4 < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[c][j] - mean;
                    q += d * d;
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = rsqrtf(q / (float)D + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c * 256 + lane * 4 < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (v[c][j] - mean) * rstd;
                    float dz = g[c][j];
                    if (gelu) dz *= gelu_grad(xh * ww[c][j] + bb[c][j]);
                    gw[c][j] += dz * xh;
                    gb[c][j] += dz;
                    const float dxh = dz * ww[c][j];
                    v[c][j] = xh;
                    g[c][j] = dxh;
                    c1 += dxh;
                    c2 += dxh * xh;
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
        }
        c1 /= (float)D;
        c2 /= (float)D;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = c * 256 + lane * 4;
            if (off < D)
                *reinterpret_cast<float4*>(dx + (int64_t)row * lddx + off) =
                    float4{rstd * (g[c][0] - c1 - v[c][0] * c2), rstd * (g[c][1] - c1 - v[c][1] * c2), rstd * (g[c][2] - c1 - v[c][2] * c2),
                           rstd * (g[c][3] - c1 - v[c][3] * c2)};
        }
    }
    for (int wv = 0; wv < 4; ++wv) {                       // ((wave 0 + wave 1) + wave 2) + wave 3: a fixed order
        if (wave == wv) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = c * 256 + lane * 4 + j;
                    red[0][i] = wv ? red[0][i] + gw[c][j] : gw[c][j];
                    red[1][i] = wv ? red[1][i] + gb[c][j] : gb[c][j];
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < D; i += 256) {
        dw_part[(int64_t)blockIdx.x * D + i] = red[0][i];
        db_part[(int64_t)blockIdx.x * D + i] = red[1][i];
    }
}

// ---- GELU (erf) forward / backward, float32 (the feed-forward activation of nn.TransformerEncoderLayer, PRE-FF:134-141) ----------------------
__global__ void k_gelu_f32(const float* __restrict__ z, const float* __restrict__ dy, float* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 a = reinterpret_cast<const float4*>(z)[i];
    float4 r;
    if (dy) {                                               // backward: dz = dy * gelu'(z)
        const float4 g = reinterpret_cast<const float4*>(dy)[i];
        r = float4{g.x * gelu_grad(a.x), g.y * gelu_grad(a.y), g.z * gelu_grad(a.z), g.w * gelu_grad(a.w)};
    } else {
        r = float4{0.5f * a.x * (1.0f + erff(a.x * 0.70710678118654752f)), 0.5f * a.y * (1.0f + erff(a.y * 0.70710678118654752f)),
                   0.5f * a.z * (1.0f + erff(a.z * 0.70710678118654752f)), 0.5f * a.w * (1.0f + erff(a.w * 0.70710678118654752f))};
    }
    reinterpret_cast<float4*>(out)[i] = r;
}

// ---- set attention backward -----------------------------------------------------------------------------------------------------
// Same decomposition as k_set_attention (a thread owns one row, the other side streams through LDS in tiles of 64), float32, head_dim 64,
// scale 1/8.  Pass Q: thread = query row i: lse_i = log sum_j exp(s_ij), D_i = dO_i . O_i, dq_i = sum_j dS_ij k_j / 8 with
// dS_ij = p_ij (dO_i . v_j - D_i).  Pass KV: thread = key row j: dv_j = sum_i p_ij dO_i, dk_j = sum_i dS_ij q_i / 8 (p recomputed from lse).
// Sums run in index order: deterministic.  `q_rows` > 0: only the first rows of each set were queried (the CLS-only last layer).
__global__ void __launch_bounds__(64)
k_set_attn_bwd_q(const float* __restrict__ qkv, const float* __restrict__ o, const float* __restrict__ dout, const int32_t* __restrict__ off, int H,
                 int q_rows, float* __restrict__ dqkv, float* __restrict__ lse, float* __restrict__ dsum) {
    constexpr int HD = 64, TK = 64;
    __shared__ float ks[TK][HD + 1];
    __shared__ float vs[TK][HD + 1];
    const int g = blockIdx.x, h = blockIdx.y, qt = blockIdx.z;
    const int t0 = off[g], L = off[g + 1] - t0;
    const int nq = q_rows > 0 ? min(q_rows, L) : L;
    if (qt * 64 >= nq) return;
    const int qi = qt * 64 + threadIdx.x;
    const bool active = qi < nq;
    const int64_t ld = (int64_t)3 * H * HD;
    float q[HD], go[HD], dq[HD];
    float D = 0.f, m = -INFINITY, l = 0.f;
    if (active) {
        const float* qp = qkv + (int64_t)(t0 + qi) * ld + h * HD;
        const float* op = o + (int64_t)(t0 + qi) * (H * HD) + h * HD;
        const float* gp = dout + (int64_t)(t0 + qi) * (H * HD) + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[d] = qp[d] * 0.125f;
            go[d] = gp[d];
            D += go[d] * op[d];
            dq[d] = 0.f;
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int k0 = 0; k0 < L; k0 += TK) {
            const int cnt = min(TK, L - k0);
            __syncthreads();
            for (int i = threadIdx.x; i < cnt * HD; i += 64) {
                const int r = i / HD, d = i % HD;
                const float* base = qkv + (int64_t)(t0 + k0 + r) * ld + h * HD + d;
                ks[r][d] = base[H * HD];
                if (pass) vs[r][d] = base[2 * H * HD];
            }
            __syncthreads();
            if (!active) continue;
            for (int r = 0; r < cnt; ++r) {
                float sdot = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) sdot += q[d] * ks[r][d];
                if (!pass) {
                    const float mn = fmaxf(m, sdot);
                    l = l * __expf(m - mn) + __expf(sdot - mn);
                    m = mn;
                } else {
                    float dp = 0.f;
#pragma unroll
                    for (int d = 0; d < HD; ++d) dp += go[d] * vs[r][d];
                    const float ds = __expf(sdot - m) * (dp - D);          // (m holds lse in the second pass)
#pragma unroll
                    for (int d = 0; d < HD; ++d) dq[d] += ds * ks[r][d];
                }
            }
        }
        if (!pass && active) m = m + __logf(l);
    }
    if (active) {
        float* dp = dqkv + (int64_t)(t0 + qi) * ld + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) dp[d] = dq[d] * 0.125f;
        lse[(int64_t)(t0 + qi) * H + h] = m;
        dsum[(int64_t)(t0 + qi) * H + h] = D;
    }
}

__global__ void __launch_bounds__(64)
k_set_attn_bwd_kv(const float* __restrict__ qkv, const float* __restrict__ dout, const int32_t* __restrict__ off, int H, int q_rows,
                  const float* __restrict__ lse, const float* __restrict__ dsum, float* __restrict__ dqkv) {
    constexpr int HD = 64, TQ = 64;
    __shared__ float qs[TQ][HD + 1];
    __shared__ float gs[TQ][HD + 1];
    __shared__ float ls[TQ], Ds[TQ];
    const int g = blockIdx.x, h = blockIdx.y, kt = blockIdx.z;
    const int t0 = off[g], L = off[g + 1] - t0;
    if (kt * 64 >= L) return;
    const int nq = q_rows > 0 ? min(q_rows, L) : L;
    const int kj = kt * 64 + threadIdx.x;
    const bool active = kj < L;
    const int64_t ld = (int64_t)3 * H * HD;
    float k[HD], v[HD], dk[HD], dv[HD];
    if (active) {
        const float* kp = qkv + (int64_t)(t0 + kj) * ld + (H + h) * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            k[d] = kp[d];
            v[d] = kp[H * HD + d];
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) dk[d] = dv[d] = 0.f;
    for (int q0 = 0; q0 < nq; q0 += TQ) {
        const int cnt = min(TQ, nq - q0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * HD; i += 64) {
            const int r = i / HD, d = i % HD;
            qs[r][d] = qkv[(int64_t)(t0 + q0 + r) * ld + h * HD + d] * 0.125f;
            gs[r][d] = dout[(int64_t)(t0 + q0 + r) * (H * HD) + h * HD + d];
        }
        if (threadIdx.x < cnt) {
            ls[threadIdx.x] = lse[(int64_t)(t0 + q0 + threadIdx.x) * H + h];
            Ds[threadIdx.x] = dsum[(int64_t)(t0 + q0 + threadIdx.x) * H + h];
        }
        __syncthreads();
        if (!active) continue;
        for (int r = 0; r < cnt; ++r) {
            float sdot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                sdot += qs[r][d] * k[d];
                dp += gs[r][d] * v[d];
            }
            const float p = __expf(sdot - ls[r]);
            const float ds = p * (dp - Ds[r]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dv[d] += p * gs[r][d];
                dk[d] += ds * qs[r][d];
            }
        }
    }
    if (active) {
        float* dp = dqkv + (int64_t)(t0 + kj) * ld + (H + h) * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dp[d] = dk[d];
            dp[H * HD + d] = dv[d];
        }
    }
}

// ---- alpha compositing backward (d3d_composite / raw2feature, PRE-FF:446-474) ---------------------------------------------------------
// One wave per ray, the forward's arithmetic recomputed: w_t = alpha_t * prod_{u before t} (1 - alpha_u + 1e-10), acc = sum_t w_t f_t,
// out = acc / max(|acc|, 1e-7).  g = dL/dout:  dacc = (g - out (out . g)) / |acc|,  df_t = w_t dacc,  dw_t = f_t . dacc,
// dalpha_t = dw_t T_t - sum_{v behind t} dw_v w_v / (1 - alpha_t + 1e-10),  dsigma_t = dalpha_t dist_t exp(-softplus(sigma_t) dist_t) sigmoid(sigma_t).
__global__ void __launch_bounds__(256)
k_composite_bwd(const uint16_t* __restrict__ feat, int64_t ldf, const uint16_t* __restrict__ dens, int64_t ldd, const float* __restrict__ rel_dist,
                const int32_t* __restrict__ topk, const float* __restrict__ gout, int n_rays, int N, int n_imp, float* __restrict__ dfeat,
                float* __restrict__ ddens) {
    constexpr int FT = 768, WV = 64;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (ray >= n_rays) return;
    float w[16], al[16], Tt[16], sp[16], dist[16], sig[16];
    int nn[16];
    for (int t = 0; t < n_imp; ++t) {
        nn[t] = topk[(int64_t)ray * n_imp + t];
        sig[t] = __half2float(*reinterpret_cast<const __half*>(dens + ((int64_t)ray * n_imp + t) * ldd));
        sp[t] = sig[t] > 20.f ? sig[t] : log1pf(expf(sig[t]));
        dist[t] = nn[t] + 1 < N ? fabsf(rel_dist[nn[t] + 1] - rel_dist[nn[t]]) : 1e10f;
        al[t] = 1.0f - expf(-fmaxf(sp[t], 0.f) * dist[t]);
    }
    for (int t = 0; t < n_imp; ++t) {
        float T = 1.0f;
        for (int u = 0; u < n_imp; ++u)
            if (nn[u] < nn[t]) T *= (1.0f - al[u]) + 1e-10f;
        Tt[t] = T;
        w[t] = al[t] * T;
    }
    float acc[12], gg[12];
    float nrm = 0.f, og = 0.f;
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    for (int t = 0; t < n_imp; ++t) {
        const uint16_t* f = feat + ((int64_t)ray * n_imp + t) * ldf;
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] += w[t] * __half2float(*reinterpret_cast<const __half*>(f + q * WV + lane));
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        gg[q] = gout[(int64_t)ray * FT + q * WV + lane];
        nrm += acc[q] * acc[q];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrm += __shfl_xor(nrm, o);
    const float len = sqrtf(nrm);
    const float inv = 1.0f / fmaxf(len, 1e-7f);
#pragma unroll
    for (int q = 0; q < 12; ++q) og += acc[q] * inv * gg[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) og += __shfl_xor(og, o);
    float dacc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) dacc[q] = len > 1e-7f ? (gg[q] - acc[q] * inv * og) * inv : gg[q] * inv;     // (the clamp branch: out = acc / 1e-7)
    float dw[16];
    for (int t = 0; t < n_imp; ++t) {
        const uint16_t* f = feat + ((int64_t)ray * n_imp + t) * ldf;
        float* df = dfeat + ((int64_t)ray * n_imp + t) * FT;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            s += __half2float(*reinterpret_cast<const __half*>(f + q * WV + lane)) * dacc[q];
            df[q * WV + lane] = w[t] * dacc[q];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        dw[t] = s;
    }
    if (lane == 0) {
        for (int t = 0; t < n_imp; ++t) {
            float da = dw[t] * Tt[t];
            for (int v = 0; v < n_imp; ++v)
                if (nn[v] > nn[t]) da -= dw[v] * w[v] / ((1.0f - al[t]) + 1e-10f);
            const float dsp = sp[t] > 0.f ? da * dist[t] * expf(-sp[t] * dist[t]) : 0.f;
            const float dsg = sig[t] > 20.f ? dsp : dsp / (1.0f + expf(-sig[t]));
            ddens[(int64_t)ray * n_imp + t] = dsg;
        }
    }
}

}  // namespace

extern "C" {

int32_t d3d_layer_norm_bwd_f32(const float* x, const float* w, const float* b, const float* dy, float* dx, float* dw_part, float* db_part,
                               int32_t rows, int32_t D, int64_t ldx, int64_t lddy, int64_t lddx, float eps, int32_t gelu, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 4 != 0 || D > 3072 || (ldx & 3) || (lddy & 3) || (lddx & 3)) {
        d3d_set_error_("d3d_layer_norm_bwd_f32: D % 4 == 0, D <= 3072, row strides % 4 == 0");
        return D3D_EINVAL;
    }
    const int nch = (D + 255) / 256;
    dim3 grid((rows + LN_ROWS - 1) / LN_ROWS), block(256);
    hipStream_t s = (hipStream_t)stream;
#define D3D_LNB_CASE(N)                                                                                                                   \
    if (nch <= N) {                                                                                                                       \
        hipLaunchKernelGGL((k_layer_norm_bwd_f32<N>), grid, block, 0, s, x, w, b, dy, dx, dw_part, db_part, rows, D, ldx, lddy, lddx, eps, gelu); \
        D3D_LAUNCH_CHECK();                                                                                                               \
    }
    D3D_LNB_CASE(1)
    D3D_LNB_CASE(3)
    D3D_LNB_CASE(12)
#undef D3D_LNB_CASE
    d3d_set_error_("d3d_layer_norm_bwd_f32: unsupported width");
    return D3D_EINVAL;
}

int32_t d3d_layer_norm_bwd_rows_per_block(void) { return LN_ROWS; }

int32_t d3d_gelu_f32(const float* z, const float* dy, float* out, int64_t n, void* stream) {
    if (n <= 0) return D3D_OK;
    if (n % 4 || ((uintptr_t)z & 15) || ((uintptr_t)out & 15) || (dy && ((uintptr_t)dy & 15))) {
        d3d_set_error_("d3d_gelu_f32: element count % 4 == 0 and 16-byte aligned buffers");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_gelu_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, dy, out, n / 4);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_set_attention_bwd(const float* qkv, const float* out, const float* dout, const int32_t* set_off, int32_t n_sets, int32_t n_heads,
                              int32_t max_len, int32_t q_rows, float* dqkv, float* lse_scratch, float* d_scratch, void* stream) {
    if (n_sets <= 0 || max_len <= 0) return D3D_OK;
    const int nq = q_rows > 0 ? (q_rows < max_len ? q_rows : max_len) : max_len;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_set_attn_bwd_q, dim3(n_sets, n_heads, (nq + 63) / 64), dim3(64), 0, s, qkv, out, dout, set_off, n_heads, q_rows, dqkv, lse_scratch,
                       d_scratch);
    hipLaunchKernelGGL(k_set_attn_bwd_kv, dim3(n_sets, n_heads, (max_len + 63) / 64), dim3(64), 0, s, qkv, dout, set_off, n_heads, q_rows, lse_scratch,
                       d_scratch, dqkv);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_composite_bwd(const void* feat16, int64_t ldf, const void* dens16, int64_t ldd, const float* rel_dist, const int32_t* topk, const float* gout,
                          int32_t n_rays, int32_t N, int32_t n_imp, float* dfeat, float* ddens, void* stream) {
    if (n_rays <= 0) return D3D_OK;
    if (n_imp > 16) {
        d3d_set_error_("d3d_composite_bwd: n_imp <= 16");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_composite_bwd, dim3((n_rays + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)feat16, ldf, (const uint16_t*)dens16, ldd,
                       rel_dist, topk, gout, n_rays, N, n_imp, dfeat, ddens);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_transpose_pad16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int32_t Rp, void* stream) {
    if (R <= 0 || C <= 0) return D3D_OK;
    if (Rp < R || Rp % 64 != 0) {
        d3d_set_error_("d3d_transpose_pad16: Rp must be a multiple of 64 and >= R");
        return D3D_EINVAL;
    }
    dim3 grid((C + 63) / 64, Rp / 64), block(256);
    hipLaunchKernelGGL(k_transpose_pad16, grid, block, 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out, R, C, ld_in, Rp);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_lrelu_bwd(const void* dy, const void* y, void* dz, int64_t n, int32_t dtype, void* stream) {
    if (n <= 0) return D3D_OK;
    if (n % 8) {
        d3d_set_error_("d3d_lrelu_bwd: element count must be a multiple of 8");
        return D3D_EINVAL;
    }
    const int64_t n8 = n / 8;
    dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(k_lrelu_bwd<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dz, n8);
    else
        hipLaunchKernelGGL(k_lrelu_bwd<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dz, n8);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
