// dense_kernels.hip -- HBM-bound row kernels around the GEMMs: LayerNorm / RMSNorm (fp32 statistics,
// 16-byte vector loads, one wave per row, the row held in registers: one read + one write per element),
// in-place half-split RoPE on the fused QKV buffer, and the bicubic resize + normalise front-end.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int WAVE = 64;

template <bool BF16>
__device__ __forceinline__ float ld16(uint16_t v) {
    if constexpr (BF16) return __uint_as_float((uint32_t)v << 16);
    else return __half2float(*reinterpret_cast<const __half*>(&v));
}
template <bool BF16>
__device__ __forceinline__ uint16_t st16(float f) {
    if constexpr (BF16) {
        const __bf16 r = (__bf16)f;                  // fptrunc selects the hardware converter (v_cvt_pk_bf16_f32: RNE, NaN-safe)
        return *reinterpret_cast<const uint16_t*>(&r);
    } else {
        __half h = __float2half_rn(f);
        return *reinterpret_cast<uint16_t*>(&h);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row; D = 512*NCH elements (8 per lane per chunk).  RMS: y = x * rsqrt(mean(x^2)+eps) * w.
// LN: y = (x-mean) * rsqrt(var+eps) * w + b   (biased variance, float32, like F.layer_norm on x.float()).
template <bool BF16, bool RMS, int NCH>
__global__ void __launch_bounds__(256)
k_norm(const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, uint16_t* __restrict__ y,
       int rows, int D, int64_t ldx, int64_t ldy, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const uint16_t* xr = x + (int64_t)row * ldx;
    float v[NCH][8];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 512 + lane * 8;
        if (off < D) {
            const uint4 raw = *reinterpret_cast<const uint4*>(xr + off);
            const uint16_t* h = reinterpret_cast<const uint16_t*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = ld16<BF16>(h[j]);
                s += v[c][j];
                ss += v[c][j] * v[c][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    float mean = 0.f, rstd;
    if constexpr (RMS) {
        ss = wave_sum(ss);
        rstd = rsqrtf(ss / (float)D + eps);
    } else {
        s = wave_sum(s);
        mean = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c * 512 + lane * 8 < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[c][j] - mean;
                    q += d * d;
                }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)D + eps);
    }
    uint16_t* yr = y + (int64_t)row * ldy;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 512 + lane * 8;
        if (off < D) {
            const float4 w0 = *reinterpret_cast<const float4*>(w + off), w1 = *reinterpret_cast<const float4*>(w + off + 4);
            const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float bb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (!RMS) {
                const float4 b0 = *reinterpret_cast<const float4*>(b + off), b1 = *reinterpret_cast<const float4*>(b + off + 4);
                bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
            }
            uint16_t o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (RMS) o[j] = st16<BF16>(ld16<BF16>(st16<BF16>(v[c][j] * rstd)) * ww[j]);      // HF Phi3RMSNorm: weight * x_hat.to(dtype)
                else o[j] = st16<BF16>((v[c][j] - mean) * rstd * ww[j] + bb[j]);
            }
            *reinterpret_cast<uint4*>(yr + off) = *reinterpret_cast<const uint4*>(o);
        }
    }
}

// In-place half-split RoPE over the first `n_rot_heads` heads of every row of the fused QKV buffer
// (q heads then k heads are contiguous in Phi-3's qkv_proj output).  pos = row % S.
template <bool BF16>
__global__ void k_rope(uint16_t* __restrict__ qkv, const float* __restrict__ cos_t, const float* __restrict__ sin_t, int rows,
                       int S, int n_rot_heads, int hd, int64_t ld, const int32_t* __restrict__ pos_of_row) {
    // one thread = 8 consecutive rotation pairs of one head: two 16-byte loads / stores (head_dim/2 % 8 == 0)
    const int half = hd >> 1, cpb = half >> 3;                    // chunks of 8 pairs per head
    const int per_row = n_rot_heads * cpb;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * per_row) return;
    const int row = (int)(i / per_row), r = (int)(i % per_row);
    const int h = r / cpb, p = (r % cpb) * 8;
    const int pos = pos_of_row ? pos_of_row[row] : row % S;      // packed (varlen) batches carry an explicit position per row
    uint16_t* base = qkv + (int64_t)row * ld + h * hd + p;
    const uint4 a = *reinterpret_cast<const uint4*>(base), b = *reinterpret_cast<const uint4*>(base + half);
    const uint16_t* ah = reinterpret_cast<const uint16_t*>(&a);
    const uint16_t* bh = reinterpret_cast<const uint16_t*>(&b);
    const float4 c0 = *reinterpret_cast<const float4*>(cos_t + pos * half + p), c1 = *reinterpret_cast<const float4*>(cos_t + pos * half + p + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sin_t + pos * half + p), s1 = *reinterpret_cast<const float4*>(sin_t + pos * half + p + 4);
    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint16_t o1[8], o2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x1 = ld16<BF16>(ah[j]), x2 = ld16<BF16>(bh[j]);
        // HF apply_rotary_pos_emb on 16-bit tensors: (x * cos) + (rotate_half(x) * sin), every product and the sum stored 16-bit
        o1[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x1 * cc[j])) - ld16<BF16>(st16<BF16>(x2 * ss[j])));
        o2[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x2 * cc[j])) + ld16<BF16>(st16<BF16>(x1 * ss[j])));
    }
    *reinterpret_cast<uint4*>(base) = *reinterpret_cast<const uint4*>(o1);
    *reinterpret_cast<uint4*>(base + half) = *reinterpret_cast<const uint4*>(o2);
}

// Bicubic (A = -0.75, align_corners = False, border-clamped taps) resize of uint8 HWC images to SxS, result
// rounded back to uint8 (torchvision tensor path), /255, (x-mean)/std -> float CHW.
__device__ __forceinline__ void cubic_w(float t, float* w) {
    const float A = -0.75f;
    float x = t + 1.0f;
    w[0] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
    x = t;
    w[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 1.0f - t;
    w[2] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    x = 2.0f - t;
    w[3] = ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A;
}

__global__ void k_resize_normalize(const uint8_t* __restrict__ rgb, float* __restrict__ out, int B, int H, int W, int S,
                                   float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * S * S) return;
    const int b = (int)(i / (S * S)), oy = (int)((i / S) % S), ox = (int)(i % S);
    const uint8_t* img = rgb + (int64_t)b * H * W * 3;
    float val[3];
    if (H == S && W == S) {
#pragma unroll
        for (int c = 0; c < 3; ++c) val[c] = (float)img[((int64_t)oy * W + ox) * 3 + c];
    } else {
        const float sy = (float)H / (float)S, sx = (float)W / (float)S;
        const float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        float wy[4], wx[4];
        cubic_w(fy - (float)iy, wy);
        cubic_w(fx - (float)ix, wx);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int yy = min(max(iy - 1 + a, 0), H - 1);
                float rowv = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int xx = min(max(ix - 1 + d, 0), W - 1);
                    rowv += (float)img[((int64_t)yy * W + xx) * 3 + c] * wx[d];
                }
                acc += rowv * wy[a];
            }
            val[c] = fminf(fmaxf(rintf(acc), 0.f), 255.f);
        }
    }
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((int64_t)b * 3 + c) * S + oy) * S + ox] = (val[c] / 255.0f - mean[c]) / stdv[c];
}

// SwiGLU over a plain [gate | up] projection output: out[m, i] = up * silu(gate); 8 elements per thread.
template <bool BF16>
__global__ void k_swiglu(const uint16_t* __restrict__ gu, uint16_t* __restrict__ out, int64_t rows, int I) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= rows * I) return;
    const int64_t r = i / I;
    const int c = (int)(i % I);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + r * 2 * I + c);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + r * 2 * I + I + c);
    const uint16_t* gh = reinterpret_cast<const uint16_t*>(&g);
    const uint16_t* uh = reinterpret_cast<const uint16_t*>(&u);
    uint16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gv = ld16<BF16>(gh[j]);
        o[j] = st16<BF16>(ld16<BF16>(uh[j]) * ld16<BF16>(st16<BF16>(gv / (1.0f + __expf(-gv)))));      // silu's result is a 16-bit tensor (HF Phi3MLP)
    }
    *reinterpret_cast<uint4*>(out + i) = *reinterpret_cast<const uint4*>(o);
}

// Variable-length self-attention over packed token sets (the patch->instance / instance->zone set encoders,
// VLN-FF:134-155): float32, head_dim 64, no mask inside a set, softmax scale 1/8.  qkv (T, 3*H*64) packed
// [q | k | v]; set g owns tokens [off[g], off[g+1]).  One workgroup per (set, head, 64-query tile): every thread
// owns one query row in registers (q, running max/sum, 64-wide output accumulator) and streams the set's keys and
// values through LDS in tiles of 64 (online softmax).  `q_rows` limits the queries to the first rows of each set
// (1 = CLS only, for the last layer whose other rows are never read).
__global__ void __launch_bounds__(64)
k_set_attention(const float* __restrict__ qkv, const int32_t* __restrict__ off, int H, int q_rows, float* __restrict__ out) {
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
    float q[HD], o[HD];
    float m = -INFINITY, l = 0.f;
    if (active) {
        const float* qp = qkv + (int64_t)(t0 + qi) * ld + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] = qp[d] * 0.125f;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    for (int k0 = 0; k0 < L; k0 += TK) {
        const int cnt = min(TK, L - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * HD; i += 64) {
            const int r = i / HD, d = i % HD;
            const float* base = qkv + (int64_t)(t0 + k0 + r) * ld + h * HD + d;
            ks[r][d] = base[H * HD];
            vs[r][d] = base[2 * H * HD];
        }
        __syncthreads();
        if (active) {
            // four keys per step: four independent 64-deep dot-product chains (the single chain was latency bound), one
            // rescale of the accumulator per step instead of per key
            int r = 0;
            for (; r + 4 <= cnt; r += 4) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    s0 += q[d] * ks[r][d];
                    s1 += q[d] * ks[r + 1][d];
                    s2 += q[d] * ks[r + 2][d];
                    s3 += q[d] * ks[r + 3][d];
                }
                const float mn = fmaxf(fmaxf(m, fmaxf(s0, s1)), fmaxf(s2, s3));
                const float a = __expf(m - mn);
                const float p0 = __expf(s0 - mn), p1 = __expf(s1 - mn), p2 = __expf(s2 - mn), p3 = __expf(s3 - mn);
                l = l * a + ((p0 + p1) + (p2 + p3));
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] = o[d] * a + ((p0 * vs[r][d] + p1 * vs[r + 1][d]) + (p2 * vs[r + 2][d] + p3 * vs[r + 3][d]));
                m = mn;
            }
            for (; r < cnt; ++r) {
                float sdot = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) sdot += q[d] * ks[r][d];
                const float mn = fmaxf(m, sdot);
                const float a = __expf(m - mn), p = __expf(sdot - mn);
                l = l * a + p;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] = o[d] * a + p * vs[r][d];
                m = mn;
            }
        }
    }
    if (active) {
        const float inv = 1.0f / l;
        float* op = out + (int64_t)(t0 + qi) * (H * HD) + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = o[d] * inv;
    }
}

// CLS-only variant (q_rows == 1: the last encoder layer, whose other rows are never read).  One wave per (set, head): the
// keys are dealt out to the 64 lanes (lane j scores keys j, j+64, ...), softmax statistics are wave reductions, and lane d
// sums column d of the probability-weighted values -- instead of one active thread walking all L keys.
__global__ void __launch_bounds__(64)
k_set_attention_cls(const float* __restrict__ qkv, const int32_t* __restrict__ off, int H, float* __restrict__ out) {
    constexpr int HD = 64;
    __shared__ float qs[HD];
    __shared__ float ps[64];
    const int g = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int t0 = off[g], L = off[g + 1] - t0;
    if (L <= 0) return;
    const int64_t ld = (int64_t)3 * H * HD;
    qs[lane] = qkv[(int64_t)t0 * ld + h * HD + lane] * 0.125f;
    __syncthreads();
    float m = -INFINITY, l = 0.f, o = 0.f;                  // lane d accumulates output column d
    for (int k0 = 0; k0 < L; k0 += 64) {
        const int r = k0 + lane;
        float sdot = -INFINITY;
        if (r < L) {
            const float* kp = qkv + (int64_t)(t0 + r) * ld + H * HD + h * HD;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kp + d);
                s0 += qs[d] * kv.x;
                s1 += qs[d + 1] * kv.y;
                s2 += qs[d + 2] * kv.z;
                s3 += qs[d + 3] * kv.w;
            }
            sdot = (s0 + s1) + (s2 + s3);
        }
        float tmax = sdot;
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, w, 64));
        const float mn = fmaxf(m, tmax);
        const float a = __expf(m - mn);
        const float p = r < L ? __expf(sdot - mn) : 0.f;
        float psum = p;
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) psum += __shfl_xor(psum, w, 64);
        l = l * a + psum;
        __syncthreads();
        ps[lane] = p;
        __syncthreads();
        const int cnt = min(64, L - k0);
        const float* vp = qkv + (int64_t)(t0 + k0) * ld + 2 * H * HD + h * HD + lane;
        float acc0 = 0.f, acc1 = 0.f;
        int j = 0;
        for (; j + 2 <= cnt; j += 2) {
            acc0 += ps[j] * vp[(int64_t)j * ld];
            acc1 += ps[j + 1] * vp[(int64_t)(j + 1) * ld];
        }
        if (j < cnt) acc0 += ps[j] * vp[(int64_t)j * ld];
        o = o * a + (acc0 + acc1);
        m = mn;
    }
    out[(int64_t)t0 * (H * HD) + h * HD + lane] = o / l;
}

template <bool BF16, bool RMS>
int32_t launch_norm(const void* x, const float* w, const float* b, void* y, int rows, int D, int64_t ldx, int64_t ldy, float eps,
                    hipStream_t s) {
    const int nch = (D + 511) / 512;
    dim3 grid((rows + 3) / 4), block(256);
#define D3D_NORM_CASE(N)                                                                                                       \
    case N:                                                                                                                    \
        hipLaunchKernelGGL((k_norm<BF16, RMS, N>), grid, block, 0, s, (const uint16_t*)x, w, b, (uint16_t*)y, rows, D, ldx, ldy, eps); \
        break;
    switch (nch) {
        D3D_NORM_CASE(1)
        D3D_NORM_CASE(2)
        D3D_NORM_CASE(3)
        D3D_NORM_CASE(4)
        D3D_NORM_CASE(6)
        D3D_NORM_CASE(8)
        default:
            d3d_set_error_("d3d_norm: unsupported width (need D <= 4096 with ceil(D/512) in {1,2,3,4,6,8})");
            return D3D_EINVAL;
    }
#undef D3D_NORM_CASE
    D3D_LAUNCH_CHECK();
}

}  // namespace

extern "C" {

// y = LayerNorm(x) (rms == 0) or RMSNorm(x) (rms == 1); x,y bf16 (dtype 0) / fp16 (dtype 1) rows of D (D % 8 == 0), fp32 w[,b].
int32_t d3d_norm(const void* x, const float* w, const float* b, void* y, int32_t rows, int32_t D, int64_t ldx, int64_t ldy,
                 float eps, int32_t rms, int32_t dtype, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 8 != 0 || (ldx & 7) || (ldy & 7)) {
        d3d_set_error_("d3d_norm: D, ldx, ldy must be multiples of 8");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == 0) return rms ? launch_norm<true, true>(x, w, b, y, rows, D, ldx, ldy, eps, s) : launch_norm<true, false>(x, w, b, y, rows, D, ldx, ldy, eps, s);
    return rms ? launch_norm<false, true>(x, w, b, y, rows, D, ldx, ldy, eps, s) : launch_norm<false, false>(x, w, b, y, rows, D, ldx, ldy, eps, s);
}

int32_t d3d_rope_inplace(void* qkv, const float* cos_t, const float* sin_t, int32_t rows, int32_t S, int32_t n_rot_heads,
                         int32_t head_dim, int64_t ld, const int32_t* pos_of_row, int32_t dtype, void* stream) {
    if (rows <= 0) return D3D_OK;
    if ((head_dim / 2) % 8 != 0 || (ld & 7)) {
        d3d_set_error_("d3d_rope_inplace: head_dim/2 and ld must be multiples of 8");
        return D3D_EINVAL;
    }
    const int64_t n = (int64_t)rows * n_rot_heads * (head_dim / 16);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(k_rope<true>, grid, block, 0, (hipStream_t)stream, (uint16_t*)qkv, cos_t, sin_t, rows, S, n_rot_heads, head_dim, ld, pos_of_row);
    else
        hipLaunchKernelGGL(k_rope<false>, grid, block, 0, (hipStream_t)stream, (uint16_t*)qkv, cos_t, sin_t, rows, S, n_rot_heads, head_dim, ld, pos_of_row);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_swiglu(const void* gate_up, void* out, int64_t rows, int32_t I, int32_t dtype, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (I % 8) {
        d3d_set_error_("d3d_swiglu: I must be a multiple of 8");
        return D3D_EINVAL;
    }
    const int64_t n = rows * I / 8;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(k_swiglu<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)gate_up, (uint16_t*)out, rows, I);
    else
        hipLaunchKernelGGL(k_swiglu<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)gate_up, (uint16_t*)out, rows, I);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_set_attention(const float* qkv, const int32_t* set_off, int32_t n_sets, int32_t n_heads, int32_t max_len, int32_t q_rows,
                          float* out, void* stream) {
    if (n_sets <= 0 || max_len <= 0) return D3D_OK;
    const int nq = q_rows > 0 ? (q_rows < max_len ? q_rows : max_len) : max_len;
    if (q_rows == 1) {
        hipLaunchKernelGGL(k_set_attention_cls, dim3(n_sets, n_heads), dim3(64), 0, (hipStream_t)stream, qkv, set_off, n_heads, out);
    } else {
        dim3 grid(n_sets, n_heads, (nq + 63) / 64);
        hipLaunchKernelGGL(k_set_attention, grid, dim3(64), 0, (hipStream_t)stream, qkv, set_off, n_heads, q_rows, out);
    }
    D3D_LAUNCH_CHECK();
}

int32_t d3d_resize_normalize(const uint8_t* rgb, float* out, int32_t B, int32_t H, int32_t W, int32_t S, const float* mean3_h,
                             const float* std3_h, void* stream) {
    if (B <= 0) return D3D_OK;
    const int64_t n = (int64_t)B * S * S;
    hipLaunchKernelGGL(k_resize_normalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, out, B, H, W, S,
                       mean3_h[0], mean3_h[1], mean3_h[2], std3_h[0], std3_h[1], std3_h[2]);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
