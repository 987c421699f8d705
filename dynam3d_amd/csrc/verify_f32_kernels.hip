// verify_f32_kernels.hip -- the FLOAT32 VERIFICATION MODE of the dense towers (a3, a15, a17; round 6).
//
// The product evaluates CLIP in fp16 and llava / Phi-3 in bf16 like the reference does (VLN-POL:125, resnet_encoders.py:260): two 16-bit
// evaluations of a 32-layer network sit a noise band (~1.7e-2 on the logits) apart, so "within 1e-3 of the float32 oracle" cannot be
// asserted on them end to end.  `PolicyConfig(clip_dtype=float32, llava_dtype=float32)` runs the SAME host wiring (towers.py: module order,
// packing, rotary positions, causal masks, sliding window, prompt assembly) on float32 kernels, with strict dispatch, so that the wiring
// of every dense primitive IS asserted at 1e-3 against the float32 oracle (tests/test_gpu_f32_mode.py, bench.py `parity.f32_mode`).
// Speed is irrelevant here (one step ~1 s); clarity and float32 accuracy are the point:
//
//   d3d_attention_f32        softmax(q k^T / sqrt(hd)) v over dense or packed variable-length batches, causal / sliding window; head_dim 64
//                            or 96; same buffer contract as d3d_flash_attention_v3 with float32 elements (VLN-POL:463 SDPA; clip/model.py:178)
//   d3d_rms_norm_f32         HF Phi3RMSNorm in float32
//   d3d_rope_inplace_f32     HF apply_rotary_pos_emb (half split) in float32, in place on the fused QKV buffer
//   d3d_swiglu_f32           up * silu(gate) of a [gate | up] projection output
//   d3d_patchify_f32, d3d_vit_embed_ln_f32, d3d_assemble_prompt_f32     float32 twins of tower_kernels.hip
// The GEMMs of this mode are d3d_gemm_nt_f32 (exact float32 multiply-add chains on v_mfma_f32_16x16x4_f32; f32_kernels.hip), the
// LayerNorms d3d_layer_norm_f32.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup (one wave) per (sequence, head, 64-query tile): a thread owns one query row (q, output accumulator, running max / sum in
// registers) and the tile's keys / values stream through LDS 64 at a time (online softmax, four keys per step).
template <int HD>
__global__ void __launch_bounds__(64)
k_attention_f32(const float* __restrict__ qkv, float* __restrict__ out, int S, int H, int64_t row_stride, int64_t batch_stride, int q_off, int k_off,
                int v_off, float scale, int seq_len, const int32_t* __restrict__ cu, int causal, int window) {
    constexpr int TK = 64;
    __shared__ float ks[TK][HD + 1];
    __shared__ float vs[TK][HD + 1];
    const int b = blockIdx.x, h = blockIdx.y, qt = blockIdx.z;
    int64_t row0 = (int64_t)b * S;
    const float* base = qkv + (int64_t)b * batch_stride;
    int L = seq_len;
    if (cu) {
        row0 = cu[b];
        L = cu[b + 1] - cu[b];
        S = L;
        base = qkv + row0 * row_stride;
    }
    if (qt * 64 >= S) return;
    const int qi = qt * 64 + threadIdx.x;
    const bool active = qi < S;
    const float* Qp = base + (int64_t)(q_off + h) * HD;
    const float* Kp = base + (int64_t)(k_off + h) * HD;
    const float* Vp = base + (int64_t)(v_off + h) * HD;
    float q[HD], o[HD];
    float m = -INFINITY, l = 0.f;
    if (active) {
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] = Qp[(int64_t)qi * row_stride + d];
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    // keys visible to this tile's queries: [k_lo, k_hi)
    const int q_last = min(qt * 64 + 63, S - 1);
    const int k_hi = causal ? min(L, q_last + 1) : L;
    const int k_lo = (window > 0 && qt * 64 - window + 1 > 0) ? (qt * 64 - window + 1) / TK * TK : 0;
    const int kmax = causal ? min(qi, L - 1) : L - 1;                 // this query's last / first visible key
    const int kmin = window > 0 ? qi - window + 1 : 0;
    for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
        const int cnt = min(TK, L - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < TK * HD; i += 64) {                // rows beyond the sequence: zeros (their scores are masked, 0 * 0 stays 0)
            const int r = i / HD, d = i % HD;
            ks[r][d] = r < cnt ? Kp[(int64_t)(k0 + r) * row_stride + d] : 0.f;
            vs[r][d] = r < cnt ? Vp[(int64_t)(k0 + r) * row_stride + d] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        for (int r = 0; r < cnt; r += 4) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < HD; ++d) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] = fmaf(q[d], ks[min(r + u, TK - 1)][d], s[u]);
            }
            bool any = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int key = k0 + r + u;
                const bool vis = r + u < cnt && key <= kmax && key >= kmin;
                s[u] = vis ? s[u] * scale : -INFINITY;
                any |= vis;
            }
            if (!any) continue;                                            // (keeps m finite from the first visible key on)
            const float mn = fmaxf(fmaxf(m, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
            const float a = expf(m - mn);                                 // m = -inf at the first visible step: exp(-inf) = 0
            float p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = expf(s[u] - mn);
            l = l * a + ((p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                float acc = o[d] * a;
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fmaf(p[u], vs[min(r + u, TK - 1)][d], acc);
                o[d] = acc;
            }
            m = mn;
        }
    }
    if (active) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        float* op = out + ((row0 + qi) * H + h) * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = o[d] * inv;
    }
}

// y = x * rsqrt(mean(x^2) + eps) * w : one wave per row
__global__ void __launch_bounds__(256)
k_rms_norm_f32(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int rows, int D, int64_t ldx, int64_t ldy, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * ldx;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xr + c);
        ss += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    float* yr = y + (int64_t)row * ldy;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xr + c), g = *reinterpret_cast<const float4*>(w + c);
        *reinterpret_cast<float4*>(yr + c) = float4{a.x * rstd * g.x, a.y * rstd * g.y, a.z * rstd * g.z, a.w * rstd * g.w};
    }
}

// one thread = 4 consecutive rotation pairs of one head
__global__ void k_rope_f32(float* __restrict__ qkv, const float* __restrict__ cos_t, const float* __restrict__ sin_t, int rows, int S, int n_rot_heads,
                           int hd, int64_t ld, const int32_t* __restrict__ pos_of_row) {
    const int half = hd >> 1, cpb = half >> 2;
    const int per_row = n_rot_heads * cpb;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * per_row) return;
    const int row = (int)(i / per_row), r = (int)(i % per_row);
    const int h = r / cpb, p = (r % cpb) * 4;
    const int pos = pos_of_row ? pos_of_row[row] : row % S;
    float* base = qkv + (int64_t)row * ld + h * hd + p;
    const float4 a = *reinterpret_cast<const float4*>(base), b = *reinterpret_cast<const float4*>(base + half);
    const float4 c = *reinterpret_cast<const float4*>(cos_t + (int64_t)pos * half + p), s = *reinterpret_cast<const float4*>(sin_t + (int64_t)pos * half + p);
    *reinterpret_cast<float4*>(base) = float4{a.x * c.x - b.x * s.x, a.y * c.y - b.y * s.y, a.z * c.z - b.z * s.z, a.w * c.w - b.w * s.w};
    *reinterpret_cast<float4*>(base + half) = float4{b.x * c.x + a.x * s.x, b.y * c.y + a.y * s.y, b.z * c.z + a.z * s.z, b.w * c.w + a.w * s.w};
}

__device__ __forceinline__ float silu(float g) { return g / (1.0f + expf(-g)); }

__global__ void k_swiglu_f32(const float* __restrict__ gu, float* __restrict__ out, int64_t rows, int I) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * I) return;
    const int64_t r = i / I;
    const int c = (int)(i % I);
    const float4 g = *reinterpret_cast<const float4*>(gu + r * 2 * I + c), u = *reinterpret_cast<const float4*>(gu + r * 2 * I + I + c);
    *reinterpret_cast<float4*>(out + i) = float4{u.x * silu(g.x), u.y * silu(g.y), u.z * silu(g.z), u.w * silu(g.w)};
}

__global__ void k_patchify_f32(const float* __restrict__ px, float* __restrict__ out, int B, int S, int P, int Kp) {
    const int G = S / P, K = 3 * P * P, cpr = Kp >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * G * G * cpr) return;
    const int64_t row = i / cpr;
    const int k0 = (int)(i % cpr) * 4;
    const int b = (int)(row / (G * G)), g = (int)(row % (G * G));
    const int gy = g / G, gx = g % G;
    const float* img = px + (int64_t)b * 3 * S * S;
    float o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u;
        float v = 0.f;
        if (k < K) {
            const int c = k / (P * P), r = k % (P * P);
            v = img[((int64_t)c * S + gy * P + r / P) * S + gx * P + r % P];
        }
        o[u] = v;
    }
    *reinterpret_cast<float4*>(out + row * Kp + k0) = float4{o[0], o[1], o[2], o[3]};
}

// one wave per output row (b, t): t == 0 -> cls, else patch row b (L - 1) + t - 1; + pos[t]; LayerNorm (two-pass variance)
__global__ void __launch_bounds__(256)
k_vit_embed_ln_f32(const float* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos, const float* __restrict__ w,
                   const float* __restrict__ bia, float* __restrict__ y, int rows, int L, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / L, t = row % L;
    const float* src = t == 0 ? cls : patch + ((int64_t)b * (L - 1) + t - 1) * D;
    const float* pr = pos + (int64_t)t * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += src[c] + pr[c];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float d = (src[c] + pr[c]) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float* yr = y + (int64_t)row * D;
    for (int c = lane; c < D; c += 64) yr[c] = ((src[c] + pr[c]) - mean) * rstd * w[c] + bia[c];
}

__global__ void __launch_bounds__(256)
k_assemble_prompt_f32(const uint32_t* __restrict__ desc, const float* __restrict__ embed, const float* __restrict__ patch_feat,
                      const float* __restrict__ patch_pos, const float* __restrict__ inst, const float* __restrict__ zone, float* __restrict__ out,
                      int rows, int D) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= rows) return;
    const uint32_t d = desc[t];
    const uint32_t src = d >> 28;
    const int64_t r = (int64_t)(d & 0x0FFFFFFFu) * D;
    float* o = out + (int64_t)t * D;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (src == 0) {
            v = *reinterpret_cast<const float4*>(embed + r + c);
        } else if (src == 1) {
            const float4 a = *reinterpret_cast<const float4*>(patch_feat + r + c), b = *reinterpret_cast<const float4*>(patch_pos + r + c);
            v = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
        } else if (src == 2) {
            v = *reinterpret_cast<const float4*>(inst + r + c);
        } else if (src == 3) {
            v = *reinterpret_cast<const float4*>(zone + r + c);
        }
        *reinterpret_cast<float4*>(o + c) = v;
    }
}

}  // namespace

extern "C" {

int32_t d3d_attention_f32(const float* qkv, float* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                          int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens, int32_t window,
                          void* stream) {
    if (B <= 0 || S <= 0) return D3D_OK;
    if ((head_dim != 64 && head_dim != 96) || window < 0 || (window > 0 && !causal) || seq_len > S || seq_len <= 0) {
        d3d_set_error_("d3d_attention_f32: head_dim 64 or 96; a window needs causal; 0 < seq_len <= S");
        return D3D_EINVAL;
    }
    if (window >= S) window = 0;
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid((unsigned)B, (unsigned)H, (unsigned)((S + 63) / 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (head_dim == 64)
        hipLaunchKernelGGL((k_attention_f32<64>), grid, block, 0, s, qkv, out, S, H, row_stride, batch_stride, q_off, k_off, v_off, scale, seq_len, cu_seqlens,
                           causal, window);
    else
        hipLaunchKernelGGL((k_attention_f32<96>), grid, block, 0, s, qkv, out, S, H, row_stride, batch_stride, q_off, k_off, v_off, scale, seq_len, cu_seqlens,
                           causal, window);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_rms_norm_f32(const float* x, const float* w, float* y, int32_t rows, int32_t D, int64_t ldx, int64_t ldy, float eps, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 4 != 0 || (ldx & 3) || (ldy & 3)) {
        d3d_set_error_("d3d_rms_norm_f32: D, ldx, ldy must be multiples of 4");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_rms_norm_f32, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, y, rows, D, ldx, ldy, eps);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_rope_inplace_f32(float* qkv, const float* cos_t, const float* sin_t, int32_t rows, int32_t S, int32_t n_rot_heads, int32_t head_dim,
                             int64_t ld, const int32_t* pos_of_row, void* stream) {
    if (rows <= 0) return D3D_OK;
    if ((head_dim / 2) % 4 != 0 || (ld & 3)) {
        d3d_set_error_("d3d_rope_inplace_f32: head_dim / 2 and ld must be multiples of 4");
        return D3D_EINVAL;
    }
    const int64_t n = (int64_t)rows * n_rot_heads * (head_dim / 8);
    hipLaunchKernelGGL(k_rope_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qkv, cos_t, sin_t, rows, S, n_rot_heads, head_dim, ld,
                       pos_of_row);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_swiglu_f32(const float* gate_up, float* out, int64_t rows, int32_t I, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (I % 4) {
        d3d_set_error_("d3d_swiglu_f32: I must be a multiple of 4");
        return D3D_EINVAL;
    }
    const int64_t n = rows * I / 4;
    hipLaunchKernelGGL(k_swiglu_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gate_up, out, rows, I);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_patchify_f32(const float* pixels, float* out, int32_t B, int32_t S, int32_t patch, int32_t Kp, void* stream) {
    if (B <= 0) return D3D_OK;
    if (patch <= 0 || S % patch != 0 || Kp % 4 != 0 || Kp < 3 * patch * patch) {
        d3d_set_error_("d3d_patchify_f32: need S % patch == 0, Kp % 4 == 0, Kp >= 3 * patch^2");
        return D3D_EINVAL;
    }
    const int G = S / patch;
    const int64_t n = (int64_t)B * G * G * (Kp / 4);
    hipLaunchKernelGGL(k_patchify_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pixels, out, B, S, patch, Kp);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_vit_embed_ln_f32(const float* patch_rows, const float* cls, const float* pos, const float* ln_w, const float* ln_b, float* y, int32_t B,
                             int32_t L, int32_t D, float eps, void* stream) {
    if (B <= 0) return D3D_OK;
    if (L < 2 || D <= 0) {
        d3d_set_error_("d3d_vit_embed_ln_f32: L >= 2 and D > 0 required");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_vit_embed_ln_f32, dim3((B * L + 3) / 4), dim3(256), 0, (hipStream_t)stream, patch_rows, cls, pos, ln_w, ln_b, y, B * L, L, D, eps);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_assemble_prompt_f32(const uint32_t* desc, const float* embed, const float* patch_feat, const float* patch_pos, const float* inst,
                                const float* zone, float* out, int32_t rows, int32_t D, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 4 != 0 || !desc || !out) {
        d3d_set_error_("d3d_assemble_prompt_f32: need D % 4 == 0, desc, out");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_assemble_prompt_f32, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, desc, embed, patch_feat, patch_pos, inst, zone, out, rows,
                       D);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
