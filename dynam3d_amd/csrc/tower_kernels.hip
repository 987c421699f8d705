// tower_kernels.hip -- the ViT front end around the patch-embedding GEMM (a3 / a15):
//
//   d3d_patchify      normalised float32 CHW pixels -> the GEMM's A operand: one row per 14x14 patch, k = c*P*P + i*P + j
//                     (the order of conv1.weight.reshape(width, 3*P*P), clip/model.py:206, 222), cast to the tower's 16-bit
//                     dtype (the reference feeds the conv fp16 / bf16 pixels) and ZERO-PADDED to `Kp` columns so that K is a
//                     multiple of the GEMM's 64-deep K tile (3*14*14 = 588 -> 640).  Replaces .to(dtype) + permute + reshape.
//   d3d_vit_embed_ln  [cls; patch rows] + positional embedding -> 16-bit (the reference's `x + positional_embedding` is a
//                     16-bit add, clip/model.py:226-227) -> ln_pre (float32 statistics, clip/model.py:153-159) -> 16-bit.
//                     Replaces cat + add + LayerNorm launches.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

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

// one thread = 8 consecutive k of one patch row (one 16-byte store)
template <bool BF16>
__global__ void k_patchify(const float* __restrict__ px, uint16_t* __restrict__ out, int B, int S, int P, int Kp) {
    const int G = S / P, K = 3 * P * P, cpr = Kp >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * G * G * cpr) return;
    const int64_t row = i / cpr;
    const int k0 = (int)(i % cpr) * 8;
    const int b = (int)(row / (G * G)), g = (int)(row % (G * G));
    const int gy = g / G, gx = g % G;
    const float* img = px + (int64_t)b * 3 * S * S;
    uint16_t o[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = k0 + u;
        float v = 0.f;
        if (k < K) {
            const int c = k / (P * P), r = k % (P * P);
            v = img[((int64_t)c * S + gy * P + r / P) * S + gx * P + r % P];
        }
        o[u] = st16<BF16>(v);
    }
    *reinterpret_cast<uint4*>(out + row * Kp + k0) = *reinterpret_cast<const uint4*>(o);
}

// One wave per output row (b, t): t == 0 -> cls, else patch row b*(L-1) + t-1; + pos[t]; round; LayerNorm; round.
template <bool BF16, int NCH>
__global__ void __launch_bounds__(256)
k_vit_embed_ln(const uint16_t* __restrict__ patch, const uint16_t* __restrict__ cls, const uint16_t* __restrict__ pos, const float* __restrict__ w,
               const float* __restrict__ bia, uint16_t* __restrict__ y, int rows, int L, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / L, t = row % L;
    const uint16_t* src = t == 0 ? cls : patch + ((int64_t)b * (L - 1) + t - 1) * D;
    const uint16_t* pr = pos + (int64_t)t * D;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 512 + lane * 8;
        if (off < D) {
            const uint4 a = *reinterpret_cast<const uint4*>(src + off), p = *reinterpret_cast<const uint4*>(pr + off);
            const uint16_t* ah = reinterpret_cast<const uint16_t*>(&a);
            const uint16_t* ph = reinterpret_cast<const uint16_t*>(&p);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = ld16<BF16>(st16<BF16>(ld16<BF16>(ah[j]) + ld16<BF16>(ph[j])));     // the reference's 16-bit add
                s += v[c][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)D;
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
    const float rstd = rsqrtf(q / (float)D + eps);
    uint16_t* yr = y + (int64_t)row * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 512 + lane * 8;
        if (off < D) {
            uint16_t o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = st16<BF16>((v[c][j] - mean) * rstd * w[off + j] + bia[off + j]);
            *reinterpret_cast<uint4*>(yr + off) = *reinterpret_cast<const uint4*>(o);
        }
    }
}

template <bool BF16>
int32_t launch_embed_ln(const void* patch, const void* cls, const void* pos, const float* w, const float* b, void* y, int rows, int L, int D,
                        float eps, hipStream_t s) {
    const int nch = (D + 511) / 512;
    dim3 grid((rows + 3) / 4), block(256);
#define D3D_EMB_CASE(N)                                                                                                                  \
    case N:                                                                                                                              \
        hipLaunchKernelGGL((k_vit_embed_ln<BF16, N>), grid, block, 0, s, (const uint16_t*)patch, (const uint16_t*)cls, (const uint16_t*)pos, w, b, \
                           (uint16_t*)y, rows, L, D, eps);                                                                               \
        break;
    switch (nch) {
        D3D_EMB_CASE(1)
        D3D_EMB_CASE(2)
        D3D_EMB_CASE(3)
        D3D_EMB_CASE(4)
        default:
            d3d_set_error_("d3d_vit_embed_ln: width must be <= 2048");
            return D3D_EINVAL;
    }
#undef D3D_EMB_CASE
    D3D_LAUNCH_CHECK();
}

// The packed prompt of all environments in ONE pass (VLN-POL:448-456): output row t is, by its descriptor desc[t] = (source << 28) | row,
//   source 0: embedding-table row `row` (the text in front of / behind the visual prefix);
//   source 1: patch token `row` = (llava patch feature + patch position token), added in float32 and rounded once (VLN-POL:448-453);
//   source 2 / 3: instance / zone token `row`;            source 7: a zero row (padding up to a multiple of 256 rows).
// One wave per row, 16 bytes per lane and trip.  Replaces an embedding gather, two float up-casts, an add, a down-cast, a cat, a fill
// and a row gather (~0.45 ms of launches and 8 passes over ~40 MB per step).
template <bool BF16>
__global__ void __launch_bounds__(256)
k_assemble_prompt(const uint32_t* __restrict__ desc, const uint16_t* __restrict__ embed, const uint16_t* __restrict__ patch_feat,
                  const uint16_t* __restrict__ patch_pos, const uint16_t* __restrict__ inst, const uint16_t* __restrict__ zone,
                  uint16_t* __restrict__ out, int rows, int D) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= rows) return;
    const uint32_t d = desc[t];
    const uint32_t src = d >> 28;
    const int64_t r = (int64_t)(d & 0x0FFFFFFFu) * D;
    uint16_t* o = out + (int64_t)t * D;
    for (int c = lane * 8; c < D; c += 512) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src == 0) {
            v = *reinterpret_cast<const uint4*>(embed + r + c);
        } else if (src == 1) {
            const uint4 a = *reinterpret_cast<const uint4*>(patch_feat + r + c), b = *reinterpret_cast<const uint4*>(patch_pos + r + c);
            const uint32_t* a4 = reinterpret_cast<const uint32_t*>(&a);
            const uint32_t* b4 = reinterpret_cast<const uint32_t*>(&b);
            uint32_t* v4 = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lo = ld16<BF16>((uint16_t)(a4[i] & 0xFFFFu)) + ld16<BF16>((uint16_t)(b4[i] & 0xFFFFu));
                const float hi = ld16<BF16>((uint16_t)(a4[i] >> 16)) + ld16<BF16>((uint16_t)(b4[i] >> 16));
                v4[i] = (uint32_t)st16<BF16>(lo) | ((uint32_t)st16<BF16>(hi) << 16);
            }
        } else if (src == 2) {
            v = *reinterpret_cast<const uint4*>(inst + r + c);
        } else if (src == 3) {
            v = *reinterpret_cast<const uint4*>(zone + r + c);
        }
        *reinterpret_cast<uint4*>(o + c) = v;
    }
}

}  // namespace

extern "C" {

int32_t d3d_patchify(const float* pixels, void* out, int32_t B, int32_t S, int32_t patch, int32_t Kp, int32_t dtype, void* stream) {
    if (B <= 0) return D3D_OK;
    if (patch <= 0 || S % patch != 0 || Kp % 8 != 0 || Kp < 3 * patch * patch) {
        d3d_set_error_("d3d_patchify: need S % patch == 0, Kp % 8 == 0, Kp >= 3 * patch^2");
        return D3D_EINVAL;
    }
    const int G = S / patch;
    const int64_t n = (int64_t)B * G * G * (Kp / 8);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(k_patchify<true>, grid, block, 0, (hipStream_t)stream, pixels, (uint16_t*)out, B, S, patch, Kp);
    else
        hipLaunchKernelGGL(k_patchify<false>, grid, block, 0, (hipStream_t)stream, pixels, (uint16_t*)out, B, S, patch, Kp);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_vit_embed_ln(const void* patch_rows, const void* cls, const void* pos, const float* ln_w, const float* ln_b, void* y, int32_t B,
                         int32_t L, int32_t D, float eps, int32_t dtype, void* stream) {
    if (B <= 0) return D3D_OK;
    if (D % 8 != 0 || L < 2) {
        d3d_set_error_("d3d_vit_embed_ln: D % 8 == 0 and L >= 2 required");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    return dtype == 0 ? launch_embed_ln<true>(patch_rows, cls, pos, ln_w, ln_b, y, B * L, L, D, eps, s)
                      : launch_embed_ln<false>(patch_rows, cls, pos, ln_w, ln_b, y, B * L, L, D, eps, s);
}

int32_t d3d_assemble_prompt(const uint32_t* desc, const void* embed, const void* patch_feat, const void* patch_pos, const void* inst, const void* zone,
                            void* out, int32_t rows, int32_t D, int32_t dtype, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 8 != 0 || !desc || !out) {
        d3d_set_error_("d3d_assemble_prompt: need D % 8 == 0, desc, out");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL((k_assemble_prompt<true>), grid, block, 0, s, desc, (const uint16_t*)embed, (const uint16_t*)patch_feat, (const uint16_t*)patch_pos,
                           (const uint16_t*)inst, (const uint16_t*)zone, (uint16_t*)out, rows, D);
    else
        hipLaunchKernelGGL((k_assemble_prompt<false>), grid, block, 0, s, desc, (const uint16_t*)embed, (const uint16_t*)patch_feat, (const uint16_t*)patch_pos,
                           (const uint16_t*)inst, (const uint16_t*)zone, (uint16_t*)out, rows, D);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
