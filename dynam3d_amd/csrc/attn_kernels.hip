// attn_kernels.hip -- fused (flash-style) self-attention forward for gfx950: bf16/fp16, head_dim 64 (ViT-L towers) and
// 96 (Phi-3-mini), causal or full, reading q/k/v straight out of the fused QKV projection buffer (no transposes,
// no materialised S x S scores).
//
// Mapping (wave64, v_mfma_f32_16x16x32):
//   * workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries (two 16-wide q-tiles) whose
//     Q fragments stay in registers; K/V tiles of 64 keys are staged once per workgroup in LDS.
//   * scores are computed TRANSPOSED: S^T = K . Q^T  (A = K fragment, B = Q fragment), so a lane holds, for ONE query
//     (column lane&15), keys 4*(lane>>4)+r of every 16-key tile: the softmax row reduction is register-local plus two
//     xor-shuffles (lanes +-16, +-32), and P^T is already in the B-operand layout of the second product
//     O^T = V^T . P^T  (A = V^T fragment from a transposed LDS image, B = P^T packed to 16 bit).
//   * the k-slot order inside a 32-deep MFMA step is a free permutation as long as A and B agree; here slots 0-3 are
//     keys 4g..4g+3 of the even 16-key tile and slots 4-7 the same rows of the odd tile (g = lane>>4), which is exactly
//     what the S^T accumulators hold -- no cross-lane data movement between the two GEMMs.
//   * O^T accumulators put 4 consecutive head-dim elements of one query in a lane: 8-byte packed stores.
//   * K rows are XOR-swizzled on 16-byte chunks, V^T rows padded by 4 elements: conflict-free ds_read_b128 / ds_read_b64.
//   * V: staged row-major and transposed by the LDS read itself (ds_read_b64_tr_b16, template flag VTR; the default), or
//     pre-transposed once per call into a workspace (k_transpose_v); K/V tiles are prefetched into registers one tile ahead.
//   * softmax VALU diet (the loop is partly VALU-bound: 34 % VALU-active against 13 % MFMA-busy per SIMD at the ViT shape):
//     three-input maxima, packed f32 fma / add for the exponent arguments and row sums, row reductions through
//     v_permlane16/32_swap instead of ds_bpermute: -4.5 % kernel time, measured A/B with alternating libraries.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <stdlib.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float4v = __attribute__((ext_vector_type(4))) float;
using v4s = __attribute__((ext_vector_type(4))) short;

// Cross-lane reductions over the four 16-lane rows of a wave (the lanes that share one query column) in the VALU:
// v_permlane16_swap exchanges the odd rows of vdst with the even rows of src, v_permlane32_swap the upper half of vdst
// with the lower half of src; with both operands holding the same value, op(vdst', src') is the xor-16 / xor-32 butterfly.
// __shfl_xor goes through ds_bpermute -- an LDS round trip per step, four of them on the critical path of every query tile.
// (The builtin folds away when both operands are the same SSA value, hence asm; s_nop 1 = the two wait states a VALU write
// needs before v_permlane*_swap reads it, LLVM gfx950 hazard rule.)
using float2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float max3(float a, float b, float c) {
    // NOT inline asm: the hazard recogniser does not see an asm statement as a VALU read, so the wait states a VALU needs after the MFMA
    // that wrote its operands were not inserted and a max taken straight off the score accumulators read registers still in flight
    // (run-to-run differences in the last bits).  hipcc folds the nested maxima into v_max3_f32 by itself.
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
__device__ __forceinline__ float row_max4(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}
__device__ __forceinline__ float row_sum4(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}

constexpr int BQ = 128, BKV = 64, NT = 256;

template <bool BF16>
__device__ __forceinline__ float4v mfma16(const uint4& a, const uint4& b, float4v c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&a), *reinterpret_cast<const half8*>(&b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (BF16) {
        // fptrunc <2 x float> -> <2 x bfloat> selects v_cvt_pk_bf16_f32 (RNE, NaN-safe); NOT inline asm: the hazard recogniser does not
        // see through asm, and a conversion scheduled right behind the MFMA that produced its operand reads a stale register
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
        return *reinterpret_cast<const uint32_t*>(&r);
    } else {
        const __half2 h = __floats2half2_rn(lo, hi);
        return *reinterpret_cast<const uint32_t*>(&h);
    }
}

// V (B,S,.,hd) token-major  ->  V^T (B,H,hd,Sp) key-major (Sp = S rounded up to 64, tail zero-filled), once per layer:
// the MFMA A-operand of O^T = V^T P^T needs 8 consecutive KEYS per lane, so the transposition is done once here
// (each element passes LDS one time) instead of once per query block inside the attention kernel.
template <int HD>
__global__ void __launch_bounds__(256)
k_transpose_v(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ vt, int S, int Sp, int H, int64_t row_stride, int64_t batch_stride,
              int v_off, const int32_t* __restrict__ cu) {
    constexpr int CH = HD / 8;
    __shared__ uint16_t tile[HD][64 + 2];
    const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    if (cu) {                                   // packed batch: sequence b = rows [cu[b], cu[b+1])
        S = cu[b + 1] - cu[b];
        if (k0 >= S) return;
    }
    const uint16_t* Vp = qkv + (cu ? (int64_t)cu[b] * row_stride : (int64_t)b * batch_stride) + (int64_t)(v_off + h) * HD;
    for (int c = threadIdx.x; c < 64 * CH; c += 256) {
        const int key = c / CH, dc = c % CH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k0 + key < S) v = *reinterpret_cast<const uint4*>(Vp + (int64_t)(k0 + key) * row_stride + dc * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[dc * 8 + j][key] = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
    }
    __syncthreads();
    uint16_t* dst = vt + (((int64_t)b * H + h) * HD) * Sp + k0;
    for (int c = threadIdx.x; c < HD * 8; c += 256) {        // 8 chunks of 8 keys per head-dim row
        const int d = c / 8, kc = c % 8;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[d][kc * 8 + 2 * j] | ((uint32_t)tile[d][kc * 8 + 2 * j + 1] << 16);
        *reinterpret_cast<uint4*>(dst + (int64_t)d * Sp + kc * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// VTR: V is staged ROW-major straight from the fused projection buffer and the A operand of O^T = V^T P^T is fetched with the
// transposing LDS read of gfx950 (ds_read_b64_tr_b16: the 16 lanes of a group address one [4 keys][16 dims] block, 8 bytes each,
// and lane j receives column j of it -- exactly "4 consecutive keys of one head-dim element").  No k_transpose_v pass, no V^T
// workspace.  V rows are padded to an odd multiple of 32 bytes: the 8 rows a half-wave touches then fall into 8 different
// 32-byte bank groups.
template <bool BF16, int HD, bool CAUSAL, bool VTR>
__global__ void __launch_bounds__(NT, 2)          // 2 waves/SIMD = 2 workgroups per CU: keep VGPR+AGPR <= 256
k_flash_attn(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ vt, int Sp, uint16_t* __restrict__ out, int S, int H,
             int64_t row_stride /* elements between tokens */, int64_t batch_stride, int q_off, int k_off, float scale_log2e, int seq_len,
             const int32_t* __restrict__ cu /* packed batch: (B+1) row offsets, or null */, int n_qblocks, int v_off) {
    // K rows are padded to a power-of-two number of 16-byte chunks and XOR-swizzled (chunk ^= row & (KCH-1)): every
    // ds_read_b128 lane group (which mixes two k-groups, e.g. lanes {0-3,12-15,20-27}) then hits 16 distinct slots.
    constexpr int KCH = HD == 96 ? 16 : 8;   // chunk positions per LDS row
    constexpr int KST = KCH * 8;             // K row stride (elements): 256 B (hd 96) / 128 B (hd 64)
    constexpr int VST = BKV + 4;             // V^T row stride 136 B: dword stride 34 -> conflict-free 8-byte reads
    constexpr int VSTR = HD + 16;            // VTR: row-major V rows, 224 B (hd 96) / 160 B (hd 64) = odd multiples of 32 B
    constexpr int DS = HD / 32;          // 32-deep steps over head_dim (QK^T)
    constexpr int DT = HD / 16;          // 16-wide head-dim tiles (PV)
    constexpr int CH = HD / 8;           // 16-byte chunks per row
    __shared__ __attribute__((aligned(16))) uint16_t Ks[BKV * KST];
    __shared__ __attribute__((aligned(16))) uint16_t Vt[VTR ? BKV * VSTR : HD * VST];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    // Query blocks.  CAUSAL: a workgroup takes the PAIR (n-1-x, x) -- the longest block with the shortest, and so on -- so that
    // every workgroup walks the same number of key tiles (2x+2 + 2(n-1-x)+2); one block per workgroup left the last
    // rounds to whatever lengths came last.  The pair index is rotated by head + batch: workgroups go to the 8 XCDs
    // round-robin by linear id, and without the rotation one XCD would receive every workgroup of one x.
    const int h = blockIdx.y, b = blockIdx.z;
    const int xq = (int)((blockIdx.x + blockIdx.y + blockIdx.z) % gridDim.x);
    int64_t row0 = (int64_t)b * S;                         // first output row of this sequence
    const uint16_t* base = qkv + (int64_t)b * batch_stride;
    if (cu) {                                              // packed batch: this sequence's own length and block count
        row0 = cu[b];
        S = cu[b + 1] - cu[b];
        seq_len = S;
        n_qblocks = (S + BQ - 1) / BQ;
        base = qkv + row0 * row_stride;
    }
    if (CAUSAL ? xq >= (n_qblocks + 1) / 2 : xq >= n_qblocks) return;   // (uniform) beyond this sequence
  for (int pass = 0; pass < (CAUSAL ? 2 : 1); ++pass) {
    const int qb = CAUSAL ? (pass == 0 ? n_qblocks - 1 - xq : xq) : xq;
    if (CAUSAL && pass == 1 && qb == n_qblocks - 1 - xq) break;          // odd count: the middle block stands alone
    const int q0 = qb * BQ, qw = q0 + wave * 32;
    const uint16_t* Qp = base + (int64_t)(q_off + h) * HD;
    const uint16_t* Kp = base + (int64_t)(k_off + h) * HD;
    const uint16_t* Vtp = VTR ? base + (int64_t)(v_off + h) * HD : vt + (((int64_t)b * H + h) * HD) * Sp;   // token-major V / (hd, Sp) key-major V^T

    // Q fragments (B operand): lane holds Q[q = qw + qt*16 + fi][d = ks*32 + fg*8 .. +7]
    uint4 qf[2][DS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int q = qw + qt * 16 + fi;
        q = q < S ? q : S - 1;
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) qf[qt][ks] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * row_stride + ks * 32 + fg * 8);
    }
    float4v oacc[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = float4v{0.f, 0.f, 0.f, 0.f};
    float m_i[2] = {-INFINITY, -INFINITY}, l_i[2] = {0.f, 0.f};

    const int kv_len = CAUSAL ? min(seq_len, q0 + BQ) : seq_len;
    const int n_tiles = (kv_len + BKV - 1) / BKV;

    // register-staged prefetch (issue-early / write-late): tile t+1 is loaded into VGPRs before tile t is computed and
    // written to LDS after it, so the global-load latency hides under the MFMA + softmax work of the current tile.
    constexpr int NK = (BKV * CH) / NT, NV = VTR ? NK : (HD * 8) / NT;      // 3,3 (hd 96) or 2,2 (hd 64) 16-byte chunks per thread
    static_assert(NK <= 3 && NV <= 3, "tile shape");
    // named scalars (not arrays: loop-carried uint4 arrays were left in scratch memory by the compiler)
    uint4 kr0, kr1, kr2 = make_uint4(0, 0, 0, 0), vr0, vr1, vr2 = make_uint4(0, 0, 0, 0);
    auto ld_k = [&](int key0_, int i) -> uint4 {
        const int c = tid + i * NT;
        int kr = key0_ + c / CH;
        kr = kr < S ? kr : S - 1;
        return *reinterpret_cast<const uint4*>(Kp + (int64_t)kr * row_stride + (c % CH) * 8);
    };
    auto ld_v = [&](int key0_, int i) -> uint4 {
        const int c = tid + i * NT;
        if constexpr (VTR) {                                                                       // (key, 8-dim chunk), like K
            int kr = key0_ + c / CH;
            kr = kr < S ? kr : S - 1;
            return *reinterpret_cast<const uint4*>(Vtp + (int64_t)kr * row_stride + (c % CH) * 8);
        } else {                                                                                   // (d, 8-key chunk)
            return *reinterpret_cast<const uint4*>(Vtp + (int64_t)(c >> 3) * Sp + key0_ + (c & 7) * 8);
        }
    };
    auto st_k = [&](int i, const uint4& v) {
        const int c = tid + i * NT;
        const int r = c / CH;
        *reinterpret_cast<uint4*>(Ks + r * KST + (((c % CH) ^ (r & (KCH - 1))) << 3)) = v;         // swizzled K rows
    };
    auto st_v = [&](int i, const uint4& v) {
        const int c = tid + i * NT;
        if constexpr (VTR) {
            *reinterpret_cast<uint4*>(Vt + (c / CH) * VSTR + (c % CH) * 8) = v;                    // row-major V rows (16-byte aligned)
        } else {
            uint16_t* d = Vt + (c >> 3) * VST + (c & 7) * 8;                                      // V^T rows of 64 keys (8-byte aligned)
            *reinterpret_cast<uint2*>(d) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2*>(d + 4) = make_uint2(v.z, v.w);
        }
    };
#define FA_LOAD_TILE(T)                                       \
    {                                                         \
        const int k0_ = (T) * BKV;                            \
        kr0 = ld_k(k0_, 0);                                   \
        kr1 = ld_k(k0_, 1);                                   \
        if constexpr (NK > 2) kr2 = ld_k(k0_, 2);             \
        vr0 = ld_v(k0_, 0);                                   \
        vr1 = ld_v(k0_, 1);                                   \
        if constexpr (NV > 2) vr2 = ld_v(k0_, 2);             \
    }

    FA_LOAD_TILE(0)
    for (int t = 0; t < n_tiles; ++t) {
        const int key0 = t * BKV;
        __syncthreads();                                   // every wave is done reading the previous tile
        st_k(0, kr0);
        st_k(1, kr1);
        if constexpr (NK > 2) st_k(2, kr2);
        st_v(0, vr0);
        st_v(1, vr1);
        if constexpr (NV > 2) st_v(2, vr2);
        __syncthreads();
        if (t + 1 < n_tiles) FA_LOAD_TILE(t + 1)
        if (CAUSAL && key0 > qw + 31) continue;            // whole tile above this wave's diagonal (wave-uniform)
        // masking is only needed on the diagonal tiles of this wave and on the tile that crosses seq_len
        const bool need_mask = (CAUSAL && key0 + BKV - 1 > qw) || (key0 + BKV > seq_len);

        // ---- S^T = K Q^T : st[qt][kt] holds keys key0 + kt*16 + 4*fg + r for query qw + qt*16 + fi ------------
        float4v st[2][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            uint4 kf[DS];
#pragma unroll
            for (int ks = 0; ks < DS; ++ks) {
                const int r = kt * 16 + fi;
                kf[ks] = *reinterpret_cast<const uint4*>(Ks + r * KST + (((ks * 4 + fg) ^ (r & (KCH - 1))) << 3));
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float4v a = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < DS; ++ks) a = mfma16<BF16>(kf[ks], qf[qt][ks], a);
                st[qt][kt] = a;
            }
        }
        // ---- online softmax in base 2 on the raw scores: p = exp2(s*c - m*c), c = log2(e)/sqrt(hd) ---------------------
        uint4 pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (need_mask) {
                const int q = qw + qt * 16 + fi;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + kt * 16 + fg * 4 + r;
                        if (key >= seq_len || (CAUSAL && key > q)) st[qt][kt][r] = -INFINITY;
                    }
            }
            // 16 scores per lane: 8 three-input maxima instead of 15 two-input ones
            float tmax = max3(max3(max3(st[qt][0][0], st[qt][0][1], st[qt][0][2]), max3(st[qt][0][3], st[qt][1][0], st[qt][1][1]),
                                   max3(st[qt][1][2], st[qt][1][3], st[qt][2][0])),
                              max3(st[qt][2][1], st[qt][2][2], st[qt][2][3]), max3(st[qt][3][0], st[qt][3][1], st[qt][3][2]));
            tmax = fmaxf(tmax, st[qt][3][3]);
            tmax = row_max4(tmax);
            // deferred max: keep the old running max while the tile max exceeds it by < 2^8 (P stays <= 256, fine for the
            // fp32 accumulators and the 16-bit P operand); the O / l rescale is then skipped for the whole wave.
            const float tm = tmax * scale_log2e;                              // scaled units (scale > 0 keeps the max)
            const bool keep = __all(tm <= m_i[qt] + 8.0f);
            const float m_new = keep ? m_i[qt] : fmaxf(m_i[qt], tm);
            const float alpha = keep ? 1.0f : __builtin_amdgcn_exp2f(m_i[qt] - m_new);
            // exponent arguments and row sums two scores at a time (v_pk_fma_f32 / v_pk_add_f32); v_exp_f32 stays scalar
            const float2v c2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
            float2v rs2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const float2v x = {st[qt][kt][2 * h2], st[qt][kt][2 * h2 + 1]};
                    const float2v a = x * c2 + nm2;
                    float2v pz;
                    pz.x = __builtin_amdgcn_exp2f(a.x);
                    pz.y = __builtin_amdgcn_exp2f(a.y);
                    st[qt][kt][2 * h2] = pz.x;
                    st[qt][kt][2 * h2 + 1] = pz.y;
                    rs2 += pz;
                }
            float rs = rs2.x + rs2.y;
            rs = row_sum4(rs);
            l_i[qt] = l_i[qt] * alpha + rs;
            m_i[qt] = m_new;
            if (!keep) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) oacc[qt][dt][r] *= alpha;
            }
            // P^T as B operand: k-step kp covers key tiles 2kp (slots 0-3) and 2kp+1 (slots 4-7)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                pf[qt][kp].x = pack2<BF16>(st[qt][2 * kp][0], st[qt][2 * kp][1]);
                pf[qt][kp].y = pack2<BF16>(st[qt][2 * kp][2], st[qt][2 * kp][3]);
                pf[qt][kp].z = pack2<BF16>(st[qt][2 * kp + 1][0], st[qt][2 * kp + 1][1]);
                pf[qt][kp].w = pack2<BF16>(st[qt][2 * kp + 1][2], st[qt][2 * kp + 1][3]);
            }
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                uint4 vf;
                if constexpr (VTR) {
                    // lane (fg, fi) addresses key 4*fg + fi/4 of the 16-key tile, dims (fi%4)*4.. of this 16-dim tile; it gets
                    // back dim fi for keys 4*fg..4*fg+3: slots 0-3 from the even key tile, 4-7 from the odd one
                    using lds_v4s = __attribute__((address_space(3))) v4s;
                    const uint16_t* vblk = Vt + (kp * 32 + fg * 4 + (fi >> 2)) * VSTR + dt * 16 + (fi & 3) * 4;
                    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)vblk);
                    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vblk + 16 * VSTR));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    vf = make_uint4(l2.x, l2.y, h2.x, h2.y);
                } else {
                    const uint16_t* vrow = Vt + (dt * 16 + fi) * VST + kp * 32 + fg * 4;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vrow);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vrow + 16);
                    vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) oacc[qt][dt] = mfma16<BF16>(vf, pf[qt][kp], oacc[qt][dt]);
            }
        }
    }
#undef FA_LOAD_TILE
    // ---- epilogue: lane holds O[q = qw + qt*16 + fi][d = dt*16 + 4*fg + r] -----------------------------------------------
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qw + qt * 16 + fi;
        if (q >= S) continue;
        const float inv = 1.0f / l_i[qt];
        uint16_t* op = out + ((row0 + q) * H + h) * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 o;
            o.x = pack2<BF16>(oacc[qt][dt][0] * inv, oacc[qt][dt][1] * inv);
            o.y = pack2<BF16>(oacc[qt][dt][2] * inv, oacc[qt][dt][3] * inv);
            *reinterpret_cast<uint2*>(op + dt * 16 + fg * 4) = o;
        }
    }
  }   // pass
}

// ================================================================================================
// Decode attention (KV cache): ONE query token per sequence against that sequence's prompt keys/values, read in place from the
// prefill's post-RoPE fused-QKV buffer of the layer (packed rows cu[b] .. cu[b+1]), plus the tokens generated so far, kept
// in a small side cache (B, Tmax, H, hd).  The current token's own k/v rows (already rotated, in `qkv_new`) are appended to
// the side cache by this kernel.  One workgroup per (head, sequence): threads own keys for the scores, a block softmax in
// LDS, then waves own key subsets and lanes own head-dim pairs for the weighted value sum.  Reference: the `use_cache`
// branch of HF Phi-3 attention under `llava.generate(..., do_sample=False)` (VLN-POL:463).
// ================================================================================================
template <bool BF16>
__device__ __forceinline__ float cvt16(uint16_t v) {
    if constexpr (BF16) return __uint_as_float((uint32_t)v << 16);
    else return __half2float(*reinterpret_cast<const __half*>(&v));
}

constexpr int DEC_MAX_KEYS = 4096 + 64;

template <bool BF16, int HD>
__global__ void __launch_bounds__(256)
k_decode_attn(const uint16_t* __restrict__ qkv_new /* (B, 3H, HD): this step's rotated q,k and v */, const uint16_t* __restrict__ prompt /* (T, 3H, HD) */,
              const int32_t* __restrict__ cu, uint16_t* __restrict__ knew, uint16_t* __restrict__ vnew /* (B, Tmax, H, HD) */,
              uint16_t* __restrict__ out /* (B, H, HD) */, int H, int t_new, int Tmax, float scale, const float* __restrict__ cos_t,
              const float* __restrict__ sin_t, const int32_t* __restrict__ pos /* RoPE fused: qkv_new is then UN-rotated */,
              int nsplit, float* __restrict__ part /* (B*H, nsplit, HD + 2) */, unsigned* __restrict__ counters /* (B*H), zero */, int fake_hm) {
    __shared__ float qs[HD];
    __shared__ __attribute__((aligned(16))) uint16_t kcur[HD];
    __shared__ float sc[DEC_MAX_KEYS];
    __shared__ float red[8];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rs = (int64_t)3 * H * HD;                                   // fused row stride (elements)
    const int r0 = cu[b], S = cu[b + 1] - r0, L = S + t_new + 1;               // prompt keys + generated (incl. the current one)
    // flash-decoding: blockIdx.z = one of `nsplit` key ranges of this (sequence, head); every range produces an un-normalised partial
    // (max, sum, output), the workgroup that finishes LAST (atomic ticket, no waiting) merges them in range order.  One workgroup per
    // (sequence, head) is 256 workgroups of 4 waves on 256 CUs: the loop is bound by load latency with 1/8 of the chip's wave slots used.
    const int split = blockIdx.z, per = (L + nsplit - 1) / nsplit;
    const int ja = split * per, jb = min(L, ja + per);
    const bool owns_cur = split == nsplit - 1;                                 // the range that holds key L-1 (this token)
    const uint16_t* qrow = qkv_new + (int64_t)b * rs + (int64_t)h * HD;
    // this token's q and k: rotated here (half-split RoPE at position pos[b], same arithmetic as k_rope, results rounded to 16 bit
    // like the prefill's in-place rotation) or taken as they are; k goes to LDS (it is key L-1) and to the side cache, v to the cache
    constexpr int HALF = HD / 2;
    if (tid < HALF) {
        float q1 = cvt16<BF16>(qrow[tid]), q2 = cvt16<BF16>(qrow[tid + HALF]);
        float k1 = cvt16<BF16>(qrow[(int64_t)H * HD + tid]), k2 = cvt16<BF16>(qrow[(int64_t)H * HD + tid + HALF]);
        uint16_t kr1, kr2;
        if (cos_t) {
            const float c = cos_t[(int64_t)pos[b] * HALF + tid], sn = sin_t[(int64_t)pos[b] * HALF + tid];
            // HF apply_rotary_pos_emb on 16-bit tensors: every product is stored before the sum (same as k_rope)
            auto r = [](float f) { return cvt16<BF16>((uint16_t)pack2<BF16>(f, 0.f)); };
            const uint32_t qp = pack2<BF16>(r(q1 * c) - r(q2 * sn), r(q2 * c) + r(q1 * sn));
            const uint32_t kp2 = pack2<BF16>(r(k1 * c) - r(k2 * sn), r(k2 * c) + r(k1 * sn));
            q1 = cvt16<BF16>((uint16_t)qp);
            q2 = cvt16<BF16>((uint16_t)(qp >> 16));
            kr1 = (uint16_t)kp2;
            kr2 = (uint16_t)(kp2 >> 16);
        } else {
            kr1 = qrow[(int64_t)H * HD + tid];
            kr2 = qrow[(int64_t)H * HD + tid + HALF];
        }
        qs[tid] = q1 * scale;
        qs[tid + HALF] = q2 * scale;
        kcur[tid] = kr1;
        kcur[tid + HALF] = kr2;
        if (owns_cur) {
            uint16_t* kd = knew + (((int64_t)b * Tmax + t_new) * H + h) * HD;
            kd[tid] = kr1;
            kd[tid + HALF] = kr2;
        }
    } else if (owns_cur && tid >= 64 && tid < 64 + HD / 8) {
        const int c = tid - 64;
        const uint4 v = *reinterpret_cast<const uint4*>(qrow + (int64_t)2 * H * HD + c * 8);
        *reinterpret_cast<uint4*>(vnew + (((int64_t)b * Tmax + t_new) * H + h) * HD + c * 8) = v;
    }
    __syncthreads();
    // key j: prompt row | earlier generated token (side cache, written by EARLIER launches) | the current token, read from
    // qkv_new itself so that nothing written by this launch is read back by it
    auto krow = [&](int j) -> const uint16_t* {
        if (j < S && fake_hm) return prompt + (((int64_t)b * H + h) * 2 * S + j) * HD;      // (timing experiment, WRONG results: head-major addressing)
        if (j < S) return prompt + (int64_t)(r0 + j) * rs + (int64_t)(H + h) * HD;
        return knew + (((int64_t)b * Tmax + (j - S)) * H + h) * HD;                 // (j == L-1 is served from LDS below)
    };
    auto vrow = [&](int j) -> const uint16_t* {
        if (j < S && fake_hm) return prompt + (((int64_t)b * H + h) * 2 * S + S + j) * HD;
        if (j < S) return prompt + (int64_t)(r0 + j) * rs + (int64_t)(2 * H + h) * HD;
        if (j == L - 1) return qrow + (int64_t)2 * H * HD;
        return vnew + (((int64_t)b * Tmax + (j - S)) * H + h) * HD;
    };
    // ---- scores: thread-per-key dot products (four partial sums), running max
    float tmax = -INFINITY;
    for (int j = ja + tid; j < jb; j += 256) {
        const uint16_t* kp = krow(j);
        const bool cur = j == L - 1;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            const uint4 kv = cur ? *reinterpret_cast<const uint4*>(kcur + c * 8) : *reinterpret_cast<const uint4*>(kp + c * 8);
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&kv);
            s0 += qs[c * 8 + 0] * cvt16<BF16>(e[0]) + qs[c * 8 + 4] * cvt16<BF16>(e[4]);
            s1 += qs[c * 8 + 1] * cvt16<BF16>(e[1]) + qs[c * 8 + 5] * cvt16<BF16>(e[5]);
            s2 += qs[c * 8 + 2] * cvt16<BF16>(e[2]) + qs[c * 8 + 6] * cvt16<BF16>(e[6]);
            s3 += qs[c * 8 + 3] * cvt16<BF16>(e[3]) + qs[c * 8 + 7] * cvt16<BF16>(e[7]);
        }
        const float sdot = (s0 + s1) + (s2 + s3);
        sc[j - ja] = sdot;
        tmax = fmaxf(tmax, sdot);
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, w, 64));
    if (lane == 0) red[wave] = tmax;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int j = ja + tid; j < jb; j += 256) {
        const float p = __expf(sc[j - ja] - m);
        sc[j - ja] = p;
        lsum += p;
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) lsum += __shfl_xor(lsum, w, 64);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float lsum_wg = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = 1.0f / lsum_wg;
    // ---- O = P V: thread = (key group g, 16-byte chunk c of the head dim): 256 / (HD/8) key groups stride through the keys,
    //      eight keys in flight per thread (the loop is bound by load latency, not bandwidth); partial sums meet in LDS
    constexpr int CH = HD / 8;                          // 12 (hd 96) / 8 (hd 64) chunks per value row
    constexpr int NG = 256 / CH;                        // 21 / 32 key groups
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int g = tid / CH, c = tid % CH;
    if (g < NG) {
        int j = ja + g;
        for (; j + 7 * NG < jb; j += 8 * NG) {
            uint4 v[8];
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = *reinterpret_cast<const uint4*>(vrow(j + u * NG) + c * 8);
                p[u] = sc[j + u * NG - ja];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint16_t* e = reinterpret_cast<const uint16_t*>(&v[u]);
#pragma unroll
                for (int d = 0; d < 8; ++d) o[d] += p[u] * cvt16<BF16>(e[d]);
            }
        }
        for (; j < jb; j += NG) {
            const uint4 v = *reinterpret_cast<const uint4*>(vrow(j) + c * 8);
            const float p = sc[j - ja];
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&v);
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += p * cvt16<BF16>(e[d]);
        }
    }
    __syncthreads();                                    // every thread is done with sc[] as probabilities
    float* acc = sc;                                    // reuse: [NG][HD] partial outputs
    if (g < NG) {
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[g * HD + c * 8 + d] = o[d];
    }
    __syncthreads();
    if (nsplit == 1) {
        if (tid < HD / 2) {
            float a0 = 0.f, a1 = 0.f;
            for (int q = 0; q < NG; ++q) {                  // fixed order: deterministic
                a0 += acc[q * HD + 2 * tid];
                a1 += acc[q * HD + 2 * tid + 1];
            }
            *reinterpret_cast<uint32_t*>(out + ((int64_t)b * H + h) * HD + 2 * tid) = pack2<BF16>(a0 * inv, a1 * inv);
        }
        return;
    }
    // partial of this key range: [0 .. HD) un-normalised output, [HD] max, [HD + 1] sum
    float* mine = part + (((int64_t)b * H + h) * nsplit + split) * (HD + 2);
    // (write-through agent-scope stores + agent-scope loads in the merge: no L2 write-back / invalidate fences needed around the ticket)
    if (tid < HD) {
        float a0 = 0.f;
        for (int q = 0; q < NG; ++q) a0 += acc[q * HD + tid];
        __hip_atomic_store(mine + tid, a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (tid == HD) {
        __hip_atomic_store(mine + HD, jb > ja ? m : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + HD + 1, jb > ja ? lsum_wg : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // every wave's write-through stores are acknowledged (the workgroup-
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                  // scope fence alone emits no vmcnt wait on gfx950) ...
    __syncthreads();                                                        // ... before the ticket is taken
    __shared__ unsigned ticket;
    if (tid == 0) ticket = __hip_atomic_fetch_add(counters + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)nsplit - 1) return;         // not the last range of this (sequence, head) to finish
    if (tid < HD / 2) {
        const float* ps = part + ((int64_t)b * H + h) * nsplit * (HD + 2);
        float mx = -INFINITY;
        for (int q = 0; q < nsplit; ++q) mx = fmaxf(mx, __hip_atomic_load(ps + q * (HD + 2) + HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float lt = 0.f, a0 = 0.f, a1 = 0.f;
        for (int q = 0; q < nsplit; ++q) {              // range order: deterministic whichever workgroup merges
            const float* pq = ps + q * (HD + 2);
            const float wgt = __expf(__hip_atomic_load(pq + HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - mx);
            lt += __hip_atomic_load(pq + HD + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * wgt;
            a0 += __hip_atomic_load(pq + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * wgt;
            a1 += __hip_atomic_load(pq + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * wgt;
        }
        const float il = 1.0f / lt;
        *reinterpret_cast<uint32_t*>(out + ((int64_t)b * H + h) * HD + 2 * tid) = pack2<BF16>(a0 * il, a1 * il);
    }
    if (tid == 0) __hip_atomic_store(counters + b * H + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
}

}  // namespace

extern "C" {

// Self-attention over a fused projection buffer.  qkv: (B, S, Htot, hd) 16-bit with q heads at [q_off, q_off+H), k heads
// at [k_off, ..), v heads at [v_off, ..); token stride = row_stride elements, batch stride = batch_stride elements.
// out: (B, S, H, hd) contiguous.  seq_len = number of valid keys (<= S).  dtype 0 = bf16, 1 = fp16; hd in {64, 96}.
// vt_scratch: NULL (V is staged row-major and transposed by ds_read_b64_tr_b16 inside the kernel), or a caller-provided
// (B, H, hd, Sp) 16-bit workspace, Sp = S rounded up to 64, for the variant that pre-transposes V once per call.
// cu_seqlens (optional, device, B+1 ints): PACKED batch -- sequence b occupies rows [cu[b], cu[b+1]) of qkv/out, S is then the
// longest sequence (grid size) and batch_stride / seq_len are ignored.
int32_t d3d_flash_attention(const void* qkv, void* out, void* vt_scratch, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride,
                            int64_t batch_stride, int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len,
                            const int32_t* cu_seqlens, int32_t dtype, void* stream) {
    if (B <= 0 || S <= 0) return D3D_OK;
    if ((head_dim != 64 && head_dim != 96) || (row_stride & 7) || (batch_stride & 7)) {
        d3d_set_error_("d3d_flash_attention: head_dim must be 64 or 96; strides multiples of 8 elements");
        return D3D_EINVAL;
    }
    const float sl2 = 1.4426950408889634f / sqrtf((float)head_dim);
    const int Sp = (S + 63) / 64 * 64;
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    uint16_t* vt = (uint16_t*)vt_scratch;
    dim3 tg(Sp / 64, H, B);
    const bool vtr = vt == nullptr;      // no workspace: V is read row-major and transposed by the LDS read itself
    if (!vtr) {
        if (head_dim == 96) hipLaunchKernelGGL(k_transpose_v<96>, tg, dim3(256), 0, s, q, vt, S, Sp, H, row_stride, batch_stride, v_off, cu_seqlens);
        else hipLaunchKernelGGL(k_transpose_v<64>, tg, dim3(256), 0, s, q, vt, S, Sp, H, row_stride, batch_stride, v_off, cu_seqlens);
    }
    const int nqb = (S + BQ - 1) / BQ;
    dim3 grid(causal ? (nqb + 1) / 2 : nqb, H, B), block(NT);            // causal: one workgroup per PAIR of query blocks
#define D3D_FA(BF, HDV, CA)                                                                                                              \
    do {                                                                                                                                \
        if (vtr)                                                                                                                        \
            hipLaunchKernelGGL((k_flash_attn<BF, HDV, CA, true>), grid, block, 0, s, q, vt, Sp, o, S, H, row_stride, batch_stride, q_off, \
                               k_off, sl2, seq_len, cu_seqlens, nqb, v_off);                                                            \
        else                                                                                                                            \
            hipLaunchKernelGGL((k_flash_attn<BF, HDV, CA, false>), grid, block, 0, s, q, vt, Sp, o, S, H, row_stride, batch_stride,      \
                               q_off, k_off, sl2, seq_len, cu_seqlens, nqb, v_off);                                                     \
    } while (0)
    if (dtype == 0) {
        if (head_dim == 96) { if (causal) D3D_FA(true, 96, true); else D3D_FA(true, 96, false); }
        else { if (causal) D3D_FA(true, 64, true); else D3D_FA(true, 64, false); }
    } else {
        if (head_dim == 96) { if (causal) D3D_FA(false, 96, true); else D3D_FA(false, 96, false); }
        else { if (causal) D3D_FA(false, 64, true); else D3D_FA(false, 64, false); }
    }
#undef D3D_FA
    D3D_LAUNCH_CHECK();
}

// One decode step of causal self-attention with a KV cache (see k_decode_attn).  qkv_new: (B, 3H, hd) this step's fused projection,
// BEFORE RoPE when cos_t / sin_t / pos are given (the kernel rotates q and k itself), after RoPE when they are null; prompt_qkv: the layer's prefill buffer (packed rows, post-RoPE) with cu_seqlens (B+1); knew / vnew: (B, Tmax, H, hd)
// side caches, filled for tokens < t_new by earlier calls -- this call appends token t_new.  out: (B, H, hd).
int32_t d3d_decode_attention(const void* qkv_new, const void* prompt_qkv, const int32_t* cu_seqlens, void* knew, void* vnew, void* out, int32_t B,
                             int32_t H, int32_t head_dim, int32_t t_new, int32_t Tmax, int32_t max_prompt_len, const float* cos_t, const float* sin_t,
                             const int32_t* pos, int32_t dtype, void* stream) {
    if (B <= 0) return D3D_OK;
    if ((head_dim != 64 && head_dim != 96) || t_new < 0 || t_new >= Tmax || max_prompt_len + Tmax > DEC_MAX_KEYS) {
        d3d_set_error_("d3d_decode_attention: head_dim must be 64 or 96, 0 <= t_new < Tmax, prompt + Tmax <= 4160 keys");
        return D3D_EINVAL;
    }
    const float scale = 1.0f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    // Key ranges per (sequence, head) -- D3D_DECODE_SPLIT, default 1.  Measured (tools/bench_decode_attn.py, 8 x 32 heads x 864 keys, K/V of
    // 32 different layers so nothing is cached): 1 range 29.3 us (2.90 TB/s), 2: 30.7, 4: 39.2, 8: 37.1; at 216 keys 13.6 us whatever the
    // split -- the launch is ~8 us of fixed cost (RoPE, two block-wide reductions, the load-latency chain of one pass) plus ~4 TB/s of
    // marginal streaming, and more workgroups buy nothing.  The split path stays as a tested knob.
    const char* fe = getenv("D3D_DECODE_SPLIT");                               // (read per call: the benchmarks sweep it)
    int nsplit = fe ? atoi(fe) : 1;
    nsplit = std::max(1, std::min(std::min(nsplit, 8), (max_prompt_len + t_new + 1) / 128));
    float* part = nullptr;
    unsigned* counters = nullptr;
    if (nsplit > 1) {
        static std::mutex mu;
        static std::unordered_map<hipStream_t, std::pair<void*, size_t>> pool;     // per stream: [counters (B*H) | partials]
        std::lock_guard<std::mutex> lock(mu);
        auto& e = pool[s];
        const size_t need = (size_t)B * H * sizeof(unsigned) + (size_t)B * H * 8 * (head_dim + 2) * sizeof(float);
        if (e.second < need) {
            if (e.first) {
                D3D_HIP(hipStreamSynchronize(s));
                D3D_HIP(hipFree(e.first));
            }
            D3D_HIP(hipMalloc(&e.first, need));
            D3D_HIP(hipMemset(e.first, 0, need));
            e.second = need;
        }
        counters = (unsigned*)e.first;
        part = (float*)((char*)e.first + (size_t)B * H * sizeof(unsigned));
    }
    const int fake_hm = getenv("D3D_DECODE_FAKE_HM") != nullptr;
    dim3 grid(H, B, nsplit), block(256);
#define D3D_DEC(BF, HDV) hipLaunchKernelGGL((k_decode_attn<BF, HDV>), grid, block, 0, s, (const uint16_t*)qkv_new, (const uint16_t*)prompt_qkv, cu_seqlens, \
                                            (uint16_t*)knew, (uint16_t*)vnew, (uint16_t*)out, H, t_new, Tmax, scale, cos_t, sin_t, pos, nsplit, part, counters, fake_hm)
    if (dtype == 0) { if (head_dim == 96) D3D_DEC(true, 96); else D3D_DEC(true, 64); }
    else { if (head_dim == 96) D3D_DEC(false, 96); else D3D_DEC(false, 64); }
#undef D3D_DEC
    D3D_LAUNCH_CHECK();
}


}  // extern "C"
