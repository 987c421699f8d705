// attn4_kernels.hip -- fourth flash-attention forward for gfx950 (round 6): ONE wave per SIMD with the whole 512-register file, 64 query rows
// per wave as TWO 32-row sub-blocks A / B whose phases are software-pipelined against each other inside the wave, so that the matrix pipe
// never waits for the softmax:
//
//     slot 1: MFMA  S_A(t)  = K(t) Q_A^T        |  VALU  softmax part 2 of B, tile t-1  (exp of key block 1, row sum, P -> 16 bit)
//     slot 2: MFMA  S_B(t)  = K(t) Q_B^T        |  VALU  softmax part 1 of A, tile t    (mask, row max, rescale vote, exp of key block 0)
//     slot 3: MFMA  O_B    += V(t-1)^T P_B(t-1) |  VALU  softmax part 2 of A, tile t
//     slot 4: MFMA  O_A    += V(t)^T   P_A(t)   |  VALU  softmax part 1 of B, tile t
//
// Every slot is 12 (head_dim 96) / 8 (head_dim 64) v_mfma_f32_32x32x16 next to ~65 VALU instructions of the OTHER sub-block -- independent
// registers, one in-order instruction stream, interleaved by sched_group_barrier -- instead of attn3's MFMA phase / VALU phase alternation
// that relied on a second resident wave per SIMD for overlap and got none of it (profiles/r05_pmc_attn.txt: matrix pipe 23 % busy, VALU
// 31 %, waves parked or issue-stalled two thirds of the time; ~1450 cycles per 32-row key tile and SIMD against 768 of matrix-pipe time).
//
// Everything else is attn3's (attn3_kernels.hip): S^T = K Q^T with one query per lane pair, P^T as the B operand of O^T = V^T P^T, V^T by
// ds_read_b64_tr_b16, K / V tiles global -> LDS by LDS-DMA into two-deep rings (K(t+1) and V(t) requested in slot 1 of tile t, one
// vmcnt(0) + one barrier per tile in front of slot 3), source-side XOR swizzle of the K rows, XCD-aware grid, causal query-block pairs,
// deferred rescale (P <= 2^8), query RoPE fused, 128 query rows per workgroup = 2 waves, two workgroups per CU.  The K fragments of a tile
// are read once for both sub-blocks.  No sliding window and no workgroup table here: those launches stay on attn3.
//
// Reference work: the SDPA inside llava.generate (VLN-POL:463) and inside both ViT towers (clip/model.py:178-180; VLN-POL:344, 448).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float16v = __attribute__((ext_vector_type(16))) float;
using v4s = __attribute__((ext_vector_type(4))) short;

constexpr int BKV = 64, NW = 2, BQ = NW * 64;

template <bool BF16>
__device__ __forceinline__ float16v mfma32(const uint4& a, const uint4& b, float16v c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&a), *reinterpret_cast<const half8*>(&b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (BF16) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);      // v_cvt_pk_bf16_f32 (RNE)
        return *reinterpret_cast<const uint32_t*>(&r);
    } else {
        const __half2 h = __floats2half2_rn(lo, hi);
        return *reinterpret_cast<const uint32_t*>(&h);
    }
}

template <bool BF16>
__device__ __forceinline__ float ld16(uint16_t v) {
    if constexpr (BF16) return __uint_as_float((uint32_t)v << 16);
    else return __half2float(*reinterpret_cast<const __half*>(&v));
}
template <bool BF16>
__device__ __forceinline__ uint16_t st16(float f) {
    if constexpr (BF16) {
        const __bf16 r = (__bf16)f;
        return *reinterpret_cast<const uint16_t*>(&r);
    } else {
        __half h = __float2half_rn(f);
        return *reinterpret_cast<uint16_t*>(&h);
    }
}

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// op(v, value of lane ^ 32) through v_permlane32_swap (the builtin: hipcc pads the VALU -> permlane hazard itself)
__device__ __forceinline__ float swap32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(threadIdx.x & 32 ? r[0] : r[1]);
}
__device__ __forceinline__ float pair_max(float v) { return __builtin_fmaxf(v, swap32(v)); }
__device__ __forceinline__ float pair_sum(float v) { return v + swap32(v); }

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// One 1 KiB piece global -> LDS (attn3_kernels.hip dma_piece): inline asm, waited for by hand
__device__ __forceinline__ void dma_piece(uint32_t voff, const void* base, uint32_t dst) {
    const uint32_t d = __builtin_amdgcn_readfirstlane(dst);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(d), "s"(base)
        : "memory");
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// interleave pattern of one slot (one scheduling region, closed by sched_barrier(0)): `nm` MFMAs, each followed by `nd` LDS reads and `nv`
// VALU / transcendental instructions of the other sub-block (masks: 0x008 MFMA, 0x100 DS read, 0x002 VALU, 0x400 TRANS)
#define SLOT_SCHED(nm, nv, nd)                            \
    _Pragma("unroll") for (int i_ = 0; i_ < (nm); ++i_) { \
        SGB(0x008, 1);                                    \
        if ((nd) > 0) SGB(0x100, (nd));                   \
        SGB(0x402, (nv));                                 \
    }

template <bool BF16, int HD, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64, 1)
k_flash_attn_w64(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int64_t row_stride, int64_t batch_stride, int q_off, int k_off,
                 int v_off, float scale_log2e, int seq_len, const int32_t* __restrict__ cu, int n_qblocks, int nx, int B,
                 const float* __restrict__ rope_cos, const float* __restrict__ rope_sin) {
    constexpr int KS = HD / 16;              // 16-deep MFMA steps over head_dim (QK^T)
    constexpr int DB = HD / 32;              // 32-wide head-dim blocks (PV)
    constexpr int CH = HD / 8;               // 16-byte chunks per row
    constexpr int KSLOT = HD == 96 ? 16 : 8;
    constexpr int KST = KSLOT * 8;           // K row stride (elements)
    constexpr int VST = 96;                  // V row stride 192 B
    constexpr int KBUF = BKV * KST, VBUF = BKV * VST;
    constexpr int KPT = KBUF * 2 / 1024, VPT = VBUF * 2 / 1024;     // 1 KiB pieces per tile: K 16 / 8, V 12
    constexpr int KPW = KPT / NW, VPW = VPT / NW;                    // per wave: 8 / 4 and 6
    static_assert(KPT % NW == 0 && VPT % NW == 0, "pieces divide among the waves");
    __shared__ __attribute__((aligned(1024))) uint16_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(1024))) uint16_t Vs[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    // XCD-aware placement (attn3_kernels.hip): the `nx` workgroups of one (sequence, head) are consecutive slots of ONE XCD
    int h, b, xq;
    {
        const int lin = blockIdx.x, G = H * B, G8 = G & ~7;
        if (lin < G8 * nx) {
            const int xcd = lin & 7, slot = lin >> 3, k = slot / nx;
            xq = slot - k * nx;
            if ((H & 7) == 0) {
                const int hp = H >> 3;
                h = xcd * hp + k % hp;
                b = k / hp;
            } else {
                const int g = xcd + 8 * k;
                h = g % H;
                b = g / H;
            }
        } else {
            const int r = lin - G8 * nx, g = G8 + r / nx;
            xq = r % nx;
            h = g % H;
            b = g / H;
        }
    }
    int64_t row0 = (int64_t)b * S;
    const uint16_t* base = qkv + (int64_t)b * batch_stride;
    if (cu) {
        row0 = cu[b];
        S = cu[b + 1] - cu[b];
        seq_len = S;
        n_qblocks = (S + BQ - 1) / BQ;
        base = qkv + row0 * row_stride;
    }
    if (CAUSAL ? xq >= (n_qblocks + 1) / 2 : xq >= n_qblocks) return;
    const uint16_t* Qp = base + (int64_t)(q_off + h) * HD;
    const uint16_t* Kp = base + (int64_t)(k_off + h) * HD;
    const uint16_t* Vp = base + (int64_t)(v_off + h) * HD;

    auto kswz = [](int r) __attribute__((always_inline)) { return HD == 96 ? (r & 15) : ((r >> 1) & 7); };
    constexpr int RPK = 1024 / (KST * 2);                // K rows per piece: 4 / 8
    auto k_src = [&](int i, int& r, int& c) __attribute__((always_inline)) {
        r = (wave + i * NW) * RPK + lane / KSLOT;
        c = (lane % KSLOT) ^ kswz(r);
        c = c < CH ? c : 0;
    };
    auto v_src = [&](int i, int& r, int& c) __attribute__((always_inline)) {
        const int o = (wave + i * NW) * 1024 + lane * 16;
        r = o / (VST * 2);
        c = (o % (VST * 2)) / 16;
        c = c < CH ? c : 0;
    };
    uint32_t koff[KPW], voff[VPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        int r, c;
        k_src(i, r, c);
        koff[i] = (uint32_t)(((int64_t)r * row_stride + c * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < VPW; ++i) {
        int r, c;
        v_src(i, r, c);
        voff[i] = (uint32_t)(((int64_t)r * row_stride + c * 8) * 2);
    }
    const uint32_t lds_k = lds_addr_of(Ks), lds_v = lds_addr_of(Vs);
    const int64_t tile_bytes = (int64_t)BKV * row_stride * 2;
    auto request_k = [&](int T) __attribute__((always_inline)) {
        const char* kb = reinterpret_cast<const char*>(Kp) + (int64_t)T * tile_bytes;
        const int last = S - 1 - T * BKV, buf = T & 1;
        if (last >= BKV - 1) {
#pragma unroll
            for (int i = 0; i < KPW; ++i) dma_piece(koff[i], kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                int r, c;
                k_src(i, r, c);
                dma_piece((uint32_t)(((int64_t)min(r, last) * row_stride + c * 8) * 2), kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
            }
        }
    };
    auto request_v = [&](int T) __attribute__((always_inline)) {
        const char* vb = reinterpret_cast<const char*>(Vp) + (int64_t)T * tile_bytes;
        const int last = S - 1 - T * BKV, buf = T & 1;
        if (last >= BKV - 1) {
#pragma unroll
            for (int i = 0; i < VPW; ++i) dma_piece(voff[i], vb, lds_v + (uint32_t)(buf * VBUF * 2 + (wave + i * NW) * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < VPW; ++i) {
                int r, c;
                v_src(i, r, c);
                dma_piece((uint32_t)(((int64_t)min(r, last) * row_stride + c * 8) * 2), vb, lds_v + (uint32_t)(buf * VBUF * 2 + (wave + i * NW) * 1024));
            }
        }
    };

    using lds_v4s = __attribute__((address_space(3))) v4s;
    const int v_off0 = (hi * 4 + ((lane & 15) >> 2)) * VST + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    auto ld_vf = [&](const uint16_t* Vb_, int s_, int d_) __attribute__((always_inline)) -> uint4 {
        const uint16_t* vb = Vb_ + s_ * 16 * VST + d_ * 32;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)vb);
        const v4s hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vb + 8 * VST));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hv);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };

  for (int pass = 0; pass < (CAUSAL ? 2 : 1); ++pass) {
    const int qb = CAUSAL ? (pass == 0 ? n_qblocks - 1 - xq : xq) : xq;
    if (CAUSAL && pass == 1 && qb == n_qblocks - 1 - xq) break;          // odd count: the middle block stands alone
    const int q0 = qb * BQ, qw = q0 + wave * 64;

    // ---- Q fragments of both sub-blocks (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + hi*8 .. +7]; rotary embedding fused -----------
    uint4 qf[2][KS];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int qrow = qw + sb * 32 + li;
        const int q = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[sb][ks] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * row_stride + ks * 16 + hi * 8);
        if (rope_cos) {      // arithmetic = dense_kernels.hip k_rope (HF apply_rotary_pos_emb on 16-bit tensors): same bits as the separate pass
            constexpr int HALF = HD / 2;
#pragma unroll
            for (int ks = 0; ks < KS / 2; ++ks) {
                const float* cp = rope_cos + (int64_t)q * HALF + ks * 16 + hi * 8;
                const float* sp = rope_sin + (int64_t)q * HALF + ks * 16 + hi * 8;
                const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const uint16_t* ah = reinterpret_cast<const uint16_t*>(&qf[sb][ks]);
                const uint16_t* bh = reinterpret_cast<const uint16_t*>(&qf[sb][ks + KS / 2]);
                uint16_t o1[8], o2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x1 = ld16<BF16>(ah[j]), x2 = ld16<BF16>(bh[j]);
                    o1[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x1 * cc[j])) - ld16<BF16>(st16<BF16>(x2 * ss[j])));
                    o2[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x2 * cc[j])) + ld16<BF16>(st16<BF16>(x1 * ss[j])));
                }
                qf[sb][ks] = *reinterpret_cast<const uint4*>(o1);
                qf[sb][ks + KS / 2] = *reinterpret_cast<const uint4*>(o2);
            }
        }
    }
    float16v oacc[2][DB];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[sb][d][r] = 0.f;
    float m_i[2] = {-INFINITY, -INFINITY}, l_i[2] = {0.f, 0.f};
    float alpha_p[2] = {1.f, 1.f};       // pending rescale of a sub-block's output accumulators (applied between slots, rarely: `resc_p`)
    bool resc_p[2] = {false, false};

    const int kv_len = CAUSAL ? min(seq_len, q0 + BQ) : seq_len;
    const int n_tiles = (kv_len + BKV - 1) / BKV;
    // this wave's last tile with a visible key (causal: the tile of its diagonal), and the first tile that needs masks
    const int t_last = CAUSAL ? min(qw / BKV, n_tiles - 1) : n_tiles - 1;
    const int kmax[2] = {CAUSAL ? min(qw + li, seq_len - 1) : seq_len - 1, CAUSAL ? min(qw + 32 + li, seq_len - 1) : seq_len - 1};

    uint4 fk[2][KS];             // K fragments of the current tile (both key blocks), shared by the two sub-blocks
    uint4 fv[4][DB];             // V^T fragments of ONE tile: read once (slot 3 of tile t), used by O_A(t) in slot 4 and -- still in registers --
                                 // by O_B(t) in slot 3 of tile t + 1, where each fragment is replaced by tile t + 1's right behind its last MFMA
    float16v st[2][2];           // S^T tiles [sub-block][key block]
    uint4 pf[2][4];              // P^T fragments [sub-block][MFMA step]

    auto load_k = [&](int T) __attribute__((always_inline)) {
        const uint16_t* Kb = Ks + (T & 1) * KBUF;
        const uint16_t* Ka = Kb + li * KST;
        const uint16_t* Kc = Kb + (32 + li) * KST;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            fk[0][ks] = *reinterpret_cast<const uint4*>(Ka + (((ks * 2 + hi) ^ kswz(li)) << 3));
            fk[1][ks] = *reinterpret_cast<const uint4*>(Kc + (((ks * 2 + hi) ^ kswz(li)) << 3));
        }
    };
    auto load_v = [&](int T) __attribute__((always_inline)) {
        const uint16_t* Vb = Vs + (T & 1) * VBUF + v_off0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) fv[s][d] = ld_vf(Vb, s, d);
    };
    auto qk = [&](int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[sb][0][r] = 0.f; st[sb][1][r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            st[sb][0] = mfma32<BF16>(fk[0][ks], qf[sb][ks], st[sb][0]);
            st[sb][1] = mfma32<BF16>(fk[1][ks], qf[sb][ks], st[sb][1]);
        }
    };
    auto pv = [&](int sb) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) oacc[sb][d] = mfma32<BF16>(fv[s][d], pf[sb][s], oacc[sb][d]);
    };
    // O_B += V(T-1)^T P_B(T-1) out of the fragments in registers, each replaced by tile T's as soon as its MFMA has been issued
    auto pv_b_reload = [&](int T) __attribute__((always_inline)) {
        const uint16_t* Vb = Vs + (T & 1) * VBUF + v_off0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) {
                oacc[1][d] = mfma32<BF16>(fv[s][d], pf[1][s], oacc[1][d]);
                fv[s][d] = ld_vf(Vb, s, d);
            }
    };
    // softmax part 1 of sub-block sb on tile T: [masks], row max over both key blocks, deferred-rescale vote, exp2 of key block 0
    auto sm1 = [&](int sb, int T, bool masked) __attribute__((always_inline)) {
        if (masked) {
            const int hi_ = kmax[sb] - T * BKV - hi * 4;                 // key - key0 - 4 hi <= hi_ is visible
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o_ = j * 8 + r;
                    if (o_ > hi_) st[sb][0][4 * j + r] = -INFINITY;
                    if (o_ + 32 > hi_) st[sb][1][4 * j + r] = -INFINITY;
                }
        }
        float tmax = max3(st[sb][0][0], st[sb][0][1], st[sb][0][2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) tmax = max3(tmax, st[sb][0][r], st[sb][0][r + 1]);
        tmax = fmaxf(tmax, st[sb][0][15]);
#pragma unroll
        for (int r = 0; r + 1 < 16; r += 2) tmax = max3(tmax, st[sb][1][r], st[sb][1][r + 1]);
        tmax = pair_max(tmax);
        const float tm = tmax * scale_log2e;
        const bool keep = __all(tm <= m_i[sb] + 8.0f);                  // deferred rescale: P stays <= 2^8
        const float m_new = keep ? m_i[sb] : fmaxf(m_i[sb], tm);
        // Branch-free inside the slot (a branch would end the scheduling region): the row sum takes alpha now (1 when kept), the 48 / 32
        // output accumulators take it in `rescale(sb)` between two slots, under a wave-uniform branch that is rarely taken.
        const float alpha = __builtin_amdgcn_exp2f(m_i[sb] - m_new);         // (m_i = -inf on the first tile: alpha = 0, on zeros)
        l_i[sb] *= alpha;
        alpha_p[sb] = alpha;
        resc_p[sb] = !keep;
        m_i[sb] = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[sb][0][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[sb][0][r], scale_log2e, -m_new));
    };
    auto rescale = [&](int sb) __attribute__((always_inline)) {
        if (resc_p[sb]) {
            asm volatile("; rescale (rare)" ::: "memory");           // keeps the block a real branch: if-converted, the multiplies (and the
            const float a = alpha_p[sb];                             // accumulators' round trip out of the AGPRs) would run on every tile
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[sb][d][r] *= a;
        }
    };
    // softmax part 2: exp2 of key block 1, row sum (pairwise tree, float32, before P is rounded), P^T fragments
    auto sm2 = [&](int sb) __attribute__((always_inline)) {
        const float mu = m_i[sb];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[sb][1][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[sb][1][r], scale_log2e, -mu));
        float ts = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) ts += (st[sb][0][r] + st[sb][0][r + 1]) + (st[sb][0][r + 2] + st[sb][0][r + 3]);
#pragma unroll
        for (int r = 0; r < 16; r += 4) ts += (st[sb][1][r] + st[sb][1][r + 1]) + (st[sb][1][r + 2] + st[sb][1][r + 3]);
        l_i[sb] += ts;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            pf[sb][s].x = pack2<BF16>(st[sb][0][8 * s + 0], st[sb][0][8 * s + 1]);
            pf[sb][s].y = pack2<BF16>(st[sb][0][8 * s + 2], st[sb][0][8 * s + 3]);
            pf[sb][s].z = pack2<BF16>(st[sb][0][8 * s + 4], st[sb][0][8 * s + 5]);
            pf[sb][s].w = pack2<BF16>(st[sb][0][8 * s + 6], st[sb][0][8 * s + 7]);
            pf[sb][2 + s].x = pack2<BF16>(st[sb][1][8 * s + 0], st[sb][1][8 * s + 1]);
            pf[sb][2 + s].y = pack2<BF16>(st[sb][1][8 * s + 2], st[sb][1][8 * s + 3]);
            pf[sb][2 + s].z = pack2<BF16>(st[sb][1][8 * s + 4], st[sb][1][8 * s + 5]);
            pf[sb][2 + s].w = pack2<BF16>(st[sb][1][8 * s + 6], st[sb][1][8 * s + 7]);
        }
    };

    // ---- prologue: K(0) lands, its fragments are read ------------------------------------------------------------------------------------
    request_k(0);                      // (every wave is past the previous pass's last barrier: both rings are free)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_k(0);

    constexpr int NM = 2 * KS;                                   // MFMAs per slot: 12 / 8
    constexpr int NV1 = (84 + NM - 1) / NM, ND3 = 8 * DB / NM;   // slots 1 / 3: ~84 VALU (softmax part 2); slot 3: 8 DB transposing V reads
    constexpr int NV2 = (60 + NM - 1) / NM, ND4 = 2 * KS / NM;   // slots 2 / 4: ~60 VALU (softmax part 1); slot 4: 2 KS K-fragment reads
    // One tile.  FIRST: no tile t-1 work of sub-block B yet.  MASKED: the masks of this wave's diagonal / the sequence's last, partial tile.
    // `on`: this wave still has visible keys in tile t (wave-uniform); a wave that is done keeps requesting its pieces and meeting the barrier.
#define FA_ITER(FIRST, MASKED)                                                                                                                 \
    {                                                                                                                                          \
        const bool on = t <= t_last, drain = t == t_last + 1;                                                                                 \
        if (t + 1 < n_tiles) request_k(t + 1);                                                                                                 \
        request_v(t);                                                                                                                          \
        if (on) {                                                                                                                              \
            /* slot 1: S_A(t)  ||  softmax part 2 of B(t-1) */                                                                                \
            if (!(FIRST)) rescale(1);                                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                                                 \
            qk(0);                                                                                                                             \
            if (!(FIRST)) { sm2(1); SLOT_SCHED(NM, NV1, 0) }                                                                                   \
            __builtin_amdgcn_sched_barrier(0);                                                                                                 \
            /* slot 2: S_B(t)  ||  softmax part 1 of A(t) */                                                                                  \
            qk(1);                                                                                                                             \
            sm1(0, t, MASKED);                                                                                                                 \
            if (!(MASKED)) { SLOT_SCHED(NM, NV2, 0) }                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                                                 \
            rescale(0);                                                                                                                        \
        } else if (drain && !(FIRST)) {                                                                                                        \
            rescale(1);                                                                                                                        \
            sm2(1);                                                                                                                            \
        }                                                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* this wave's pieces of K(t+1), V(t) have landed */                           \
        __syncthreads();                                       /* ... and everybody's; every wave is done reading K(t-1)'s / V(t-2)'s buffers */ \
        if (on) {                                                                                                                              \
            /* slot 3: O_B += V(t-1)^T P_B(t-1), V(t) fragments behind them  ||  softmax part 2 of A(t) */                                    \
            if (FIRST) load_v(t); else pv_b_reload(t);                                                                                         \
            sm2(0);                                                                                                                            \
            if (!(FIRST)) { SLOT_SCHED(NM, NV1, ND3) }                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                                 \
            /* slot 4: O_A += V(t)^T P_A(t)  ||  softmax part 1 of B(t), K(t+1) fragments */                                                  \
            pv(0);                                                                                                                             \
            sm1(1, t, MASKED);                                                                                                                 \
            if (t + 1 < n_tiles) load_k(t + 1);                                                                                                \
            if (!(MASKED)) { SLOT_SCHED(NM, NV2, ND4) }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                                                 \
        } else if (drain && !(FIRST)) {                                                                                                        \
            pv(1);                                                                                                                             \
        }                                                                                                                                      \
    }

    // tiles below t_mask are visible in full to every query of this wave
    const int t_mask = min(CAUSAL ? (qw + 1) / BKV : n_tiles, min(S / BKV, n_tiles));
    int t = 0;
    if (t_mask > 0) FA_ITER(true, false) else FA_ITER(true, true)
    for (t = 1; t < min(t_mask, n_tiles); ++t) FA_ITER(false, false)
    for (; t < n_tiles; ++t) FA_ITER(false, true)
#undef FA_ITER
    // drain of the wave that computed the last tile: softmax part 2 and P V of sub-block B on it (its V fragments are still in registers)
    if (t_last == n_tiles - 1) {
        rescale(1);
        sm2(1);
        pv(1);
    }

    // ---- epilogue: lane holds O[qrow][32d + 8j + 4hi + r]; lane pairs exchange so that each stores 16 contiguous bytes ------------------------
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const float l = pair_sum(l_i[sb]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qrow = qw + sb * 32 + li;
        uint16_t* op = out + ((row0 + qrow) * H + h) * HD;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                uint32_t a0 = pack2<BF16>(oacc[sb][d][8 * jp + 0] * inv, oacc[sb][d][8 * jp + 1] * inv), a1 = pack2<BF16>(oacc[sb][d][8 * jp + 2] * inv, oacc[sb][d][8 * jp + 3] * inv);
                uint32_t b0 = pack2<BF16>(oacc[sb][d][8 * jp + 4] * inv, oacc[sb][d][8 * jp + 5] * inv), b1 = pack2<BF16>(oacc[sb][d][8 * jp + 6] * inv, oacc[sb][d][8 * jp + 7] * inv);
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                if (qrow < S) *reinterpret_cast<uint4*>(op + d * 32 + jp * 16 + hi * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
    __syncthreads();                   // the next pass's prologue overwrites ring buffers this pass's last reads came from
  }   // pass
}

}  // namespace

extern "C" {

// Same contract as d3d_flash_attention_v3_rope_q without a window: dense or packed, causal or not, head_dim 64 / 96, query RoPE optional.
int32_t d3d_flash_attention_v4(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                               int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens,
                               const float* rope_cos, const float* rope_sin, int32_t dtype, void* stream) {
    if (B <= 0 || S <= 0) return D3D_OK;
    if ((head_dim != 64 && head_dim != 96) || (row_stride & 7) || (batch_stride & 7)) {
        d3d_set_error_("d3d_flash_attention_v4: head_dim must be 64 or 96; strides multiples of 8 elements");
        return D3D_EINVAL;
    }
    if ((int64_t)BKV * row_stride * 2 >= (1ll << 31)) {
        d3d_set_error_("d3d_flash_attention_v4: a 64-row tile of the QKV buffer must span less than 2 GiB (32-bit per-lane offsets)");
        return D3D_EINVAL;
    }
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) {
        d3d_set_error_("d3d_flash_attention_v4: rope_cos and rope_sin come together");
        return D3D_EINVAL;
    }
    const float sl2 = 1.4426950408889634f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    const int nqb = (S + BQ - 1) / BQ;
    const int nx = causal ? (nqb + 1) / 2 : nqb;
    dim3 grid((unsigned)((int64_t)nx * H * B)), block(NW * 64);
#define D3D_FA4(BF, HDV, CA) hipLaunchKernelGGL((k_flash_attn_w64<BF, HDV, CA>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, \
                                                sl2, seq_len, cu_seqlens, nqb, nx, B, rope_cos, rope_sin)
    if (dtype == 0) {
        if (head_dim == 96) { if (causal) D3D_FA4(true, 96, true); else D3D_FA4(true, 96, false); }
        else { if (causal) D3D_FA4(true, 64, true); else D3D_FA4(true, 64, false); }
    } else {
        if (head_dim == 96) { if (causal) D3D_FA4(false, 96, true); else D3D_FA4(false, 96, false); }
        else { if (causal) D3D_FA4(false, 64, true); else D3D_FA4(false, 64, false); }
    }
#undef D3D_FA4
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
