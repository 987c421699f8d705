// attn4_kernels.hip -- fourth flash-attention forward for gfx950 (round 6): attn3's data path (LDS-DMA K / V tiles, S^T = K Q^T with one
// query per lane pair, P^T as the B operand of O^T = V^T P^T, ds_read_b64_tr_b16 for V^T, XCD-aware grid, causal query-block pairs, query
// RoPE fused, deferred rescale) with the tile body SOFTWARE-PIPELINED INSIDE THE WAVE at the granularity of one 32-key block:
//
//     slot b :   MFMA   O += V(b-1)^T P(b-1)   and   S(b+1) = K(b+1) Q^T           (6 + 6 v_mfma_f32_32x32x16 at head_dim 96, 4 + 4 at 64)
//                VALU   online softmax of block b  (row max, rescale vote, exp2, row sum, P -> 16 bit: ~75 instructions)
//
// One in-order instruction stream in which the matrix instructions of blocks b-1 / b+1 sit between the vector instructions of block b
// (sched_group_barrier); the S and P registers are double-buffered by block parity (the slot loop is unrolled by two), the K / V fragments
// of a slot are read at its top and land under the row-max chain.  attn3 ran a tile as MFMA phase / VALU phase / MFMA phase and counted on
// the second resident wave of the SIMD to fill the other pipe -- profiles/r05_pmc_attn.txt: matrix pipe 23 % busy, VALU 31 %, waves parked
// or issue-stalled two thirds of the time (~1450 cycles per wave and 64-key tile against 768 of matrix-pipe time).  A first attempt at the
// textbook fix -- one wave per SIMD with the whole 512-register file and two 32-row sub-blocks pipelined against each other -- is kept under
// tools/experiments/attn4_single_wave/: hipcc spills it (~1 000 registers) and shuttles the accumulators between AGPRs and VGPRs on every
// tile; that structure needs an assembly-owned register file.  This one stays within 256 architectural VGPRs at two waves per SIMD.
//
// Block = 32 keys = half a 64-key tile; the rings stay tile-sized.  Slot b reads the V fragments of block b-1 and the K fragments of block
// b+1, so K runs one tile ahead of V: after the barrier that closes slot 2t the workgroup requests K tile t+2 and V tile t+1, both waited
// for in front of the barrier that closes slot 2t+2 (one vmcnt(0) + one barrier per 64 keys, as in attn3).  Online softmax per 32-key block
// (attn3: per 64): rescale votes and row sums are taken at a finer grain -- results differ from attn3 in the last bits, not in accuracy.
// No sliding window and no workgroup table here: those launches stay on attn3.
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

constexpr int BKV = 64, NW = 4, BQ = NW * 32;

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

// the value of lane ^ 32 through v_permlane32_swap (the builtin: hipcc pads the VALU -> permlane hazard itself)
__device__ __forceinline__ float swap32(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(threadIdx.x & 32 ? r[0] : r[1]);
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// One 1 KiB piece global -> LDS (attn3_kernels.hip dma_piece): inline asm (hipcc would drain a builtin LDS-DMA in front of the next
// ds_read of ANY buffer), waited for by hand
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
// Timing ablations (tools/experiments/attn_ablate.py builds one library per value; results are WRONG for every value but 0):
//   1 no softmax arithmetic   2 no MFMAs   3 no tile requests after the prologue (barriers stay)   4 no requests and no barriers
//   5 fragment reads only in the prologue   6 no key tiles at all (prologue + epilogue only)
#ifndef D3D_ATTN_ABL
#define D3D_ATTN_ABL 0
#endif
constexpr int ABL = D3D_ATTN_ABL;

template <bool BF16, int HD, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64, 2)
k_flash_attn_pipe(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int64_t row_stride, int64_t batch_stride, int q_off, int k_off,
                  int v_off, float scale_log2e, int seq_len, const int32_t* __restrict__ cu, int n_qblocks, int nx, int B,
                  const float* __restrict__ rope_cos, const float* __restrict__ rope_sin) {
    constexpr int KS = HD / 16;              // 16-deep MFMA steps over head_dim (QK^T)
    constexpr int DB = HD / 32;              // 32-wide head-dim blocks (PV)
    constexpr int CH = HD / 8;               // 16-byte chunks per row
    constexpr int KSLOT = HD == 96 ? 16 : 8; // chunk slots per K row in LDS (power of two: XOR swizzle)
    constexpr int KST = KSLOT * 8;           // K row stride (elements): 256 B / 128 B
    constexpr int VST = 96;                  // V row stride 192 B (attn2_kernels.hip: the transposing read's bank spread)
    constexpr int KBUF = BKV * KST, VBUF = BKV * VST;
    constexpr int KPT = KBUF * 2 / 1024, VPT = VBUF * 2 / 1024;      // 1 KiB pieces per tile: K 16 / 8, V 12
    constexpr int KPW = KPT / NW, VPW = VPT / NW;                    // per wave: 4 / 2 and 3
    static_assert(KPT % NW == 0 && VPT % NW == 0, "pieces divide among the waves");
    __shared__ __attribute__((aligned(1024))) uint16_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(1024))) uint16_t Vs[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    // XCD-aware placement (attn3_kernels.hip): the `nx` workgroups of one (sequence, head) are consecutive slots of ONE XCD
    int h, b_, xq;
    {
        const int lin = blockIdx.x, G = H * B, G8 = G & ~7;
        if (lin < G8 * nx) {
            const int xcd = lin & 7, slot = lin >> 3, k = slot / nx;
            xq = slot - k * nx;
            if ((H & 7) == 0) {
                const int hp = H >> 3;
                h = xcd * hp + k % hp;
                b_ = k / hp;
            } else {
                const int g = xcd + 8 * k;
                h = g % H;
                b_ = g / H;
            }
        } else {
            const int r = lin - G8 * nx, g = G8 + r / nx;
            xq = r % nx;
            h = g % H;
            b_ = g / H;
        }
    }
    int64_t row0 = (int64_t)b_ * S;
    const uint16_t* base = qkv + (int64_t)b_ * batch_stride;
    if (cu) {
        row0 = cu[b_];
        S = cu[b_ + 1] - cu[b_];
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
        if ((ABL == 3 || ABL == 4) && T > 1) return;
        const char* kb = reinterpret_cast<const char*>(Kp) + (int64_t)T * tile_bytes;
        const int last = S - 1 - T * BKV, buf = T & 1;                  // last valid row of the tile (>= 0)
        if (last >= BKV - 1) {
#pragma unroll
            for (int i = 0; i < KPW; ++i) dma_piece(koff[i], kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
        } else {                                                         // a sequence's last, partial tile: rows clamped to its last row
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                int r, c;
                k_src(i, r, c);
                dma_piece((uint32_t)(((int64_t)min(r, last) * row_stride + c * 8) * 2), kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
            }
        }
    };
    auto request_v = [&](int T) __attribute__((always_inline)) {
        if ((ABL == 3 || ABL == 4) && T > 0) return;
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
    // V fragment base (A operand of O^T = V^T P^T through the transposing read): lane addresses key 16s + 8jj + 4hi + (l & 15) / 4,
    // dims 32d + 16 ((l >> 4) & 1) + 4 (l & 3) and receives dim 32d + li of 4 consecutive keys
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
    const int q0 = qb * BQ, qw = q0 + wave * 32;
    const int qrow = qw + li;                                            // this lane's query
    const int kv_len = CAUSAL ? min(seq_len, q0 + BQ) : seq_len;
    const int n_tiles = (kv_len + BKV - 1) / BKV;
    // K tiles 0 and 1 and V tile 0 are requested FIRST: they fly under the query loads and the rotary arithmetic (every wave is past the
    // previous pass's last barrier: the rings are free)
    request_k(0);
    if (n_tiles > 1) request_k(1);
    request_v(0);

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + hi*8 .. +7]; rotary embedding fused (attn3_kernels.hip) -------
    uint4 qf[KS];
    {
        const int q = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * row_stride + ks * 16 + hi * 8);
        if (rope_cos) {
            constexpr int HALF = HD / 2;
#pragma unroll
            for (int ks = 0; ks < KS / 2; ++ks) {
                const float* cp = rope_cos + (int64_t)q * HALF + ks * 16 + hi * 8;
                const float* sp = rope_sin + (int64_t)q * HALF + ks * 16 + hi * 8;
                const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
                const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
                const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const uint16_t* ah = reinterpret_cast<const uint16_t*>(&qf[ks]);
                const uint16_t* bh = reinterpret_cast<const uint16_t*>(&qf[ks + KS / 2]);
                uint16_t o1[8], o2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x1 = ld16<BF16>(ah[j]), x2 = ld16<BF16>(bh[j]);
                    o1[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x1 * cc[j])) - ld16<BF16>(st16<BF16>(x2 * ss[j])));
                    o2[j] = st16<BF16>(ld16<BF16>(st16<BF16>(x2 * cc[j])) + ld16<BF16>(st16<BF16>(x1 * ss[j])));
                }
                qf[ks] = *reinterpret_cast<const uint4*>(o1);
                qf[ks + KS / 2] = *reinterpret_cast<const uint4*>(o2);
            }
        }
    }
    float16v oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    float alpha_p = 1.f;               // rescale of the output accumulators decided by the last softmax, applied at the top of the next slot
    bool resc_p = false;

    // ---- block ranges (32-key blocks) --------------------------------------------------------------------------------------------------------
    const int nb_seq = (kv_len + 31) / 32;
    const int nbw = CAUSAL ? min(qw / 32 + 1, nb_seq) : nb_seq;                          // this wave's blocks: [0, nbw)
    const int nb_max = CAUSAL ? min((q0 + BQ - 32) / 32 + 1, nb_seq) : nb_seq;           // the workgroup's (its last wave's)
    const int b_full = min(CAUSAL ? qw / 32 : nb_seq, seq_len / 32);                     // blocks below are visible in full to all 32 queries
    const int kmax = CAUSAL ? min(qrow, seq_len - 1) : seq_len - 1;

    float16v sv[2];                    // S^T of the blocks of even / odd index
    uint4 pf[2][2];                    // P^T fragments [block parity][MFMA step]
    uint4 kf[KS];                      // K fragments of block b + 1
    uint4 vf[2][DB];                   // V^T fragments of block b - 1

    auto load_k = [&](int bk) __attribute__((always_inline)) {       // K fragments of block bk
        if (ABL == 5 && bk > 0) return;
        const uint16_t* Ka = Ks + ((bk >> 1) & 1) * KBUF + ((bk & 1) * 32 + li) * KST;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(Ka + (((ks * 2 + hi) ^ kswz(li)) << 3));
    };
    auto load_v = [&](int bv) __attribute__((always_inline)) {       // V^T fragments of block bv
        if (ABL == 5 && bv > 0) return;
        const uint16_t* Vb = Vs + ((bv >> 1) & 1) * VBUF + v_off0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) vf[s][d] = ld_vf(Vb, (bv & 1) * 2 + s, d);
    };
    auto qk = [&](int par) __attribute__((always_inline)) {          // S(par) = K fragments x Q
        if (ABL == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sv[par][r] = __uint_as_float(kf[r % KS].x ^ qf[r % KS].y) * 1e-30f;
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[par][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sv[par] = mfma32<BF16>(kf[ks], qf[ks], sv[par]);
    };
    auto pv = [&](int par) __attribute__((always_inline)) {          // O += V fragments x P(par)
        if (ABL == 2) {
            oacc[0][0] += __uint_as_float(vf[0][0].x ^ vf[1][DB - 1].w ^ pf[par][0].x ^ pf[par][1].w) * 1e-30f;
            return;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int d = 0; d < DB; ++d) oacc[d] = mfma32<BF16>(vf[s][d], pf[par][s], oacc[d]);
    };
    // both of a MAIN slot, alternating: two MFMAs on the same accumulator are never neighbours in the stream (a vector instruction between
    // two MFMAs of one accumulation chain costs ~40 cycles, MI355X_MICROARCH.md; between different accumulators ~6)
    auto pv_qk = [&](int par) __attribute__((always_inline)) {
        if (ABL == 2) { pv(par); qk(par); return; }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[par][r] = 0.f;
        constexpr int NP = 2 * DB;
#pragma unroll
        for (int i = 0; i < (NP > KS ? NP : KS); ++i) {
            if (i < NP) oacc[i % DB] = mfma32<BF16>(vf[i / DB][i % DB], pf[par][i / DB], oacc[i % DB]);
            if (i < KS) sv[par] = mfma32<BF16>(kf[i], qf[i], sv[par]);
        }
    };
    auto rescale = [&]() __attribute__((always_inline)) {
        if (resc_p) {
            asm volatile("; rescale (rare)" ::: "memory");            // keeps the block a real branch (if-converted, the 48 multiplies would run
            const float a = alpha_p;                                  // in every slot)
#pragma unroll
            for (int d = 0; d < DB; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= a;
            resc_p = false;
        }
    };
    // online softmax of block bb (registers sv[par] -> pf[par]), base 2, one query per lane pair, branch-free
    auto softmax = [&](int par, int bb, bool masked) __attribute__((always_inline)) {
        float16v& s = sv[par];
        if (ABL == 1) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                pf[par][st].x = pack2<BF16>(s[8 * st + 0], s[8 * st + 1]);
                pf[par][st].y = pack2<BF16>(s[8 * st + 2], s[8 * st + 3]);
                pf[par][st].z = pack2<BF16>(s[8 * st + 4], s[8 * st + 5]);
                pf[par][st].w = pack2<BF16>(s[8 * st + 6], s[8 * st + 7]);
            }
            l_i += 1.f;
            return;
        }
        if (masked) {
            const int hi_ = kmax - bb * 32 - hi * 4;                   // key - key0 - 4 hi <= hi_ is visible
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j * 8 + r > hi_) s[4 * j + r] = -INFINITY;
        }
        float tmax = max3(s[0], s[1], s[2]);
#pragma unroll
        for (int r = 3; r + 1 < 16; r += 2) tmax = max3(tmax, s[r], s[r + 1]);
        tmax = fmaxf(tmax, s[15]);
        tmax = fmaxf(tmax, swap32(tmax));
        const float tm = tmax * scale_log2e;
        const bool keep = __all(tm <= m_i + 8.0f);                     // deferred rescale: P stays <= 2^8
        const float m_new = keep ? m_i : fmaxf(m_i, tm);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;          // no visible key yet (a masked block of a short sequence): exp2(-inf) = 0
        const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);       // 1 when kept; 0 on the first block (m_i = -inf, accumulators zero)
        l_i *= alpha;
        alpha_p = alpha;
        resc_p = !keep;
        m_i = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, -m_use));
        float ts = 0.f;                                                // row sum: pairwise tree, float32, before P is rounded
#pragma unroll
        for (int r = 0; r < 16; r += 4) ts += (s[r] + s[r + 1]) + (s[r + 2] + s[r + 3]);
        l_i += ts;
#pragma unroll
        for (int st = 0; st < 2; ++st) {   // P^T fragments: MFMA step st covers keys 16 st .. 16 st + 15 of the block (attn3's k-slot order)
            pf[par][st].x = pack2<BF16>(s[8 * st + 0], s[8 * st + 1]);
            pf[par][st].y = pack2<BF16>(s[8 * st + 2], s[8 * st + 3]);
            pf[par][st].z = pack2<BF16>(s[8 * st + 4], s[8 * st + 5]);
            pf[par][st].w = pack2<BF16>(s[8 * st + 6], s[8 * st + 7]);
        }
    };

    // ---- prologue: K tiles 0 and 1 and V tile 0 have landed; S(0) ---------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_k(0);
    qk(0);

    constexpr int NM = KS + 2 * DB;    // MFMAs of a full slot: 12 / 8
    // One slot, PAR = parity of b (compile time).  MAIN: every part present and no masks -- ONE scheduling region: the fragment reads first,
    // the row-max chain of block b under their latency, then one MFMA per ~5 vector instructions.  GEN: the edges (first / diagonal / last
    // blocks, waves that are done), each part under its wave-uniform condition, compiler-scheduled.
#define FA_SLOT_MAIN(PAR)                                                                                                                      \
    {                                                                                                                                          \
        rescale();                                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                                     \
        load_v(b - 1);                                                                                                                         \
        load_k(b + 1);                                                                                                                         \
        pv_qk((PAR) ^ 1);                                                                                                                      \
        softmax(PAR, b, false);                                                                                                                \
        /* the block's P fragments and row sum are "used" here: keeps LLVM from sinking the exps below the tile barrier, next to their MFMAs */ \
        asm volatile("" ::"v"(pf[PAR][0].x), "v"(pf[PAR][0].y), "v"(pf[PAR][0].z), "v"(pf[PAR][0].w), "v"(pf[PAR][1].x), "v"(pf[PAR][1].y),   \
                     "v"(pf[PAR][1].z), "v"(pf[PAR][1].w), "v"(l_i));                                                                          \
        SGB(0x100, 4 * DB + KS);                  /* all fragment reads */                                                                     \
        SGB(0x402, 16);                           /* row max, pair exchange, vote */                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 2 * DB; ++i_) {   /* the P V MFMAs (rotating accumulators), vector work between them */       \
            SGB(0x008, 1);                                                                                                                     \
            SGB(0x402, HD == 96 ? 9 : 14);                                                                                                     \
        }                                                                                                                                      \
        SGB(0x008, KS);                           /* the S chain (ONE accumulator) back to back: a vector instruction between two MFMAs of  */ \
        SGB(0x402, 8);                            /* one accumulation chain costs ~40 cycles (MI355X_MICROARCH.md), between different ones ~6 */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                                     \
    }
#define FA_SLOT_GEN(PAR)                                                                                                                       \
    {                                                                                                                                          \
        const bool do_pv = b >= 1 && b <= nbw, do_sm = b < nbw, do_qk = b + 1 < nbw;                                                          \
        rescale();                                                                                                                             \
        if (do_pv) { load_v(b - 1); }                                                                                                          \
        if (do_qk) { load_k(b + 1); }                                                                                                          \
        if (do_pv) pv((PAR) ^ 1);                                                                                                              \
        if (do_qk) qk((PAR) ^ 1);                                                                                                              \
        if (do_sm) softmax(PAR, b, b >= b_full);                                                                                               \
    }
    // the tile's barrier: this wave's pieces of K tile t+1 and V tile t (requested a tile ago) have landed -- then everybody's; and every wave
    // is done reading K tile t's and V tile t-1's buffers, which the next requests overwrite
#define FA_SYNC()                                                                                                                              \
    {                                                                                                                                          \
        if (ABL != 4) {                                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                                   \
            __syncthreads();                                                                                                                   \
        }                                                                                                                                      \
        if (t + 2 < n_tiles) request_k(t + 2);                                                                                                 \
        if (t + 1 < n_tiles) request_v(t + 1);                                                                                                 \
    }

    // Slots 0 .. nb_max in pairs (even, odd), one barrier per pair = per 64-key tile.  Every wave runs the same number of pairs; which form a
    // pair takes is the wave's own business: pairs [1, t_main) have both slots in the MAIN form (1 <= b, b + 1 < nbw, b < b_full).
    if (ABL != 6) {
    const int b_hi = min(nbw - 2, b_full - 1);                 // last slot that can take the MAIN form
    const int t_main = (b_hi + 1) / 2;                         // pairs t < t_main: 2t + 1 <= b_hi
    int t = 0;
    {
        int b = 0;
        FA_SLOT_GEN(0)
        FA_SYNC()
        b = 1;
        if (b <= nb_max) FA_SLOT_GEN(1)
    }
    for (t = 1; t < t_main; ++t) {
        int b = 2 * t;
        FA_SLOT_MAIN(0)
        FA_SYNC()
        b = 2 * t + 1;
        FA_SLOT_MAIN(1)
    }
    for (; 2 * t <= nb_max; ++t) {
        int b = 2 * t;
        FA_SLOT_GEN(0)
        FA_SYNC()
        b = 2 * t + 1;
        if (b <= nb_max) FA_SLOT_GEN(1)
    }
    }
#undef FA_SLOT_MAIN
#undef FA_SLOT_GEN
#undef FA_SYNC
    rescale();                         // (a pending decision of the last softmax has no P V behind it for this wave only if nbw == 0: harmless)

    // ---- epilogue: lane holds O[qrow][32d + 8j + 4hi + r]; lane pairs exchange so that each stores 16 contiguous bytes ------------------------
    {
        const float l = l_i + swap32(l_i);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        uint16_t* op = out + ((row0 + qrow) * H + h) * HD;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                uint32_t a0 = pack2<BF16>(oacc[d][8 * jp + 0] * inv, oacc[d][8 * jp + 1] * inv), a1 = pack2<BF16>(oacc[d][8 * jp + 2] * inv, oacc[d][8 * jp + 3] * inv);
                uint32_t b0 = pack2<BF16>(oacc[d][8 * jp + 4] * inv, oacc[d][8 * jp + 5] * inv), b1 = pack2<BF16>(oacc[d][8 * jp + 6] * inv, oacc[d][8 * jp + 7] * inv);
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                if (qrow < S) *reinterpret_cast<uint4*>(op + d * 32 + jp * 16 + hi * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
        }
    }
    __syncthreads();                   // the next pass's prologue overwrites ring buffers this pass's last slots read
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
#define D3D_FA4(BF, HDV, CA) hipLaunchKernelGGL((k_flash_attn_pipe<BF, HDV, CA>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, \
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
