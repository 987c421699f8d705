// attn3_kernels.hip -- third flash-attention forward for gfx950: the arithmetic and data layout of attn2_kernels.hip (v_mfma_f32_32x32x16,
// S^T = K Q^T with one query per lane pair, P^T as the B operand of O^T = V^T P^T, ds_read_b64_tr_b16 for V^T, XCD-aware grid, causal
// pairs, sliding window), with the two things a wave WAITED for removed from its dependent chain:
//
//   * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR round trip, no ds_write): tile
//     t + 1 is requested at the top of tile t and only waited for in front of the barrier that publishes it.  v2 staged through 24 VGPRs
//     (6 global loads + 6 ds_write_b128 per wave and tile, a vmcnt wait in the middle of the tile, 12 64-bit address computations).
//   * fragment reads are issued a PHASE ahead, in bulk: all 12 K fragments of a tile before its first QK^T MFMA (one exposed LDS latency
//     per tile; v2 requested each fragment two MFMA slots ahead and waited 12 times), and all V^T fragments right behind the last QK^T
//     MFMA, so that they land under the softmax and the PV MFMAs never wait.  The K and V fragment registers are the same 48 VGPRs
//     (disjoint lifetimes), paid for by the 24 staging registers.
//
// Why: two waves per SIMD overlap MFMA and VALU work almost perfectly when neither is waiting
// (tools/probe/mfma_valu_probe.hip), yet v2 gained only 1.4x from its second resident workgroup and ran a wave-tile in ~2 700 cycles per
// SIMD against ~900 of matrix pipe and ~900 of VALU: 29 s_waitcnt per tile, most of them an LDS round trip in front of a single MFMA.
//
// LDS image = lane-linear 1 KiB pieces.  K rows keep v2's layout (row stride 256 B / 128 B, chunk c of row r in slot c ^ f(r)); the swizzle
// is applied to the per-lane SOURCE chunk.  V rows are 192 B apart (64-byte pad for head_dim 64): lanes that map to pad bytes re-load
// chunk 0 of their row.  Rows beyond the sequence are clamped to its last row (finite values, masked scores) in the per-lane offsets of the
// one tile that needs it.
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

constexpr int BKV = 64, NW4 = 4, BQ4 = NW4 * 32;     // (NW: a template parameter of the kernel; 4 waves = 128 query rows is the default)

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
        const __bf16 r = (__bf16)f;                  // hardware converter: round to nearest even, NaN-safe
        return *reinterpret_cast<const uint16_t*>(&r);
    } else {
        __half h = __float2half_rn(f);
        return *reinterpret_cast<uint16_t*>(&h);
    }
}

// (builtins, not inline asm: an asm statement hides its VALU reads from the hazard recogniser -- see attn2_kernels.hip)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// op(v, value of lane ^ 32) through v_permlane32_swap (both operands the same value: the xor-32 butterfly); the operand is a VALU result
__device__ __forceinline__ float pair_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}
__device__ __forceinline__ float pair_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// One 1 KiB piece global -> LDS: lane l's 16 bytes at base + voff land at LDS byte dst + 16 l.  Inline asm on purpose (hipcc would count
// a builtin LDS-DMA as a pending LDS write and drain it in front of the next ds_read of ANY buffer); waited for by hand (vmcnt).
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

// NW = waves per workgroup (32 query rows each).  4 everywhere except the ViT shape (non-causal, 577 rows, head_dim 64), where 5 waves = 160
// rows make ceil(577 / 160) x 16 heads x 8 images = 512 workgroups = ONE round of the chip's 512 slots instead of 640 = two (round 5).
template <bool BF16, int HD, bool CAUSAL, int NW = 4>
__global__ void __launch_bounds__(NW * 64, NW == 5 ? 3 : 2)     // (second argument = waves per SIMD: two 5-wave workgroups per CU need 3)
k_flash_attn_dma(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int64_t row_stride, int64_t batch_stride, int q_off, int k_off,
                 int v_off, float scale_log2e, int seq_len, const int32_t* __restrict__ cu, int n_qblocks, int window, int nx, int B,
                 const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const int32_t* __restrict__ wg_table) {
    constexpr int KS = HD / 16;              // 16-deep MFMA steps over head_dim (QK^T)
    constexpr int DB = HD / 32;              // 32-wide head-dim blocks (PV)
    constexpr int CH = HD / 8;               // 16-byte chunks per row
    constexpr int KSLOT = HD == 96 ? 16 : 8; // chunk slots per K row in LDS (power of two: XOR swizzle)
    constexpr int KST = KSLOT * 8;           // K row stride (elements): 256 B / 128 B
    constexpr int VST = 96, VSLOT = 12;      // V row stride 192 B = 12 slots (see attn2_kernels.hip: the transposing read's bank spread)
    constexpr int KBUF = BKV * KST, VBUF = BKV * VST;
    constexpr int BQ = NW * 32;
    constexpr int KPT = BKV * KST * 2 / 1024, VPT = BKV * VST * 2 / 1024;    // 1 KiB pieces per tile: K 16 (hd 96) / 8 (hd 64), V 12
    constexpr int KPW = (KPT + NW - 1) / NW;             // K pieces per wave and tile: 4 (hd 96) / 2 (hd 64); piece wave + i * NW exists if < KPT
    constexpr int VPW = (VPT + NW - 1) / NW;             // V pieces per wave and tile: 3
    static_assert(KPT * 1024 == KBUF * 2 && VPT * 1024 == VBUF * 2, "tile = whole pieces");
    __shared__ __attribute__((aligned(1024))) uint16_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(1024))) uint16_t Vs[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    // XCD-aware placement (attn2_kernels.hip): the `nx` workgroups of one (sequence, head) are consecutive slots of ONE XCD
    int h, b, xq;
    // Scheduled launch (round 5, packed causal prefill): `wg_table[blockIdx.x]` = (sequence << 20 | head << 8 | query block) -- ONE query
    // block per workgroup, heaviest blocks first (the host orders them; the dispatcher hands workgroups out in index order as slots free
    // up, so the ragged tail of a causal batch fills with small blocks instead of waiting for paired ones: 896 paired workgroups of 14-16
    // tiles on 512 slots were two full rounds; 1 664 single blocks of 2-16 tiles pack to ~26 tile-times), entry p on XCD p % 8 = the XCD of
    // its (sequence, head).  Every query block is computed exactly as in the paired launch: bit-identical results.  MEASURED SLOWER in the
    // step (Phi-3 prefill 40.4 -> 41.4 ms, profiles/r05_attention_schedule_ab.txt): the level-by-level order spreads a head's query blocks
    // over the whole launch, so its keys / values are fetched again from the Infinity Cache instead of the XCD's L2 that the paired,
    // head-adjacent grid keeps them in -- the callers leave it off (D3D_ATTN_SCHED=1 turns it on in towers.py).
    const bool tab = wg_table != nullptr;
    if (tab) {
        const int e = wg_table[blockIdx.x];
        b = e >> 20;
        h = (e >> 8) & 0xfff;
        xq = e & 0xff;
    } else {
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
    if (tab ? xq >= n_qblocks : (CAUSAL ? xq >= (n_qblocks + 1) / 2 : xq >= n_qblocks)) return;
    const uint16_t* Qp = base + (int64_t)(q_off + h) * HD;
    const uint16_t* Kp = base + (int64_t)(k_off + h) * HD;
    const uint16_t* Vp = base + (int64_t)(v_off + h) * HD;

    auto kswz = [](int r) { return HD == 96 ? (r & 15) : ((r >> 1) & 7); };

    // ---- per-lane source offsets of this wave's pieces (bytes from the tile's first row), loop-invariant for whole tiles ------------------
    // K piece p (of KPW * NW): LDS bytes [1024 p, +1024) = rows p * RPK .. of KST * 2 bytes; lane -> (row, slot); source chunk = slot ^ f(row)
    constexpr int RPK = 1024 / (KST * 2);                // K rows per piece: 4 / 8
    auto k_src = [&](int i, int& r, int& c) {            // K piece i of this wave: this lane's tile row and source chunk
        r = (wave + i * NW) * RPK + lane / KSLOT;
        c = (lane % KSLOT) ^ kswz(r);
        c = c < CH ? c : 0;                               // (pad slots of a 256-byte row: any valid chunk)
    };
    auto v_src = [&](int i, int& r, int& c) {
        const int o = (wave + i * NW) * 1024 + lane * 16;   // byte offset in the V buffer
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
    // request key tile T into buffer BUF (wave-uniform branch: only a sequence's last, partial tile needs clamped rows)
    auto request_tile = [&](int T, int buf) {
        const char* kb = reinterpret_cast<const char*>(Kp) + (int64_t)T * tile_bytes;
        const char* vb = reinterpret_cast<const char*>(Vp) + (int64_t)T * tile_bytes;
        const int last = S - 1 - T * BKV;                  // last valid row of the tile (>= 0)
        if (NW == 4 && last >= BKV - 1) {                 // (NW 5 recomputes its offsets every tile -- a handful of VALU operations -- instead of
#pragma unroll                                            //  holding five more registers: it has to fit 168 for three waves per SIMD)
            for (int i = 0; i < KPW; ++i)
                if (KPT % NW == 0 || wave + i * NW < KPT) dma_piece(koff[i], kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
#pragma unroll
            for (int i = 0; i < VPW; ++i)
                if (VPT % NW == 0 || wave + i * NW < VPT) dma_piece(voff[i], vb, lds_v + (uint32_t)(buf * VBUF * 2 + (wave + i * NW) * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                if (!(KPT % NW == 0 || wave + i * NW < KPT)) continue;
                int r, c;
                k_src(i, r, c);
                dma_piece((uint32_t)(((int64_t)min(r, last) * row_stride + c * 8) * 2), kb, lds_k + (uint32_t)(buf * KBUF * 2 + (wave + i * NW) * 1024));
            }
#pragma unroll
            for (int i = 0; i < VPW; ++i) {
                if (!(VPT % NW == 0 || wave + i * NW < VPT)) continue;
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

  for (int pass = 0; pass < (CAUSAL ? 2 : 1); ++pass) {
    if (tab && pass == 1) break;                                         // scheduled launch: one query block per workgroup
    const int qb = (CAUSAL && !tab) ? (pass == 0 ? n_qblocks - 1 - xq : xq) : xq;
    if (CAUSAL && !tab && pass == 1 && qb == n_qblocks - 1 - xq) break;  // odd count: the middle block stands alone
    const int q0 = qb * BQ, qw = q0 + wave * 32;
    const int qrow = qw + li;                                            // this lane's query
    const int kv_len = CAUSAL ? min(seq_len, q0 + BQ) : seq_len;
    const int n_tiles = (kv_len + BKV - 1) / BKV;
    const int t_first = (window > 0 && q0 - window + 1 > 0) ? (q0 - window + 1) / BKV : 0;
    // The first key tile is requested BEFORE the query loads and the rotary arithmetic (round 6): it flies under them instead of behind them
    // (every wave is past the previous pass's last barrier: both buffers are free)
    request_tile(t_first, 0);

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + hi*8 .. +7]
    uint4 qf[KS];
    {
        const int q = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * row_stride + ks * 16 + hi * 8);
        // Rotary embedding of the queries, fused (the buffer holds un-rotated q; k was rotated in place by d3d_rope_inplace).  Half-split
        // RoPE pairs dim d with d + head_dim / 2 = fragment ks with fragment ks + KS / 2 of the SAME lane; position = the query's row in its
        // sequence.  Arithmetic = dense_kernels.hip k_rope (HF apply_rotary_pos_emb on 16-bit tensors: every product and the sum stored
        // 16-bit), so the fused and the separate path give the same bits.
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
    float m_i = -INFINITY;

    const int kmax = CAUSAL ? min(qrow, seq_len - 1) : seq_len - 1;
    const int kmin = window > 0 ? qrow - window + 1 : 0;

    // Row sums.  head_dim 64 has registers to spare: on the matrix pipe (attn2_kernels.hip: an A fragment whose row 0 is all ones puts
    // sum_k P[k][q] into row 0 of `lacc`, 4 extra MFMAs per tile, no VALU).  head_dim 96 is at the 256-register limit: there the 16
    // accumulator registers of that trick spilled (18 VGPRs in the tile loop), so each lane adds its 32 values (pairwise tree, float32,
    // before P is rounded) -- 78 -> 74 us per Phi-3 launch in spite of the extra VALU work.
    constexpr bool MFMA_SUM = HD == 64;
    const uint32_t one2 = BF16 ? 0x3F803F80u : 0x3C003C00u;
    const uint4 ones = (lane & 31) == 0 ? make_uint4(one2, one2, one2, one2) : make_uint4(0, 0, 0, 0);
    float16v lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
    float l_i = 0.f;                         // (VALU variant) this lane's half of the row sum: its 32 of the 64 keys of every tile

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // tiles [t_first, t_main): visible in full to every query of the block -> the branch-free body
    int t_main = window > 0 ? t_first : min(min(CAUSAL ? (q0 + 1) / BKV : n_tiles, S / BKV), n_tiles);
    t_main = max(t_first, t_main);

    auto ld_vf = [&](const uint16_t* Vb_, int s_, int d_) -> uint4 {
        const uint16_t* vb = Vb_ + s_ * 16 * VST + d_ * 32;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)vb);
        const v4s hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vb + 8 * VST));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hv);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };

// One key tile.  TAIL false: the branch-free body; TAIL true: per-wave activity (tile / second key block above the wave's diagonal or below
// its window) and masks.
#define FA_TILE(TAIL)                                                                                                                        \
    {                                                                                                                                        \
        const int cur = (t - t_first) & 1;                                                                                                   \
        const int key0 = t * BKV;                                                                                                            \
        const uint16_t* Kb = Ks + cur * KBUF;                                                                                                \
        const uint16_t* Vb = Vs + cur * VBUF + v_off0;                                                                                       \
        if (t + 1 < n_tiles) request_tile(t + 1, cur ^ 1);       /* in flight under the whole tile; every wave left that buffer before the last barrier */ \
        const bool tile_on = TAIL ? (!(CAUSAL && key0 > qw + 31) && !(window > 0 && key0 + BKV - 1 <= qw - window)) : true;                  \
        const bool blk1_on = TAIL ? (tile_on && !(CAUSAL && key0 + 32 > qw + 31)) : true;                                                    \
        if (tile_on) {                                                                                                                       \
            /* ---- all K fragments of the tile, then S^T = K Q^T : st{b}[4j + r] = S[key0 + 32b + 8j + 4hi + r][qrow] ---------------- */   \
            float16v st0, st1;                                                                                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; }                                                   \
            const uint16_t* Ka = Kb + li * KST;                                                                                              \
            const uint16_t* Kc = Kb + (32 + li) * KST;                                                                                       \
            uint4 kf[2][KS];                                                                                                                 \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) kf[0][ks] = *reinterpret_cast<const uint4*>(Ka + (((ks * 2 + hi) ^ kswz(li)) << 3)); \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) kf[1][ks] = *reinterpret_cast<const uint4*>(Kc + (((ks * 2 + hi) ^ kswz(li)) << 3)); \
            }                                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                 /* the reads are issued as one burst; the MFMAs wait for them one by one */    \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                                          \
                    st0 = mfma32<BF16>(kf[0][ks], qf[ks], st0);                                                                              \
                    st1 = mfma32<BF16>(kf[1][ks], qf[ks], st1);                                                                              \
                }                                                                                                                            \
            } else {                                                                                                                         \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) st0 = mfma32<BF16>(kf[0][ks], qf[ks], st0);                                \
            }                                                                                                                                \
            /* ---- all V^T fragments of the tile, requested now, used after the softmax ------------------------------------------------ */  \
            uint4 vf[4][DB];                                                                                                                 \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                                    \
            _Pragma("unroll") for (int d = 0; d < DB; ++d) vf[s][d] = ld_vf(Vb, s, d);                                                       \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int s = 2; s < 4; ++s)                                                                                \
                _Pragma("unroll") for (int d = 0; d < DB; ++d) vf[s][d] = ld_vf(Vb, s, d);                                                   \
            }                                                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                                               \
            if (TAIL) {   /* masking: key - key0 - 4hi in [lo_, hi_] is visible */                                                           \
                const bool need_mask = (CAUSAL && key0 + BKV - 1 > qw) || (key0 + BKV > seq_len) || (window > 0 && key0 <= qw + 31 - window); \
                if (need_mask) {                                                                                                             \
                    const int lo_ = kmin - key0 - hi * 4, hi_ = kmax - key0 - hi * 4;                                                        \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                            \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                          \
                        const int o_ = j * 8 + r;                                                                                            \
                        if (o_ > hi_ || o_ < lo_) st0[4 * j + r] = -INFINITY;                                                                \
                        if (o_ + 32 > hi_ || o_ + 32 < lo_) st1[4 * j + r] = -INFINITY;                                                      \
                    }                                                                                                                        \
                }                                                                                                                            \
            }                                                                                                                                \
            /* ---- online softmax, base 2, one query per lane pair ------------------------------------------------------------------ */    \
            float tmax = max3(st0[0], st0[1], st0[2]);                                                                                       \
            _Pragma("unroll") for (int r = 3; r + 1 < 16; r += 2) tmax = max3(tmax, st0[r], st0[r + 1]);                                     \
            tmax = fmaxf(tmax, st0[15]);                                                                                                     \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int r = 0; r + 1 < 16; r += 2) tmax = max3(tmax, st1[r], st1[r + 1]);                                 \
            }                                                                                                                                \
            tmax = pair_max(tmax);                                                                                                           \
            const float tm = tmax * scale_log2e;                                                                                             \
            const bool keep = __all(tm <= m_i + 8.0f);                     /* deferred rescale: P stays <= 2^8 */                            \
            const float m_new = keep ? m_i : fmaxf(m_i, tm);                                                                                 \
            const float m_use = (TAIL && m_new == -INFINITY) ? 0.f : m_new; /* no visible key yet (window / padding): exp2(-inf) = 0 */      \
            if (!keep) {                                                                                                                     \
                const float alpha = __builtin_amdgcn_exp2f(m_i - m_use);                                                                     \
                _Pragma("unroll") for (int d = 0; d < DB; ++d)                                                                               \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;                                                          \
                if (MFMA_SUM) lacc[0] *= alpha; else l_i *= alpha;                                                                           \
            }                                                                                                                                \
            m_i = m_new;                                                                                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) st0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[r], scale_log2e, -m_use));     \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) st1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st1[r], scale_log2e, -m_use)); \
            }                                                                                                                                \
            if (!MFMA_SUM) {   /* row sum: pairwise tree, 16 + 16 values */                                                                  \
                float ts = 0.f;                                                                                                              \
                _Pragma("unroll") for (int r = 0; r < 16; r += 4) ts += (st0[r] + st0[r + 1]) + (st0[r + 2] + st0[r + 3]);                   \
                if (blk1_on) {                                                                                                               \
                    _Pragma("unroll") for (int r = 0; r < 16; r += 4) ts += (st1[r] + st1[r + 1]) + (st1[r + 2] + st1[r + 3]);               \
                }                                                                                                                            \
                l_i += ts;                                                                                                                   \
            }                                                                                                                                \
            /* P^T fragments: MFMA step s covers keys 16s .. 16s+15 of the tile; k-slot (hi*8 + jj*4 + r) = key 16s + 8jj + 4hi + r */       \
            uint4 pf[4];                                                                                                                     \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                                  \
                pf[s].x = pack2<BF16>(st0[8 * s + 0], st0[8 * s + 1]);                                                                       \
                pf[s].y = pack2<BF16>(st0[8 * s + 2], st0[8 * s + 3]);                                                                       \
                pf[s].z = pack2<BF16>(st0[8 * s + 4], st0[8 * s + 5]);                                                                       \
                pf[s].w = pack2<BF16>(st0[8 * s + 6], st0[8 * s + 7]);                                                                       \
                pf[2 + s].x = pack2<BF16>(st1[8 * s + 0], st1[8 * s + 1]);                                                                   \
                pf[2 + s].y = pack2<BF16>(st1[8 * s + 2], st1[8 * s + 3]);                                                                   \
                pf[2 + s].z = pack2<BF16>(st1[8 * s + 4], st1[8 * s + 5]);                                                                   \
                pf[2 + s].w = pack2<BF16>(st1[8 * s + 6], st1[8 * s + 7]);                                                                   \
            }                                                                                                                                \
            /* ---- O^T += V^T P^T out of registers --------------- -------------------------------------------------------------------- */  \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                                  \
                _Pragma("unroll") for (int d = 0; d < DB; ++d) oacc[d] = mfma32<BF16>(vf[s][d], pf[s], oacc[d]);                             \
                if (MFMA_SUM) lacc = mfma32<BF16>(ones, pf[s], lacc);                                                                        \
            }                                                                                                                                \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int s = 2; s < 4; ++s) {                                                                              \
                    _Pragma("unroll") for (int d = 0; d < DB; ++d) oacc[d] = mfma32<BF16>(vf[s][d], pf[s], oacc[d]);                         \
                    if (MFMA_SUM) lacc = mfma32<BF16>(ones, pf[s], lacc);                                                                    \
                }                                                                                                                            \
            }                                                                                                                                \
        }                                                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* this wave's pieces of tile t + 1 have landed */                            \
        __syncthreads();          /* tile t+1 is visible; every wave is done with buffer `cur` */                                            \
    }

    int t = t_first;
    for (; t < t_main; ++t) FA_TILE(false)
    for (; t < n_tiles; ++t) FA_TILE(true)
#undef FA_TILE
    // MFMA variant: the row sum of query li lives in lane li (hi = 0), register 0; VALU variant: the two lanes of a query hold its two halves
    l_i = pair_sum(MFMA_SUM ? (hi == 0 ? lacc[0] : 0.f) : l_i);
    // ---- epilogue: lane holds O[qrow][32d + 8j + 4hi + r]; lane pairs exchange so that each stores 16 contiguous bytes ---------------------
    const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
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
  }   // pass
}

}  // namespace

extern "C" {

// d3d_flash_attention_v3 + the queries' rotary embedding: the buffer holds UN-ROTATED q (and rotated k); rope_cos / rope_sin are the
// (positions, head_dim / 2) float32 tables of d3d_rope_inplace, a query's position is its row inside its sequence.
int32_t d3d_flash_attention_v3_rope_q(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                                      int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens,
                                      int32_t window, const float* rope_cos, const float* rope_sin, int32_t dtype, void* stream);

int32_t d3d_flash_attention_v3_sched(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                                     int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens,
                                     int32_t window, const float* rope_cos, const float* rope_sin, const int32_t* wg_table, int32_t n_wg, int32_t dtype,
                                     void* stream);

// Same contract as d3d_flash_attention_v2 (attn2_kernels.hip); 128 query rows per workgroup.
int32_t d3d_flash_attention_v3(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                               int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens, int32_t window,
                               int32_t dtype, void* stream) {
    return d3d_flash_attention_v3_rope_q(qkv, out, B, S, H, head_dim, row_stride, batch_stride, q_off, k_off, v_off, causal, seq_len, cu_seqlens, window,
                                         nullptr, nullptr, dtype, stream);
}

int32_t d3d_flash_attention_v3_rope_q(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                                      int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens,
                                      int32_t window, const float* rope_cos, const float* rope_sin, int32_t dtype, void* stream) {
    return d3d_flash_attention_v3_sched(qkv, out, B, S, H, head_dim, row_stride, batch_stride, q_off, k_off, v_off, causal, seq_len, cu_seqlens, window,
                                        rope_cos, rope_sin, nullptr, 0, dtype, stream);
}

int32_t d3d_flash_attention_v3_sched(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                                     int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens,
                                     int32_t window, const float* rope_cos, const float* rope_sin, const int32_t* wg_table, int32_t n_wg, int32_t dtype,
                                     void* stream) {
    if (B <= 0 || S <= 0) return D3D_OK;
    if (wg_table && (n_wg <= 0 || H > 4095 || B > 2047 || (S + 127) / 128 > 256)) {
        d3d_set_error_("d3d_flash_attention_v3_sched: a workgroup table needs n_wg > 0, H <= 4095, B <= 2047, S <= 32768 (8-bit query-block field)");
        return D3D_EINVAL;
    }
    if ((head_dim != 64 && head_dim != 96) || (row_stride & 7) || (batch_stride & 7) || window < 0 || (window > 0 && !causal)) {
        d3d_set_error_("d3d_flash_attention_v3: head_dim must be 64 or 96; strides multiples of 8 elements; a window needs causal");
        return D3D_EINVAL;
    }
    if ((int64_t)BKV * row_stride * 2 >= (1ll << 31)) {
        d3d_set_error_("d3d_flash_attention_v3: a 64-row tile of the QKV buffer must span less than 2 GiB (32-bit per-lane offsets)");
        return D3D_EINVAL;
    }
    if ((rope_cos == nullptr) != (rope_sin == nullptr)) {
        d3d_set_error_("d3d_flash_attention_v3_rope_q: rope_cos and rope_sin come together");
        return D3D_EINVAL;
    }
    // Round 6: D3D_ATTN_V4=1 sends launches without a window and without a workgroup table to the software-pipelined kernel of
    // attn4_kernels.hip (read per call, so that one process can compare the two).  OFF by default: measured 2-3 % faster on the Phi-3 shape
    // and level on the ViT shape (profiles/r06_attention_ab.txt) -- not worth a change of the product path's last bits.
    if (!wg_table && (window == 0 || window >= S)) {
        const char* e4 = getenv("D3D_ATTN_V4");
        if (e4 && e4[0] == '1')
            return d3d_flash_attention_v4(qkv, out, B, S, H, head_dim, row_stride, batch_stride, q_off, k_off, v_off, causal, seq_len, cu_seqlens, rope_cos,
                                          rope_sin, dtype, stream);
    }
    const float sl2 = 1.4426950408889634f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    // 5-wave workgroups (160 query rows) when that saves a round of the chip's slots (two workgroups per CU): dense non-causal head_dim 64
    // -- the ViT shape: 577 rows x 16 heads x 8 images = 640 workgroups of 128 rows on 512 slots, 512 of 160 rows.  MEASURED SLOWER
    // (profiles/r05_attention_schedule_ab.txt: each ViT tower 4.97 -> 5.14 ms; ten waves per CU at three per SIMD contend for the VALU
    // that bounds this kernel): OFF unless D3D_ATTN_NW5=1; bit-identical results, kept as a tested knob.
    bool nw5 = false;
    if (!causal && head_dim == 64 && !cu_seqlens && !wg_table) {
        const char* e5 = getenv("D3D_ATTN_NW5");                     // (read per call: the test compares the two launches in one process)
        const bool allow = e5 && e5[0] == '1';
        static const int slots = [] {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            return 2 * cus;
        }();
        const int64_t n4 = (int64_t)((S + 127) / 128) * H * B, n5 = (int64_t)((S + 159) / 160) * H * B;
        nw5 = allow && (n5 + slots - 1) / slots < (n4 + slots - 1) / slots;
    }
    const int bq = nw5 ? 160 : BQ4;
    const int nqb = (S + bq - 1) / bq;
    if (window >= S) window = 0;
    const int nx = causal ? (nqb + 1) / 2 : nqb;
    dim3 grid(wg_table ? (unsigned)n_wg : (unsigned)((int64_t)nx * H * B)), block(nw5 ? 320 : NW4 * 64);
    if (nw5) {
        if (dtype == 0)
            hipLaunchKernelGGL((k_flash_attn_dma<true, 64, false, 5>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, sl2, seq_len,
                               cu_seqlens, nqb, window, nx, B, rope_cos, rope_sin, wg_table);
        else
            hipLaunchKernelGGL((k_flash_attn_dma<false, 64, false, 5>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, sl2, seq_len,
                               cu_seqlens, nqb, window, nx, B, rope_cos, rope_sin, wg_table);
        D3D_LAUNCH_CHECK();
    }
#define D3D_FA3(BF, HDV, CA) hipLaunchKernelGGL((k_flash_attn_dma<BF, HDV, CA>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, \
                                                sl2, seq_len, cu_seqlens, nqb, window, nx, B, rope_cos, rope_sin, wg_table)
    if (dtype == 0) {
        if (head_dim == 96) { if (causal) D3D_FA3(true, 96, true); else D3D_FA3(true, 96, false); }
        else { if (causal) D3D_FA3(true, 64, true); else D3D_FA3(true, 64, false); }
    } else {
        if (head_dim == 96) { if (causal) D3D_FA3(false, 96, true); else D3D_FA3(false, 96, false); }
        else { if (causal) D3D_FA3(false, 64, true); else D3D_FA3(false, 64, false); }
    }
#undef D3D_FA3
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
