// attn2_kernels.hip -- second-generation fused self-attention forward for gfx950 (bf16/fp16, head_dim 64 / 96, causal or full, dense or
// packed variable-length, optional sliding window), reading q/k/v in place from the fused QKV projection buffer.
//
// What changed against attn_kernels.hip::k_flash_attn (kept as the A/B baseline, D3D_ATTN_V2=0) and why -- that kernel ran at 0.15-0.17 of
// the MFMA peak with the matrix pipe busy 13 % and a SIMD issuing VALU 34 % of the time (profiles/r01_pmc_attn_vit.txt):
//   * v_mfma_f32_32x32x16 instead of 16x16x32.  S^T = K Q^T on 32x32 tiles puts ONE query column in a lane (and its partner lane + 32):
//     a lane holds 32 of the 64 scores of its query, so the row maximum / row sum are 31 register-local operations plus ONE
//     v_permlane32_swap step (the 16x16 layout held two query sets per lane, each with two cross-lane steps), there is one running
//     max / sum / rescale decision per lane instead of two, and half as many MFMA and LDS-read instructions are issued per FLOP.
//   * P^T is ALREADY the B operand of O^T = V^T P^T: a lane's accumulator registers hold keys {4*hi + 8*j + r} of each 32-key block for
//     its query; declaring k-slot (hi*8 + jj*4 + r) of a 16-key MFMA step to be key (16*s + 8*jj + 4*hi + r) makes the lane's own 8
//     packed values the B fragment -- no cross-lane movement between the two GEMMs -- and the V^T fragment (A operand) in that slot
//     order is two ds_read_b64_tr_b16 of [4 keys][16 dims] blocks.
//   * K/V tiles are DOUBLE-buffered in LDS with ONE barrier per key tile: tile t+1 is written (from registers, loaded a tile earlier)
//     between the softmax and the PV product of tile t, into the buffer every wave finished reading before the previous barrier.
//   * diagonal tiles: a 32-key block entirely above a wave's diagonal is skipped (its QK^T and PV MFMAs and its softmax work).
//   * the epilogue exchanges half-rows between lane pairs (v_permlane32_swap) and stores 16 bytes per instruction.
//   * sliding window (Phi-3-mini-4k: 2047 keys): key <= query - window is masked, tiles entirely outside the window are never visited.
//   * XCD-aware grid (see the kernel): the query blocks of one (sequence, head) share an L2.
// Measured against v1 in one process (tools/bench_attn_ab.py): Phi-3 packed causal 8 x 1024: 107 -> 93 us, the step's lengths 100 -> 92 us, ViT
// 8 x 577: 28.0 -> 25.6 us.  In-kernel stamps of this structure and of a software-pipelined / hand-scheduled variant (QK^T of tile t+1
// beside the softmax of tile t) say what is left: a wave's tile costs ~1 000 cycles of VALU issue (32 v_exp_f32 at a
// quarter rate + ~120 plain instructions) against 768 cycles of matrix pipe, and two waves share one SIMD's VALU.
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
using float2v = __attribute__((ext_vector_type(2))) float;
using v4s = __attribute__((ext_vector_type(4))) short;

constexpr int BKV = 64;      // keys per tile; a workgroup of NW waves owns NW * 32 query rows

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

__device__ __forceinline__ float max3(float a, float b, float c) {
    // NOT inline asm: the hazard recogniser does not see an asm statement as a VALU read, so the wait states a VALU needs after the MFMA
    // that wrote its operands were not inserted and a max taken straight off the score accumulators read registers still in flight
    // (run-to-run differences in the last bits).  hipcc folds the nested maxima into v_max3_f32 by itself.
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}
// op(v, value of lane ^ 32): v_permlane32_swap exchanges the upper half of vdst with the lower half of src; with both operands holding
// the same value, op(vdst', src') is the xor-32 butterfly.  Inline asm: the builtin folds away when both operands are the same SSA
// value; s_nop 1 = the two wait states a VALU write needs before v_permlane32_swap reads it (LLVM gfx950 hazard rule).
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

#ifdef D3D_FA_STAMP
// diagnostics build (tools/attn_barrier_stamps.py): [0] cycles inside the key-tile loops summed over waves, [1] of those, cycles between
// arriving at a tile's barrier and leaving it, [2] wave-tiles, [3] waves
__device__ unsigned long long g_fa_stamp[4];
#endif

template <bool BF16, int HD, bool CAUSAL, int NW>
__global__ void __launch_bounds__(NW * 64, 2)
k_flash_attn32(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int64_t row_stride, int64_t batch_stride, int q_off, int k_off,
               int v_off, float scale_log2e, int seq_len, const int32_t* __restrict__ cu, int n_qblocks, int window, int nx, int B) {
    constexpr int NT = NW * 64, BQ = NW * 32;
    constexpr int KS = HD / 16;              // 16-deep MFMA steps over head_dim (QK^T)
    constexpr int DB = HD / 32;              // 32-wide head-dim blocks (PV)
    constexpr int CH = HD / 8;               // 16-byte chunks per row
    constexpr int KCH = HD == 96 ? 16 : 8;   // chunk slots per K row in LDS (power of two: XOR swizzle)
    constexpr int KST = KCH * 8;             // K row stride (elements): 256 B / 128 B
    constexpr int VST = 96;                  // V row stride (elements) = 192 B = 64 (mod 256): the 4 rows x 64 B a half-wave's transposing
                                             // read touches fall into 4 different quarter-rows of the 256-byte bank space
    constexpr int KBUF = BKV * KST, VBUF = BKV * VST;
    __shared__ __attribute__((aligned(16))) uint16_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and PROVABLY so: tile / block activity below are scalar branches
    const int li = lane & 31, hi = lane >> 5;
    // XCD-aware placement.  Workgroups go to the 8 XCDs round-robin by linear id, each XCD has its own L2, and every query block of a
    // (sequence, head) streams the SAME K/V rows: with (x, head, batch) as the grid the blocks of one head landed on different XCDs and each
    // pulled its own copy through the fabric (~0.45 GB of K/V reads per Phi-3 launch, 192-byte row pieces at an 18 KB stride).  Now the `nx`
    // workgroups of one (sequence, head) are consecutive slots of ONE XCD (dispatched together, walking the key tiles in step), and
    // neighbouring slots are neighbouring heads of the same sequence, whose 192-byte row pieces share 128-byte lines: -7 ... -14 %.
    int h, b, xq;
    {
        const int lin = blockIdx.x, G = H * B, G8 = G & ~7;
        if (lin < G8 * nx) {
            const int xcd = lin & 7, slot = lin >> 3, k = slot / nx;
            xq = slot - k * nx;
            if ((H & 7) == 0) {
                const int hp = H >> 3;                                       // heads per XCD
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

    // K swizzle: chunk c of row r lives in slot c ^ f(r); f makes the 16 rows of a ds_read_b128 lane group hit 16 different 16-byte
    // slots of the 256-byte bank space (hd 96: 256-byte rows, f = r & 15; hd 64: 128-byte rows alternate halves, f = (r >> 1) & 7)
    auto kswz = [](int r) { return HD == 96 ? (r & 15) : ((r >> 1) & 7); };

    constexpr int NCHUNK = BKV * CH;                     // 16-byte pieces of a K (or V) tile
    constexpr int NK = (NCHUNK + NT - 1) / NT;           // pieces per thread: 3 / 2 (4 waves, hd 96 / 64), 2 / 1 (8 waves; hd 96: the second only in waves 0-3)
    static_assert(NK >= 1 && NK <= 3, "tile shape");
    auto piece_ok = [&](int i) { return (NCHUNK % NT == 0) || (tid + i * NT < NCHUNK); };      // wave-uniform
    uint4 kr0, kr1 = make_uint4(0, 0, 0, 0), kr2 = make_uint4(0, 0, 0, 0), vr0, vr1 = make_uint4(0, 0, 0, 0), vr2 = make_uint4(0, 0, 0, 0);
    auto ld_kv = [&](const uint16_t* P, int key0_, int i) -> uint4 {          // rows beyond the sequence are clamped (valid memory, masked later)
        const int c = tid + i * NT;
        const int kr = min(key0_ + c / CH, S - 1);
        return *reinterpret_cast<const uint4*>(P + (int64_t)kr * row_stride + (c % CH) * 8);
    };
    auto st_k = [&](uint16_t* Kb, int i, const uint4& v) {
        const int c = tid + i * NT, r = c / CH;
        *reinterpret_cast<uint4*>(Kb + r * KST + (((c % CH) ^ kswz(r)) << 3)) = v;
    };
    auto st_v = [&](uint16_t* Vb, int i, const uint4& v) {
        const int c = tid + i * NT;
        *reinterpret_cast<uint4*>(Vb + (c / CH) * VST + (c % CH) * 8) = v;
    };
#define FA_LOAD_TILE(T)                                                        \
    {                                                                          \
        const int k0_ = (T) * BKV;                                             \
        kr0 = ld_kv(Kp, k0_, 0);                                               \
        if constexpr (NK > 1) { if (piece_ok(1)) kr1 = ld_kv(Kp, k0_, 1); }    \
        if constexpr (NK > 2) { if (piece_ok(2)) kr2 = ld_kv(Kp, k0_, 2); }    \
        vr0 = ld_kv(Vp, k0_, 0);                                               \
        if constexpr (NK > 1) { if (piece_ok(1)) vr1 = ld_kv(Vp, k0_, 1); }    \
        if constexpr (NK > 2) { if (piece_ok(2)) vr2 = ld_kv(Vp, k0_, 2); }    \
    }
#define FA_STORE_TILE(BUF)                                                     \
    {                                                                          \
        uint16_t* Kb_ = Ks + (BUF) * KBUF;                                     \
        uint16_t* Vb_ = Vs + (BUF) * VBUF;                                     \
        st_k(Kb_, 0, kr0);                                                     \
        if constexpr (NK > 1) { if (piece_ok(1)) st_k(Kb_, 1, kr1); }          \
        if constexpr (NK > 2) { if (piece_ok(2)) st_k(Kb_, 2, kr2); }          \
        st_v(Vb_, 0, vr0);                                                     \
        if constexpr (NK > 1) { if (piece_ok(1)) st_v(Vb_, 1, vr1); }          \
        if constexpr (NK > 2) { if (piece_ok(2)) st_v(Vb_, 2, vr2); }          \
    }
    using lds_v4s = __attribute__((address_space(3))) v4s;
    // V fragment base (A operand of O^T = V^T P^T through the transposing read): lane addresses key 16s + 8jj + 4hi + (l & 15) / 4,
    // dims 32d + 16 ((l >> 4) & 1) + 4 (l & 3) and receives dim 32d + li of 4 consecutive keys
    const int v_off0 = (hi * 4 + ((lane & 15) >> 2)) * VST + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;

  for (int pass = 0; pass < (CAUSAL ? 2 : 1); ++pass) {
    const int qb = CAUSAL ? (pass == 0 ? n_qblocks - 1 - xq : xq) : xq;
    if (CAUSAL && pass == 1 && qb == n_qblocks - 1 - xq) break;          // odd count: the middle block stands alone
    const int q0 = qb * BQ, qw = q0 + wave * 32;
    const int qrow = qw + li;                                            // this lane's query

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[qrow][ks*16 + hi*8 .. +7]
    uint4 qf[KS];
    {
        const int q = qrow < S ? qrow : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(Qp + (int64_t)q * row_stride + ks * 16 + hi * 8);
    }
    float16v oacc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_i = -INFINITY;

    const int kv_len = CAUSAL ? min(seq_len, q0 + BQ) : seq_len;
    const int n_tiles = (kv_len + BKV - 1) / BKV;
    // sliding window: the first key any query of this block may see is q0 - window + 1 -> start at its tile
    const int t_first = (window > 0 && q0 - window + 1 > 0) ? (q0 - window + 1) / BKV : 0;
    // this lane's visible key range [kmin, kmax]
    const int kmax = CAUSAL ? min(qrow, seq_len - 1) : seq_len - 1;
    const int kmin = window > 0 ? qrow - window + 1 : 0;

    // row sums on the matrix pipe: one extra MFMA per 16-key step with an A fragment whose row 0 is all ones puts sum_k P[k][q] -- of the
    // 16-bit P the PV product consumes -- into row 0 of `lacc` (lane q, register 0); it is rescaled with O and needs no VALU adds
    const uint32_t one2 = BF16 ? 0x3F803F80u : 0x3C003C00u;
    const uint4 ones = (lane & 31) == 0 ? make_uint4(one2, one2, one2, one2) : make_uint4(0, 0, 0, 0);
    float16v lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] = 0.f;

    FA_LOAD_TILE(t_first)
    FA_STORE_TILE(0)
    if (t_first + 1 < n_tiles) FA_LOAD_TILE(t_first + 1)
    __syncthreads();

    // Tiles [t_first, t_main): visible IN FULL to every query of the block (below the block's diagonal, inside the sequence, no window edge),
    // and tile t + 2 -- requested while tile t is computed -- lies inside the sequence: the branch-free body with unclamped loads off a
    // wave-uniform base.  The rest (the tiles around the diagonal, the sequence end, the window) run the general body.
    int t_main = window > 0 ? t_first : min(min(CAUSAL ? (q0 + 1) / BKV : n_tiles, S / BKV - 2), n_tiles - 1);
    t_main = max(t_first, t_main);
    const int64_t tile_bytes = (int64_t)BKV * row_stride * 2;
    const char* kbase = reinterpret_cast<const char*>(Kp) + (int64_t)(t_first + 2) * tile_bytes;       // tile t + 2 of the main body
    const char* vbase = reinterpret_cast<const char*>(Vp) + (int64_t)(t_first + 2) * tile_bytes;
    uint32_t goff[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const int c = tid + i * NT;
        goff[i] = (uint32_t)(((int64_t)(c / CH) * row_stride + (c % CH) * 8) * 2);
    }
    auto ld_vf = [&](const uint16_t* Vb_, int s_, int d_) -> uint4 {
        const uint16_t* vb = Vb_ + s_ * 16 * VST + d_ * 32;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)vb);
        const v4s hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(vb + 8 * VST));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hv);
        return make_uint4(l2.x, l2.y, h2.x, h2.y);
    };

// One key tile.  TAIL false: the branch-free body; TAIL true: per-wave activity (tile / second key block above the wave's diagonal or
// below its window), masks, clamped loads.
#define FA_TILE(TAIL)                                                                                                                        \
    {                                                                                                                                        \
        const int cur = (t - t_first) & 1;                                                                                                   \
        const int key0 = t * BKV;                                                                                                            \
        const uint16_t* Kb = Ks + cur * KBUF;                                                                                                \
        const uint16_t* Vb = Vs + cur * VBUF + v_off0;                                                                                       \
        const bool tile_on = TAIL ? (!(CAUSAL && key0 > qw + 31) && !(window > 0 && key0 + BKV - 1 <= qw - window)) : true;                  \
        const bool blk1_on = TAIL ? (tile_on && !(CAUSAL && key0 + 32 > qw + 31)) : true;                                                    \
        uint4 pf[4];                                                                                                                         \
        if (tile_on) {                                                                                                                       \
            /* ---- S^T = K Q^T : st{b}[4j + r] = S[key0 + 32b + 8j + 4hi + r][qrow]; block 1's fragments are requested one behind each   */  \
            /* MFMA of block 0's chain (hipcc left to itself kept two reads in flight and waited for each pair) ------------------------- */  \
            float16v st0, st1;                                                                                                               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) { st0[r] = 0.f; st1[r] = 0.f; }                                                   \
            const uint16_t* Ka = Kb + li * KST;                                                                                              \
            const uint16_t* Kc = Kb + (32 + li) * KST;                                                                                       \
            auto k_frag = [&](int ks, int blk) -> uint4 {                                                                                    \
                return *reinterpret_cast<const uint4*>((blk ? Kc : Ka) + (((ks * 2 + hi) ^ kswz(li)) << 3));   /* kswz(32 + li) == kswz(li) */ \
            };                                                                                                                               \
            if (blk1_on) {                                                                                                                   \
                /* MFMA slot i = step i/2 of key block i&1 (two accumulators alternate); its K fragment is requested two slots ahead, so */ \
                /* three fragments are live instead of twelve */                                                                            \
                uint4 kf0 = k_frag(0, 0), kf1 = k_frag(0, 1), kf2 = make_uint4(0, 0, 0, 0);                                                  \
                _Pragma("unroll") for (int i = 0; i < 2 * KS; ++i) {                                                                         \
                    const uint4 kf_ = (i % 3 == 0) ? kf0 : (i % 3 == 1) ? kf1 : kf2;                                                         \
                    if (i + 2 < 2 * KS) {                                                                                                    \
                        const uint4 nx_ = k_frag((i + 2) >> 1, (i + 2) & 1);                                                                 \
                        if ((i + 2) % 3 == 0) kf0 = nx_; else if ((i + 2) % 3 == 1) kf1 = nx_; else kf2 = nx_;                               \
                    }                                                                                                                        \
                    if (i & 1) st1 = mfma32<BF16>(kf_, qf[i >> 1], st1);                                                                     \
                    else st0 = mfma32<BF16>(kf_, qf[i >> 1], st0);                                                                           \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                       \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                       \
                }                                                                                                                            \
            } else {                                                                                                                         \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) st0 = mfma32<BF16>(k_frag(ks, 0), qf[ks], st0);                            \
            }                                                                                                                                \
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
                lacc[0] *= alpha;                                                                                                            \
            }                                                                                                                                \
            m_i = m_new;                                                                                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) st0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st0[r], scale_log2e, -m_use));     \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) st1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st1[r], scale_log2e, -m_use)); \
            }                                                                                                                                \
            /* P^T fragments: MFMA step s covers keys 16s .. 16s+15 of the tile; k-slot (hi*8 + jj*4 + r) = key 16s + 8jj + 4hi + r */       \
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
        }                                                                                                                                    \
        /* ---- stage tile t+1 (in registers since the previous iteration) into the other buffer; request tile t+2 ----------------------- */  \
        if (TAIL) {                                                                                                                          \
            if (t + 1 < n_tiles) {                                                                                                           \
                FA_STORE_TILE(cur ^ 1)                                                                                                       \
                if (t + 2 < n_tiles) FA_LOAD_TILE(t + 2)                                                                                     \
            }                                                                                                                                \
        } else {                                                                                                                             \
            FA_STORE_TILE(cur ^ 1)                                                                                                           \
            kr0 = *reinterpret_cast<const uint4*>(kbase + goff[0]);                                                                          \
            if constexpr (NK > 1) { if (piece_ok(1)) kr1 = *reinterpret_cast<const uint4*>(kbase + goff[NK > 1 ? 1 : 0]); }                  \
            if constexpr (NK > 2) { if (piece_ok(2)) kr2 = *reinterpret_cast<const uint4*>(kbase + goff[NK - 1]); }                          \
            vr0 = *reinterpret_cast<const uint4*>(vbase + goff[0]);                                                                          \
            if constexpr (NK > 1) { if (piece_ok(1)) vr1 = *reinterpret_cast<const uint4*>(vbase + goff[NK > 1 ? 1 : 0]); }                  \
            if constexpr (NK > 2) { if (piece_ok(2)) vr2 = *reinterpret_cast<const uint4*>(vbase + goff[NK - 1]); }                          \
            kbase += tile_bytes;                                                                                                             \
            vbase += tile_bytes;                                                                                                             \
        }                                                                                                                                    \
        /* ---- O^T += V^T P^T (+ the row sums): steps in (s, d) order, the V fragments of step (s + 1, d) requested before the MFMA of   */  \
        /* step (s, d) is issued -------------------------------------------------------------------------------------------------------- */  \
        if (tile_on) {                                                                                                                       \
            uint4 vf[2][DB];                                                                                                                 \
            _Pragma("unroll") for (int d = 0; d < DB; ++d) vf[0][d] = ld_vf(Vb, 0, d);                                                       \
            if (blk1_on) {                                                                                                                   \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                              \
                    _Pragma("unroll") for (int d = 0; d < DB; ++d) {                                                                         \
                        if (s + 1 < 4) vf[(s + 1) & 1][d] = ld_vf(Vb, s + 1, d);                                                             \
                        oacc[d] = mfma32<BF16>(vf[s & 1][d], pf[s], oacc[d]);                                                                \
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                                   \
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                   \
                    }                                                                                                                        \
                    lacc = mfma32<BF16>(ones, pf[s], lacc);                                                                                  \
                }                                                                                                                            \
            } else {                                                                                                                         \
                _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                              \
                    _Pragma("unroll") for (int d = 0; d < DB; ++d) {                                                                         \
                        if (s + 1 < 2) vf[(s + 1) & 1][d] = ld_vf(Vb, s + 1, d);                                                             \
                        oacc[d] = mfma32<BF16>(vf[s & 1][d], pf[s], oacc[d]);                                                                \
                    }                                                                                                                        \
                    lacc = mfma32<BF16>(ones, pf[s], lacc);                                                                                  \
                }                                                                                                                            \
            }                                                                                                                                \
        }                                                                                                                                    \
        FA_BAR_STAMP_BEGIN                                                                                                                   \
        __syncthreads();          /* tile t+1 is visible; every wave is done with buffer `cur` */                                            \
        FA_BAR_STAMP_END                                                                                                                     \
    }

#ifdef D3D_FA_STAMP
    unsigned long long st_bar = 0, st_b0 = 0;
    const unsigned long long st_t0 = __builtin_readcyclecounter();
#define FA_BAR_STAMP_BEGIN st_b0 = __builtin_readcyclecounter();
#define FA_BAR_STAMP_END st_bar += __builtin_readcyclecounter() - st_b0;
#else
#define FA_BAR_STAMP_BEGIN
#define FA_BAR_STAMP_END
#endif
    int t = t_first;
    for (; t < t_main; ++t) FA_TILE(false)
    for (; t < n_tiles; ++t) FA_TILE(true)
#undef FA_TILE
#undef FA_BAR_STAMP_BEGIN
#undef FA_BAR_STAMP_END
#ifdef D3D_FA_STAMP
    if (lane == 0) {
        atomicAdd(&g_fa_stamp[0], __builtin_readcyclecounter() - st_t0);
        atomicAdd(&g_fa_stamp[1], st_bar);
        atomicAdd(&g_fa_stamp[2], (unsigned long long)(n_tiles - t_first));
        atomicAdd(&g_fa_stamp[3], 1ull);
    }
#endif
    // the row sum of query li lives in lane li (hi = 0), register 0: hand it to the partner lane
    float l_i = lacc[0];
    l_i = pair_sum(hi == 0 ? l_i : 0.f);
    // ---- epilogue: lane holds O[qrow][32d + 8j + 4hi + r]; lane pairs exchange so that each stores 16 contiguous bytes ---------------------
    const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
    uint16_t* op = out + ((row0 + qrow) * H + h) * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {                 // j = 2jp, 2jp+1: dims 32d + 16jp + {0..3 (hi 0), 4..7 (hi 1), 8..11 (hi 0), 12..15 (hi 1)}
            uint32_t a0 = pack2<BF16>(oacc[d][8 * jp + 0] * inv, oacc[d][8 * jp + 1] * inv), a1 = pack2<BF16>(oacc[d][8 * jp + 2] * inv, oacc[d][8 * jp + 3] * inv);
            uint32_t b0 = pack2<BF16>(oacc[d][8 * jp + 4] * inv, oacc[d][8 * jp + 5] * inv), b1 = pack2<BF16>(oacc[d][8 * jp + 6] * inv, oacc[d][8 * jp + 7] * inv);
            // v_permlane32_swap(vdst = a, src = b) swaps upper-half(a) with lower-half(b): a lower lane ends with [own a | upper's a] = dims
            // 0..7 of the group pair, an upper lane with [lower's b | own b] = dims 8..15
            const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            if (qrow < S) *reinterpret_cast<uint4*>(op + d * 32 + jp * 16 + hi * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    }
  }   // pass
#undef FA_LOAD_TILE
#undef FA_STORE_TILE
}

}  // namespace

extern "C" {

#ifdef D3D_FA_STAMP
int32_t d3d_fa_stamp_read(unsigned long long* host4, int32_t reset) {
    if (hipMemcpyFromSymbol(host4, HIP_SYMBOL(g_fa_stamp), 32) != hipSuccess) return D3D_EHIP;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fa_stamp), z, 32) != hipSuccess) return D3D_EHIP;
    }
    return D3D_OK;
}
#endif

// Same contract as d3d_flash_attention (attn_kernels.hip) plus `window` (0 = none; > 0: a query attends to the last `window` keys,
// itself included -- HF sliding-window masking, transformers 4.46 `_prepare_4d_causal_attention_mask_with_cache_position`).
int32_t d3d_flash_attention_v2(const void* qkv, void* out, int32_t B, int32_t S, int32_t H, int32_t head_dim, int64_t row_stride, int64_t batch_stride,
                               int32_t q_off, int32_t k_off, int32_t v_off, int32_t causal, int32_t seq_len, const int32_t* cu_seqlens, int32_t window,
                               int32_t dtype, void* stream) {
    if (B <= 0 || S <= 0) return D3D_OK;
    if ((head_dim != 64 && head_dim != 96) || (row_stride & 7) || (batch_stride & 7) || window < 0 || (window > 0 && !causal)) {
        d3d_set_error_("d3d_flash_attention_v2: head_dim must be 64 or 96; strides multiples of 8 elements; a window needs causal");
        return D3D_EINVAL;
    }
    const float sl2 = 1.4426950408889634f / sqrtf((float)head_dim);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)qkv;
    uint16_t* o = (uint16_t*)out;
    // Query rows per workgroup: 4 waves = 128 rows (two workgroups per CU), the default; 8 waves = 256 rows (one workgroup per CU, half the
    // K/V bytes staged per query) is selectable with D3D_ATTN_WAVES=8 and measures THE SAME time on the Phi-3 shapes and 8 % slower on the
    // ViT shape (577 rows = 2.25 blocks of 256): the kernel is not bound by the K/V loads.  Neither is it bound by instruction counts (a
    // software-pipelined and a hand-scheduled variant, the branch-free main body, row sums moved to the matrix pipe: all within 3 %).
    // In-kernel stamps put a wave's key tile at ~3 300 cycles on an otherwise empty SIMD and ~4 500 with two waves per SIMD, against
    // 768 cycles of matrix-pipe time: DESIGN.md section 4b.
    static const int force_nw = [] { const char* e = getenv("D3D_ATTN_WAVES"); return e ? atoi(e) : 0; }();
    const int nw = force_nw == 8 ? 8 : 4;
    const int BQ = nw * 32;
    const int nqb = (S + BQ - 1) / BQ;
    if (window >= S) window = 0;                                       // no query is further than S - 1 keys from the first key
    const int nx = causal ? (nqb + 1) / 2 : nqb;                        // causal: one workgroup per PAIR of query blocks (longest + shortest)
    dim3 grid((unsigned)((int64_t)nx * H * B)), block(nw * 64);
#define D3D_FA2N(BF, HDV, CA, NWV) hipLaunchKernelGGL((k_flash_attn32<BF, HDV, CA, NWV>), grid, block, 0, s, q, o, S, H, row_stride, batch_stride, q_off, k_off, v_off, \
                                                      sl2, seq_len, cu_seqlens, nqb, window, nx, B)
#define D3D_FA2(BF, HDV, CA) do { if (nw == 8) D3D_FA2N(BF, HDV, CA, 8); else D3D_FA2N(BF, HDV, CA, 4); } while (0)
    if (dtype == 0) {
        if (head_dim == 96) { if (causal) D3D_FA2(true, 96, true); else D3D_FA2(true, 96, false); }
        else { if (causal) D3D_FA2(true, 64, true); else D3D_FA2(true, 64, false); }
    } else {
        if (head_dim == 96) { if (causal) D3D_FA2(false, 96, true); else D3D_FA2(false, 96, false); }
        else { if (causal) D3D_FA2(false, 64, true); else D3D_FA2(false, 64, false); }
    }
#undef D3D_FA2N
#undef D3D_FA2
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
