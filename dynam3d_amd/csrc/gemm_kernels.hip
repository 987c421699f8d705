// gemm_kernels.hip -- bf16/fp16 "NT" GEMM with fused epilogues for the ViT / Phi-3 linears on gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        A, W row-major (K contiguous), fp32 accumulate on MFMA
//
// Every dense layer of the step has this shape (nn.Linear weights are [out,in]); both operands are
// K-contiguous, so A- and B-fragments of v_mfma_f32_16x16x32_{bf16,f16} are 16-byte K-vectors.
//
// Structure (cdna_hip_programming.md section 5, "step-3" structure):
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles),
//     BK = 64, two LDS buffers of 2 x 16 KiB (A,B)  -> 64 KiB / workgroup, 2 workgroups / CU.
//   * global -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR round trip).  The LDS image is
//     lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone in
//     the ds_read address (chunk ^= row & 7): conflict-free ds_read_b128 for both operands.
//   * operands are swapped in the MFMA (D = W_frag x A_frag = C^T tile) so each lane owns 4 CONSECUTIVE
//     output columns of one row -> 8-byte packed stores, bias/activation/residual fused on registers.
//   * XCD-aware workgroup remap: consecutive tiles of one XCD share A row-panels / W column-panels in
//     that XCD's private 4 MiB L2.
//
// Epilogues: none | +bias | +bias,QuickGELU | +bias,GELU(erf) | +residual | +bias+residual |
//            SwiGLU over interleaved gate/up column blocks (writes N/2 columns).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <vector>
#include <unordered_map>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;

#include "gemm_epilogue.h"

// Transposed epilogue of one wave's NI*16 x 64 sub-tile (256-tile kernel: NI = 8, 128-tile kernel: NI = 4).  In the accumulator layout a
// lane owns 4 neighbouring columns of one row, so a direct store instruction touches 16 rows x 32 B: ~8200 shader cycles per 256 x 256
// tile (cycle stamps, profiles/r02_gemm_cycle_stamps.txt: the same with ONE workgroup on the chip -- it is the CU's own store path, not
// the fabric).  Instead the wave packs its sub-tile to 16 bit, parks it in `reg` -- its own NI * 2 KiB of the (now idle) K-tile
// buffers: ds_write_b64, 16-byte chunks XOR-swizzled by row -- and reads it back row-major: ds_read_b128 + global_store_dwordx4,
// 8 rows x 128 B per instruction (SwiGLU: 16 rows x 64 B).  The residual is added on the way out, loaded with the same
// row-contiguous 16-byte pattern (same rounding points).  Needs ldc % 8 == 0 and 16-byte aligned C / residual; every wave of the
// workgroup must have passed a barrier behind its last read of the K-tile buffers.
template <bool BF16, int EPI, int NI, int NJ = 4>
__device__ __forceinline__ void epilogue_transposed(const float4v (&acc)[NI][NJ], char* reg, int lane, int row_base /* first row of the
                                                    sub-tile */, int col_base /* first (input) column of the sub-tile */, int M,
                                                    uint16_t* __restrict__ C, const uint16_t* __restrict__ bias,
                                                    const uint16_t* __restrict__ residual, int64_t ldc) {
    const int fi = lane & 15, fg = lane >> 4;
    constexpr int RB = (EPI == EPI_SWIGLU ? 16 : 32) * NJ;   // bytes per sub-tile row (NJ 16-column tiles; SwiGLU halves them)
    constexpr int LPR = RB / 16, RPP = 64 / LPR;         // lanes per row, rows per read-back pass
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = i * 16 + fi;
        const int m = row_base + row;
#pragma unroll
        for (int j = 0; j < NJ; j += (EPI == EPI_SWIGLU ? 2 : 1)) {
            const uint2 o = epi_pack<BF16, EPI, true>(acc[i][j], acc[i][EPI == EPI_SWIGLU ? j + 1 : j], bias, residual, m, col_base + j * 16, fg, ldc);
            const int cb = (EPI == EPI_SWIGLU ? j * 8 + fg * 4 : j * 16 + fg * 4) * 2;          // byte offset in the row
            *reinterpret_cast<uint2*>(reg + row * RB + ((((cb >> 4) ^ row) & (LPR - 1)) << 4) + (cb & 15)) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS operations execute in order: its own writes are visible to its reads
    const int ocol = (EPI == EPI_SWIGLU ? col_base / 2 : col_base) + (lane % LPR) * 8;
#pragma unroll
    for (int p = 0; p < NI * 16 / RPP; ++p) {
        const int row = p * RPP + lane / LPR;
        const int m = row_base + row;
        uint4 v = *reinterpret_cast<const uint4*>(reg + row * RB + ((((lane % LPR) ^ row) & (LPR - 1)) << 4));
        if (m >= M) continue;
        if constexpr (EPI == EPI_RES || EPI == EPI_BIAS_RES) {
            const uint4 r = *reinterpret_cast<const uint4*>(residual + (int64_t)m * ldc + ocol);
            v.x = add2_16<BF16>(v.x, r.x), v.y = add2_16<BF16>(v.y, r.y), v.z = add2_16<BF16>(v.z, r.z), v.w = add2_16<BF16>(v.w, r.w);
        }
        *reinterpret_cast<uint4*>(C + (int64_t)m * ldc + ocol) = v;
    }
}

// NSTAGE = LDS K-tile ring depth.  2 (64 KiB, two workgroups per CU) is the throughput configuration; 4 (128 KiB, one
// workgroup per CU) keeps three K tiles in flight for grids that cannot give every CU two workgroups anyway (the ViT
// projections at M = 2056: 136 tiles) -- there a K step is bounded by the LDS-DMA latency, not by its 32 MFMAs.
// BNT = tile width: 128 (wave tile 64 x 64, 2 workgroups per CU) or 64 (wave tile 64 x 32, 48 KiB of LDS at two stages -> 3 workgroups per
// CU, twice as many workgroups: the finer grid for GEMMs whose 128 x 128 grid is a small, badly divisible number of rounds -- the ViT
// projections at M = 4616).
template <bool BF16, int EPI, int NSTAGE, int BNT = 128>
__global__ void __launch_bounds__(NTHREADS, 2)
k_gemm_nt(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C,
          const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual, int M, int N, int K, int64_t lda,
          int64_t ldw, int64_t ldc, int tiles_m, int tiles_n, int wide_stores /* ldc % 8 == 0, 16-byte aligned C / residual */) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [2][A 128x64 | B 128x64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- XCD-aware tile mapping (bijective for any grid size) ------------------------------------------
    const int nwg = tiles_m * tiles_n;
    int wg = blockIdx.x;
    {
        const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    // grouped ordering: 8 M-tiles share consecutive N-tiles -> W panel reuse inside an XCD's L2
    constexpr int GM = 8;
    const int group = wg / (GM * tiles_n);
    const int gm0 = group * GM;
    const int gsz = min(GM, tiles_m - gm0);
    const int tm = gm0 + (wg % (GM * tiles_n)) % gsz;
    const int tn = (wg % (GM * tiles_n)) / gsz;
    const int row0 = tm * BM, col0 = tn * BNT;
    constexpr int NJ = BNT / 32;                            // 16-column MFMA tiles per wave: 4 (64 columns) or 2 (32 columns)

    const int wr = wave >> 1, wc = wave & 1;                // wave -> 64 x (BNT/2) sub-tile
    float4v acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    // LDS: buffer b -> A tile at smem + b*2*BM*BK, B tile right behind it
    const uint32_t lds0 = lds_addr_of(smem);
    TileLanes<4> ta, tw;
    ta.init(lda, row0, M, wave, lane);
    tw.init(ldw, col0, N, wave, lane);
    const uint16_t* abase = A + (int64_t)row0 * lda;
    const uint16_t* wbase = W + (int64_t)col0 * ldw;
    constexpr uint32_t STAGE_BYTES = (BM + BNT) * BK * 2;              // A tile + B tile
    constexpr int NPIECE = NJ * 2 + 4;                                 // LDS-DMA instructions per wave and K tile: 8 (BNT 128) / 6 (BNT 64)
    auto STAGE = [&](int t) {
        const uint32_t d = lds0 + (uint32_t)(t % NSTAGE) * STAGE_BYTES;
        stage_tile_dma<4>(ta, abase + t * BK, d, wave);
        if constexpr (BNT == 128) stage_tile_dma<4>(tw, wbase + t * BK, d + BM * BK * 2, wave);
        else stage_tile_dma2<4>(tw, wbase + t * BK, d + BM * BK * 2, wave);
    };
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
        if (t < nk) STAGE(t);

    const int fi = lane & 15, fg = lane >> 4;               // fragment row, k-group
    for (int t = 0; t < nk; ++t) {
        // tile t has landed when at most the (up to NSTAGE-2) younger tiles are still in flight: 8 LDS-DMA per wave and tile
        const int younger = min(t + NSTAGE - 2, nk - 1) - t;
        if (NSTAGE > 3 && younger >= 2) {
            if constexpr (NPIECE == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else if (NSTAGE > 2 && younger == 1) {
            if constexpr (NPIECE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                          // tile t visible to all; everyone is done reading tile t-1
        if (t + NSTAGE - 1 < nk) STAGE(t + NSTAGE - 1);         // ... whose buffer the new tile takes
        const uint16_t* la = smem + (t % NSTAGE) * ((BM + BNT) * BK);
        const uint16_t* lb = la + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wr * 64 + i * 16 + fi;
                af[i] = *reinterpret_cast<const uint4*>(la + ra * BK + (((kk * 4 + fg) ^ (ra & 7)) << 3));
                if (i < NJ) {
                    const int rb = wc * (BNT / 2) + i * 16 + fi;
                    bf[i] = *reinterpret_cast<const uint4*>(lb + rb * BK + (((kk * 4 + fg) ^ (rb & 7)) << 3));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<BF16>(bf[j], af[i], acc[i][j]);   // D = C^T tile: rows n, cols m
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- epilogue: lane holds, for tile (i,j): row m = row0+wr*64+i*16+fi, cols n = col0+wc*64+j*16+fg*4 .. +3
    if constexpr (EPI != EPI_LRELU_BWD) {
        if (wide_stores) {                                // (workgroup-uniform) through LDS: 16-byte row-contiguous stores, see epilogue_transposed
            __builtin_amdgcn_s_barrier();                 // every wave is done reading the last K tile
            epilogue_transposed<BF16, EPI, 4, NJ>(acc, reinterpret_cast<char*>(smem) + wave * (64 * (EPI == EPI_SWIGLU ? 16 : 32) * NJ), lane,
                                                  row0 + wr * 64, col0 + wc * (BNT / 2), M, C, bias, residual, ldc);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = row0 + wr * 64 + i * 16 + fi;
        if (m >= M) continue;
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < NJ; j += 2) store4<BF16, EPI>(acc[i][j], acc[i][j + 1], C, bias, residual, m, col0 + wc * (BNT / 2) + j * 16, fg, ldc);
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) store4<BF16, EPI>(acc[i][j], acc[i][j], C, bias, residual, m, col0 + wc * (BNT / 2) + j * 16, fg, ldc);
        }
    }
}

// ================================================================================================
// 256 x 256 x 64 tile, 8 waves (512 threads, one workgroup per CU), two wave GROUPS running half a
// K-tile apart so that on every SIMD one wave streams fragments LDS -> registers (and the whole
// workgroup's next tile HBM -> LDS) while its partner wave issues MFMAs back to back:
//
//     step k   : group 0: LOAD(half k)        group 1: COMPUTE(half k-1)      raw barrier
//     step k+1 : group 0: COMPUTE(half k)     group 1: LOAD(half k)           raw barrier
//
// A step handles one K-half (32 deep) of the wave's 128 x 64 sub-tile: LOAD = 12 conflict-free ds_read_b128
// (8 A + 4 B fragments, 48 VGPRs), COMPUTE = 32 x v_mfma_f32_16x16x32 from registers under s_setprio(1); four
// steps per 64-deep K-tile.  The LDS-DMA of tile t+1 is issued at the START of tile t's first step and only
// waited for at the END of its fourth (four MFMA phases in flight); raw s_barrier (not __syncthreads) keeps
// hipcc from draining it early.  Two 64 KiB LDS buffers (128 KiB), 128 accumulator VGPRs per wave.
// ================================================================================================
constexpr int TM = 256, TN = 256, T_THREADS = 512;

// SPLIT: the grid is `dp_tiles` whole tiles (a multiple of the CU count: full rounds, data parallel) followed by the TAIL tiles
// that would not fill a round, each cut into `splits` K-slices so that the partial last round lasts 1/splits of a tile
// instead of a whole one.  Every slice writes its fp32 accumulators to the workspace and leaves; k_splitk_fixup (the next launch)
// sums the slices in slice order -- deterministic -- and runs the fused epilogue.
// (Round 6 tried a persistent data-parallel tile walk on top of this kernel: +0.3 ... 1.1 %, but it spills -- parked with its evidence
//  under tools/experiments/gemm_persistent_walk/.  This kernel runs at the 256-register budget with ZERO spills in the step's
//  instantiations; tests/test_kernel_resources.py keeps it that way.)
template <bool BF16, int EPI, bool KFULL, bool SPLIT, int DMAV = 0>
__global__ void __launch_bounds__(T_THREADS, 2)
k_gemm_nt_256(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C,
              const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual, int M, int N, int K, int64_t lda,
              int64_t ldw, int64_t ldc, int tiles_m, int tiles_n, int dp_tiles, int splits, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];   // [2][A 256x64 | B 256x64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K / BK;
    // DMAV bit 4: timing build (tile codes 301-303, diagnostics only): every wave 0 reports shader-clock cycles (s_memtime) at kernel
    // start / K-loop start / K-loop end / kernel end and the constant 100 MHz counter at start / end into `ws` (6 x u64 per workgroup).
    constexpr bool TIMED = (DMAV & 16) != 0;
    constexpr int DV = DMAV & 15;
    unsigned long long tk[4] = {0, 0, 0, 0}, tr = 0;
    if constexpr (TIMED) {
        tk[0] = __builtin_readcyclecounter();
        tr = __builtin_amdgcn_s_memrealtime();
    }
    const int nwg = SPLIT ? dp_tiles : tiles_m * tiles_n;
    int wg = blockIdx.x;
    int kb = 0, ke = nk, slice = 0, tail_idx = -1;
    if (!SPLIT || wg < nwg) {
        const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    } else {
        // tail: slice-major, so that the workgroups running side by side work on the same K range of neighbouring tiles
        const int w = wg - dp_tiles, tail = tiles_m * tiles_n - dp_tiles;
        slice = __builtin_amdgcn_readfirstlane(w / tail);
        tail_idx = __builtin_amdgcn_readfirstlane(w - slice * tail);
        wg = dp_tiles + tail_idx;
        kb = __builtin_amdgcn_readfirstlane(slice * nk / splits);
        ke = __builtin_amdgcn_readfirstlane((slice + 1) * nk / splits);
    }
    // Tile order inside an XCD: GM consecutive M tiles share consecutive N tiles, so the 32 tiles one XCD runs side by side form a
    // GM x 32/GM block and pull GM A panels + 32/GM W panels per K step through that XCD's L2.  GM = 4 (4 + 8 = 12 panels for 32
    // tiles) is the production order; the unsplit launch can override it (`splits` carries GM there: D3D_GEMM_GM, an experiment knob
    // for the cross-XCD duplication measurement in DESIGN.md section 4: GM = 1 is 1 + 32 = 33 panels, 2.75 x the fabric reads).
    const int GM = SPLIT ? 4 : splits;
    const int group = wg / (GM * tiles_n);
    const int gm0 = group * GM;
    const int gsz = min(GM, tiles_m - gm0);
    const int tm = gm0 + (wg % (GM * tiles_n)) % gsz;
    const int tn = (wg % (GM * tiles_n)) / gsz;
    const int row0 = tm * TM, col0 = tn * TN;

    const int grp = wave >> 2;                 // wave group == M half of the tile; waves w and w+4 share a SIMD
    const int wn = wave & 3;
    const int fi = lane & 15, fg = lane >> 4;
    constexpr int BUF = 2 * TM * BK;           // elements per K-tile buffer (A then B)

    float4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    constexpr int NH = KFULL ? 2 : 1;          // K-halves held in registers per step
    uint4 af[NH][8], bf[NH][4];

    // LOAD(t, kk): the 12 fragments of K-half kk (32 deep) of tile t -> 48 VGPRs
    auto LOAD = [&](int t, int kk, int slot) {
        const uint16_t* la = smem + ((t - kb) & 1) * BUF;
        const uint16_t* lb = la + TM * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rb = wn * 64 + j * 16 + fi;
            bf[slot][j] = *reinterpret_cast<const uint4*>(lb + rb * BK + (((kk * 4 + fg) ^ (rb & 7)) << 3));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ra = grp * 128 + i * 16 + fi;
            af[slot][i] = *reinterpret_cast<const uint4*>(la + ra * BK + (((kk * 4 + fg) ^ (ra & 7)) << 3));
        }
    };
    auto COMPUTE = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<BF16>(bf[slot][j], af[slot][i], acc[i][j]);
    };

    const uint32_t lds0 = lds_addr_of(smem);
    TileLanes<8> ta, tw;
    ta.init(lda, row0, M, wave, lane);
    tw.init(ldw, col0, N, wave, lane);
    const uint16_t* abase = A + (int64_t)row0 * lda;
    const uint16_t* wbase = W + (int64_t)col0 * ldw;
    stage_tile_dma<8>(ta, abase + kb * BK, lds0, wave);
    stage_tile_dma<8>(tw, wbase + kb * BK, lds0 + TM * BK * 2, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ONE instruction stream for both groups: LOAD | barrier | COMPUTE | barrier ...; group 1 executes one extra
    // barrier up front, so it always runs exactly one step behind group 0 (and group 0 one extra at the end).
    // Tile t+1 goes in flight at the start of tile t and must have landed one step before group 0 reads it:
    // group 0 drains its share at the end of its last step of the tile, group 1 (a step late) one step earlier.
    if constexpr (TIMED) {
        tk[1] = __builtin_readcyclecounter();
    }
    if constexpr (DV >= 2) {
        // DMAV 2 ("interleaved"): both waves of a SIMD run the SAME software-pipelined stream -- no LOAD phase at all.  A phase is the
        // 32 MFMAs of one K-half out of fragment set `cs`, with the 12 ds_read_b128 of the NEXT K-half (set cs^1) issued one behind
        // every ~3rd MFMA, and, in the second phase of a tile, the 8 LDS-DMA pieces of tile t+2 in between as well:
        //
        //     P(t,0): MFMA(t, kk=0)  ||  reads (t, kk=1)                           | wait DMA(t+1) + own reads, ONE barrier per tile
        //     P(t,1): MFMA(t, kk=1)  ||  reads (t+1, kk=0)  ||  DMA(t+2) -> buffer of tile t (every wave is done with it)
        //
        // tools/probe/mfma_lds_probe.hip mode 5 is this loop without the DMA and the barrier: the matrix pipe is busy 97 % of the
        // time, against 80 % when the 24 reads of a tile are issued as one burst (mode 3) -- reads behind MFMAs are nearly free,
        // bursts are not (profiles/r02_mfma_lds_probe.txt).
        static_assert(KFULL, "DMAV 2 holds both K-halves in registers");
        auto PHASE = [&](const int cs, const uint16_t* la, const int kk, const bool rd, const bool dma, const uint16_t* asrc,
                         const uint16_t* wsrc, const uint32_t dst) {
            const int ns = cs ^ 1;
            const uint16_t* lb = la + TM * BK;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int i = m >> 2, j = m & 3;
                acc[i][j] = mfma16<BF16>(bf[cs][j], af[cs][i], acc[i][j]);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (rd) {
#pragma unroll
                    for (int r = 0; r < 12; ++r) {
                        if ((r * 8) / 3 != m) continue;
                        if (r < 4) {
                            const int rb = wn * 64 + r * 16 + fi;
                            bf[ns][r] = *reinterpret_cast<const uint4*>(lb + rb * BK + (((kk * 4 + fg) ^ (rb & 7)) << 3));
                        } else {
                            const int ra = grp * 128 + (r - 4) * 16 + fi;
                            af[ns][r - 4] = *reinterpret_cast<const uint4*>(la + ra * BK + (((kk * 4 + fg) ^ (ra & 7)) << 3));
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                if (dma && m % 3 == 1 && m / 3 < 8) {
                    const int p = m / 3;                       // pieces 0-3: A, 4-7: W
                    if (p < 4) stage_piece_dma(ta.off[p], asrc, dst + (uint32_t)p * 8192u);
                    else stage_piece_dma(tw.off[p - 4], wsrc, dst + TM * BK * 2 + (uint32_t)(p - 4) * 8192u);
                }
            }
        };
        auto TILE = [&](const int t, const bool rd, const bool dma) {
            const uint16_t* la = smem + ((t - kb) & 1) * BUF;
            const uint16_t* ln = smem + ((t + 1 - kb) & 1) * BUF;
            PHASE(0, la, 1, true, false, nullptr, nullptr, 0u);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (DMAV 3 / 4 are timing experiments with WRONG results: 3 re-reads the first K tile every step -- cache-hot operands --,
            //  4 issues no loads inside the loop at all: what is left is MFMA + ds_read + barrier.  DESIGN.md section 4.)
            const int ts = DV == 3 ? kb : t + 2;
            PHASE(1, ln, 0, rd, dma && DV != 4, abase + ts * BK, wbase + ts * BK, lds0 + ((t - kb) & 1) * (BUF * 2) + (uint32_t)wave * 1024u);
        };
        if (kb + 1 < ke) {
            stage_tile_dma<8>(ta, abase + (kb + 1) * BK, lds0 + BUF * 2, wave);
            stage_tile_dma<8>(tw, wbase + (kb + 1) * BK, lds0 + BUF * 2 + TM * BK * 2, wave);
        }
        LOAD(kb, 0, 0);
        int t = kb;
        for (; t < ke - 2; ++t) TILE(t, true, true);
        if (t < ke - 1) TILE(t++, true, false);
        TILE(t, false, false);
    } else if constexpr (DV == 1) {
        // DMAV 1: the A pieces of tile t+1 are issued in the LOAD phase, the W pieces between the MFMAs of the COMPUTE phase
        // (tile t+1 for group 0, tile t+2 for group 1, whose COMPUTE(t) runs beside group 0's LOAD(t+1)): the LOAD phase --
        // 8 LDS-DMA issues + 24 ds_read_b128 -- was longer than the partner's 64 MFMAs.
        static_assert(KFULL && !SPLIT, "DMAV 1 is built on the whole-K-tile, unsplit loop");
        if (grp == 1) {
            if (kb + 1 < ke) stage_tile_dma<8>(tw, wbase + (kb + 1) * BK, lds0 + (BUF * 2) + TM * BK * 2, wave);
            __builtin_amdgcn_s_barrier();
        }
        for (int t = kb; t < ke; ++t) {
            if (t + 1 < ke) stage_tile_dma<8>(ta, abase + (t + 1) * BK, lds0 + ((t + 1 - kb) & 1) * (BUF * 2), wave);
            LOAD(t, 0, 0);
            LOAD(t, 1, 1);
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            const int tw_t = t + 1 + grp;
            const bool dma = tw_t < ke;
            const uint16_t* wsrc = wbase + tw_t * BK;
            const uint32_t wdst = lds0 + ((tw_t - kb) & 1) * (BUF * 2) + TM * BK * 2 + (uint32_t)wave * 1024u;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<BF16>(bf[0][j], af[0][i], acc[i][j]);
                if ((i & 1) == 0 && dma) stage_piece_dma(tw.off[i >> 1], wsrc, wdst + (uint32_t)(i >> 1) * 8192u);
            }
            COMPUTE(1);
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int t = kb; t < ke; ++t) {
        if (t + 1 < ke) {
            const uint32_t nxt = lds0 + ((t + 1 - kb) & 1) * (BUF * 2);
            stage_tile_dma<8>(ta, abase + (t + 1) * BK, nxt, wave);
            stage_tile_dma<8>(tw, wbase + (t + 1) * BK, nxt + TM * BK * 2, wave);
        }
        if constexpr (KFULL) {
            LOAD(t, 0, 0);
            LOAD(t, 1, 1);
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            COMPUTE(0);
            COMPUTE(1);
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            LOAD(t, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            COMPUTE(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            LOAD(t, 1, 0);
            if (grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            COMPUTE(0);
            __builtin_amdgcn_s_setprio(0);
            if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    }
    if (DV < 2 && grp == 0) __builtin_amdgcn_s_barrier();
    if constexpr (TIMED) {
        tk[2] = __builtin_readcyclecounter();
    }

    if constexpr (DV == 5) {
        if (K != 7) return;                    // timing experiment: no epilogue at all (the stores stay reachable, so the loop is kept)
    }
    if constexpr (SPLIT) {
        if (tail_idx >= 0) {
            // A K-slice of a tail tile: its fp32 accumulators go to the workspace slot (tile, slice) -- entry e = i*4 + j holds
            // acc[i][j] of all 512 threads, 16 bytes each -- and the workgroup LEAVES.  k_splitk_fixup, the next launch on the
            // stream, sums the slices in slice order and runs the epilogue: no workgroup ever waits for another one, so the
            // scheme needs no co-residency and cannot deadlock however the chip is shared (other streams, processes, CU masks).
            float4v* slot = reinterpret_cast<float4v*>(ws) + ((int64_t)tail_idx * splits + slice) * (32 * T_THREADS) + tid;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(acc[i][j], slot + (i * 4 + j) * T_THREADS);
            return;
        }
    }

    if constexpr (EPI == EPI_LRELU_BWD) {          // (needs the activation in the accumulator layout: direct 8-byte stores; training only)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = row0 + grp * 128 + i * 16 + fi;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) store4<BF16, EPI>(acc[i][j], acc[i][j], C, bias, residual, m, col0 + wn * 64 + j * 16, fg, ldc);
        }
    } else {
        __builtin_amdgcn_s_barrier();              // every wave is done reading the K-tile buffers
        epilogue_transposed<BF16, EPI, 8>(acc, reinterpret_cast<char*>(smem) + wave * (128 * (EPI == EPI_SWIGLU ? 64 : 128)), lane,
                                          row0 + grp * 128, col0 + wn * 64, M, C, bias, residual, ldc);
    }
    if constexpr (TIMED) {
        tk[3] = __builtin_readcyclecounter();
        const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(ws) + (size_t)blockIdx.x * 6;
            o[0] = tk[0]; o[1] = tk[1]; o[2] = tk[2]; o[3] = tk[3]; o[4] = tr; o[5] = tr1;
        }
    }
}

// K-loop of the 256 x 256 kernels picked by d3d_gemm_nt: the interleaved loop (tile codes 260 / 264) unless D3D_GEMM_LOOP=0 asks for
// the staggered wave-group loop (257 / 258) -- an A/B knob; both are parity-tested.
inline bool gemm_loop_interleaved() {
    static const bool v = [] { const char* e = getenv("D3D_GEMM_LOOP"); return !(e && e[0] == '0'); }();
    return v;
}

inline int tile_group_m() {
    static const int gm = [] {
        const char* e = getenv("D3D_GEMM_GM");
        const int v = e ? atoi(e) : 4;
        return v >= 1 && v <= 64 ? v : 4;
    }();
    return gm;
}

template <bool BF16, int EPI, bool KFULL, int DMAV = 0>
int32_t launch256(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
                  int64_t ldw, int64_t ldc, hipStream_t s) {
    const int tm = (M + TM - 1) / TM, tn = N / TN;
    const size_t sh = 2 * 2 * TM * BK * sizeof(uint16_t);   // 128 KiB
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt_256<BF16, EPI, KFULL, false, DMAV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    });
    D3D_HIP(attr_err);
    float* dbg = nullptr;
    if constexpr ((DMAV & 16) != 0) D3D_HIP(hipMalloc(&dbg, (size_t)tm * tn * 6 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((k_gemm_nt_256<BF16, EPI, KFULL, false, DMAV>), dim3(tm * tn), dim3(T_THREADS), sh, s, (const uint16_t*)A, (const uint16_t*)W,
                       (uint16_t*)C, (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, tm, tn, 0, tile_group_m(), dbg);
    if constexpr ((DMAV & 16) != 0) {
        // diagnostics: per-workgroup cycle stamps -> one line on stderr (mean over workgroups)
        D3D_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)tm * tn * 6);
        D3D_HIP(hipMemcpy(h.data(), dbg, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
        D3D_HIP(hipFree(dbg));
        double pre = 0, loop = 0, epi = 0, ghz = 0;
        for (int i = 0; i < tm * tn; ++i) {
            const unsigned long long* o = &h[(size_t)i * 6];
            pre += (double)(o[1] - o[0]), loop += (double)(o[2] - o[1]), epi += (double)(o[3] - o[2]);
            ghz += (double)(o[3] - o[0]) / ((double)(o[5] - o[4]) * 10.0);
        }
        const double n = tm * tn;
        fprintf(stderr, "[d3d gemm timing] variant %d  M %d N %d K %d: workgroups %d, shader clock %.3f GHz, cycles per workgroup: prologue %.0f, K loop %.0f (%.1f per K tile; 128 MFMAs per SIMD = 2061 pipe cycles), epilogue %.0f\n",
                DMAV & 15, M, N, K, tm * tn, ghz / n, pre / n, loop / n, loop / n / (K / BK), epi / n);
    }
    D3D_LAUNCH_CHECK();
}

// Fix-up of the split-K tail: workgroup (tile, i) sums entries e = i*4 .. i*4+3 of the tile's `splits` slots in slice order and
// runs the GEMM's own epilogue on them (same thread -> element mapping as k_gemm_nt_256: thread tid of the GEMM held these values).
template <bool BF16, int EPI>
__global__ void __launch_bounds__(T_THREADS)
k_splitk_fixup(const float* __restrict__ ws, uint16_t* __restrict__ C, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual,
               int M, int64_t ldc, int tiles_m, int tiles_n, int dp_tiles, int splits) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tail_idx = blockIdx.x >> 3, i = blockIdx.x & 7;
    const int wg = dp_tiles + tail_idx;
    constexpr int GM = 4;                                       // the tile order of k_gemm_nt_256
    const int group = wg / (GM * tiles_n);
    const int gm0 = group * GM;
    const int gsz = min(GM, tiles_m - gm0);
    const int tm = gm0 + (wg % (GM * tiles_n)) % gsz;
    const int tn = (wg % (GM * tiles_n)) / gsz;
    const int row0 = tm * TM, col0 = tn * TN;
    const int grp = wave >> 2, wn = wave & 3, fi = lane & 15, fg = lane >> 4;
    const float4v* base = reinterpret_cast<const float4v*>(ws) + (int64_t)tail_idx * splits * (32 * T_THREADS) + (i * 4) * T_THREADS + tid;
    float4v sum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sum[j] = __builtin_nontemporal_load(base + j * T_THREADS);
    for (int sl = 1; sl < splits; ++sl) {
        const float4v* p = base + (int64_t)sl * (32 * T_THREADS);
        float4v v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(p + j * T_THREADS);
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[j] += v[j];
    }
    const int m = row0 + grp * 128 + i * 16 + fi;
    if (m >= M) return;
    if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int j = 0; j < 4; j += 2) store4<BF16, EPI>(sum[j], sum[j + 1], C, bias, residual, m, col0 + wn * 64 + j * 16, fg, ldc);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) store4<BF16, EPI>(sum[j], sum[j], C, bias, residual, m, col0 + wn * 64 + j * 16, fg, ldc);
    }
}

// Per-stream workspace of the split-K tail: 256 KiB of fp32 per (tail tile, slice), at most one round of them.  Streams never
// share a workspace (a GEMM and its fix-up are ordered by the stream; two streams may run split GEMMs concurrently).  Allocated on
// first use -- or ahead of time with d3d_gemm_reserve_workspace(stream), which is what a caller that captures the stream into a
// hipGraph must do (no hipMalloc inside a capture).
struct SplitWorkspace {
    float* ws = nullptr;
    int slots = 0;
};

int32_t split_workspace(hipStream_t s, int slots, SplitWorkspace** out) {
    static std::mutex mu;
    static std::unordered_map<hipStream_t, SplitWorkspace> table;
    std::lock_guard<std::mutex> lock(mu);
    SplitWorkspace& w = table[s];
    if (w.slots < slots) {
        if (w.ws) (void)hipFree(w.ws);
        w.ws = nullptr;
        w.slots = 0;
        D3D_HIP(hipMalloc(&w.ws, (size_t)slots * 32 * T_THREADS * sizeof(float4v)));
        w.slots = slots;
    }
    *out = &w;
    return D3D_OK;
}

int cu_count() {
    // per device: a process may drive several GPUs
    static std::mutex mu;
    static std::unordered_map<int, int> table;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(dev);
    if (it != table.end()) return it->second;
    int n = 0;
    const int cus = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    table[dev] = cus;
    return cus;
}

// Split plan for M x N output tiles of 256 x 256: full rounds stay whole; the remainder is cut `splits` ways if that fits one round.
inline void split_plan(int tiles, int nk, int* dp_tiles, int* splits) {
    const int P = cu_count();
    const int tail = tiles % P;
    int sp = tail ? P / tail : 1;
    if (sp > 8) sp = 8;
    if (sp > nk / 8) sp = nk / 8;          // a slice keeps at least 8 K steps: shorter ones are all pipeline fill and fix-up
    *dp_tiles = tiles - tail;
    *splits = sp < 1 ? 1 : sp;
}

template <bool BF16, int EPI, int DMAV = 0>
int32_t launch256_split(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
                        int64_t ldw, int64_t ldc, hipStream_t s) {
    const int tm = (M + TM - 1) / TM, tn = N / TN, tiles = tm * tn;
    int dp_tiles, splits;
    split_plan(tiles, K / BK, &dp_tiles, &splits);
    if (splits < 2) return launch256<BF16, EPI, true, DMAV>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s);
    const int tail = tiles - dp_tiles;
    SplitWorkspace* w = nullptr;
    int32_t rc = split_workspace(s, cu_count(), &w);
    if (rc != D3D_OK) return rc;
    const size_t sh = 2 * 2 * TM * BK * sizeof(uint16_t);   // 128 KiB
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt_256<BF16, EPI, true, true, DMAV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    });
    D3D_HIP(attr_err);
    hipLaunchKernelGGL((k_gemm_nt_256<BF16, EPI, true, true, DMAV>), dim3(dp_tiles + tail * splits), dim3(T_THREADS), sh, s, (const uint16_t*)A,
                       (const uint16_t*)W, (uint16_t*)C, (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, tm, tn, dp_tiles,
                       splits, w->ws);
    hipLaunchKernelGGL((k_splitk_fixup<BF16, EPI>), dim3(tail * 8), dim3(T_THREADS), 0, s, (const float*)w->ws, (uint16_t*)C, (const uint16_t*)bias,
                       (const uint16_t*)res, M, ldc, tm, tn, dp_tiles, splits);
    D3D_LAUNCH_CHECK();
}

template <bool BF16, int EPI, int NSTAGE, int BNT = 128>
int32_t launch_stages(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
                      int64_t ldw, int64_t ldc, hipStream_t s) {
    const int tm = (M + BM - 1) / BM, tn = N / BNT;
    const size_t sh = (size_t)NSTAGE * (BM + BNT) * BK * sizeof(uint16_t);
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt<BF16, EPI, NSTAGE, BNT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    });
    D3D_HIP(attr_err);
    hipLaunchKernelGGL((k_gemm_nt<BF16, EPI, NSTAGE, BNT>), dim3(tm * tn), dim3(NTHREADS), sh, s, (const uint16_t*)A, (const uint16_t*)W, (uint16_t*)C,
                       (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, tm, tn,
                       (int)((ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (!res || ((uintptr_t)res & 15) == 0)));
    D3D_LAUNCH_CHECK();
}

// 128 x 128 tile: ring depth by grid size (see k_gemm_nt)
template <bool BF16, int EPI>
int32_t launch(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
               int64_t ldw, int64_t ldc, hipStream_t s, int stages = 0) {
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    if (stages == 64) return launch_stages<BF16, EPI, 2, 64>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s);      // 128 x 64 tiles, 3 workgroups per CU
    // between one and one and a half workgroups per CU (ViT out-proj / fc2 at M = 4616: 296 tiles for 256 CUs): the 128 x 64 tile doubles
    // the grid and fits three workgroups per CU -- 22.5 -> 20.3 us (K = 1024), 61.0 -> 57.7 us (K = 4096); larger grids lose with it
    if (stages == 0 && tiles > cu_count() && tiles * 2 <= cu_count() * 3) return launch_stages<BF16, EPI, 2, 64>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s);
    if (stages == 0) stages = (tiles <= cu_count() && K / BK >= 4) ? 4 : 2;
    return stages == 4 ? launch_stages<BF16, EPI, 4>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s)
                       : launch_stages<BF16, EPI, 2>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s);
}

// ================================================================================================
// Skinny GEMM for M <= 16 rows (KV-cache decode: 8 sequences x one token): pure weight streaming, bounded by HBM.
//
// One WORKGROUP owns 32 output columns; its 4 / 8 / 16 waves split K.  Per 32-deep step a wave loads two 16x32 W fragments
// straight from global memory in MFMA operand layout (16 bytes per lane; no LDS staging -- a weight element is used once) plus
// the matching 16x32 x fragment (tiny, cache resident) and issues two MFMAs; the waves' accumulators meet in LDS and wave 0
// sums them in wave order (deterministic) and runs the epilogue.  (A first version cut K across workgroups with fp32 partials
// in global memory and an arrival counter: 6 us of fix-up on a 12 us o_proj.)  SwiGLU uses the per-16 interleaved gate/up
// rows: the two fragments ARE a gate tile and its up tile.
// (Folding the preceding RMSNorm into the x fragments was built and measured -- bit-identical, but the per-step gain loads and
// the row-statistics prologue cost 13-31 us per projection against 9 us for the separate norm launch; dropped.)
// ================================================================================================
// HALF: 16 instead of 32 columns per workgroup (twice as many workgroups: the narrow o_proj / down_proj then cover 192 CUs, not 96)
// NORM: X is the RAW residual stream and the kernel applies HF Phi3RMSNorm (gain `nw`, float32) to it -- every workgroup normalises
// the <= 16 rows once into LDS (wave per row, the lane / chunk order of k_norm: bit-identical to d3d_norm) while its first weight
// fragments are in flight, and the K loop reads its activation fragments from there.  Saves the d3d_norm launch in front of the
// qkv / gate_up / lm_head projections of a decode token (64 + 1 launches of ~7 us each per token).
// NT: the weight stream loaded non-temporally (global_load_dwordx4 ... nt; MI355X_MICROARCH.md "nt-weights").  A knob, off by default: see launch_skinny.
template <bool NT>
__device__ __forceinline__ uint4 ld_weight(const uint16_t* p) {
    if constexpr (NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}

template <bool BF16, int EPI, bool HALF, bool NORM = false, bool NT = false>
__global__ void __launch_bounds__(1024)
k_gemm_skinny(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W, uint16_t* __restrict__ C, const uint16_t* __restrict__ bias,
              const uint16_t* __restrict__ residual, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldc,
              const float* __restrict__ nw = nullptr, float eps = 0.f) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];      // [NW][2][64] float4
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int ct = blockIdx.x;
    const int nsteps = K / 32;
    const int s0 = (int)((int64_t)wave * nsteps / NW), s1 = (int)((int64_t)(wave + 1) * nsteps / NW);
    float4v* red = reinterpret_cast<float4v*>(sk_lds);
    constexpr int COLS = HALF ? 16 : 32;
    const uint16_t* w0 = W + (int64_t)(ct * COLS + fi) * ldw + fg * 8;
    const uint16_t* w1 = HALF ? w0 : w0 + 16 * ldw;
    uint4 a[4], b[4];
    int st = s0;
    const bool first = st + 4 <= s1;
    if (first) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = ld_weight<NT>(w0 + (st + u) * 32);
            if constexpr (!HALF) b[u] = ld_weight<NT>(w1 + (st + u) * 32);
        }
    }
    const int xrow = fi < M ? fi : M - 1;                                           // rows >= M: a duplicate, never stored
    const uint16_t* xr = X + (int64_t)xrow * ldx + fg * 8;
    // NORM: the <= 16 rows are normalised ONCE per workgroup into LDS (x_hat stored 16-bit, times the gain, stored 16-bit: HF Phi3RMSNorm;
    // statistics in k_norm's lane / chunk order: bit-identical to d3d_norm) while the first weight fragments are in flight; the K loop then
    // reads its activation fragments from LDS and does nothing but stream weights.  (Round 2's version normalised every fragment inside
    // the K loop -- gain loads + 16 conversions per step on the path that should only wait for weights: slower than the separate launch.)
    const int XS = K + 8;                                                            // LDS row stride in elements (16 B pad: rows 4 banks apart)
    uint16_t* xs = reinterpret_cast<uint16_t*>(sk_lds);       // shares the LDS with the wave-reduction buffer (used only after the K loop):
                                                              // 49 KB at 8 rows instead of 81 -- two 16-wave workgroups per CU either way
    if constexpr (NORM) {
        for (int r = wave; r < M; r += NW) {
            const uint16_t* row = X + (int64_t)r * ldx;
            float ss = 0.f;
            for (int c = 0; c < K / 512; ++c) {
                const uint4 raw = *reinterpret_cast<const uint4*>(row + c * 512 + lane * 8);
                const uint16_t* h = reinterpret_cast<const uint16_t*>(&raw);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = to_f32<BF16>(h[j]);
                    ss += v * v;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            const float rstd = rsqrtf(ss / (float)K + eps);
            for (int c = 0; c < K / 512; ++c) {
                const int off = c * 512 + lane * 8;
                const uint4 raw = *reinterpret_cast<const uint4*>(row + off);
                const float4 g0 = *reinterpret_cast<const float4*>(nw + off), g1 = *reinterpret_cast<const float4*>(nw + off + 4);
                const float ww[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const uint16_t* h = reinterpret_cast<const uint16_t*>(&raw);
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float a = to_f32<BF16>(h[j]) * rstd, b = to_f32<BF16>(h[j + 1]) * rstd;
                    r16x2<BF16>(a, b);
                    o[j >> 1] = pack2<BF16>(a * ww[j], b * ww[j + 1]);
                }
                *reinterpret_cast<uint4*>(xs + r * XS + off) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
    }
    auto xfrag = [&](int stp) -> uint4 {
        if constexpr (NORM) return *reinterpret_cast<const uint4*>(xs + xrow * XS + stp * 32 + fg * 8);
        else return *reinterpret_cast<const uint4*>(xr + stp * 32);
    };
    float4v acc0 = float4v{0.f, 0.f, 0.f, 0.f}, acc1 = float4v{0.f, 0.f, 0.f, 0.f};
    if (first) {
        for (;;) {                                             // software pipeline: batch i+1's weights load under batch i's MFMAs
            uint4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = xfrag(st + u);
            const bool more = st + 8 <= s1;
            uint4 an[4], bn[4];
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    an[u] = ld_weight<NT>(w0 + (st + 4 + u) * 32);
                    if constexpr (!HALF) bn[u] = ld_weight<NT>(w1 + (st + 4 + u) * 32);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = mfma16<BF16>(a[u], x[u], acc0);
                if constexpr (!HALF) acc1 = mfma16<BF16>(b[u], x[u], acc1);
            }
            st += 4;
            if (!more) break;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = an[u];
                if constexpr (!HALF) b[u] = bn[u];
            }
        }
    }
    for (; st < s1; ++st) {
        const uint4 aw = ld_weight<NT>(w0 + st * 32);
        const uint4 x = xfrag(st);
        acc0 = mfma16<BF16>(aw, x, acc0);
        if constexpr (!HALF) {
            const uint4 bw = ld_weight<NT>(w1 + st * 32);
            acc1 = mfma16<BF16>(bw, x, acc1);
        }
    }
    if constexpr (NORM) __syncthreads();                       // every wave is done reading the normalised rows the buffer below overlays
    red[(wave * 2 + 0) * 64 + lane] = acc0;
    red[(wave * 2 + 1) * 64 + lane] = acc1;
    __syncthreads();
    if (wave != 0 || fi >= M) return;
    acc0 = red[lane];
    acc1 = red[64 + lane];
    for (int q = 1; q < NW; ++q) {                             // wave order: deterministic
        acc0 += red[(q * 2 + 0) * 64 + lane];
        acc1 += red[(q * 2 + 1) * 64 + lane];
    }
    if constexpr (EPI == EPI_SWIGLU) {
        store4<BF16, EPI>(acc0, acc1, C, bias, residual, fi, ct * 32, fg, ldc);
    } else {
        store4<BF16, EPI>(acc0, acc0, C, bias, residual, fi, ct * COLS, fg, ldc);
        if constexpr (!HALF) store4<BF16, EPI>(acc1, acc1, C, bias, residual, fi, ct * 32 + 16, fg, ldc);
    }
}

template <bool BF16, int EPI, bool NORM = false>
int32_t launch_skinny(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
                      int64_t ldw, int64_t ldc, hipStream_t s, const float* nw = nullptr, float eps = 0.f) {
    const int nsteps = K / 32;
    const bool half = EPI != EPI_SWIGLU && N / 32 < 2 * cu_count();   // fewer than two 32-column tiles per CU: 16-column tiles (measured:
                                                                     //   qkv 2.6 -> 3.0 TB/s, down_proj 1.9 -> 2.8 TB/s)
    const int ntiles = half ? N / 16 : N / 32;
    int nwv = ntiles <= 256 ? 16 : (ntiles <= 512 ? 8 : 4);           // ~2000-4000 waves on the chip, K / 32 / nw steps each
    while (nwv > 1 && nsteps / nwv < 2) nwv >>= 1;
    const size_t sh_red = (size_t)nwv * 2 * 64 * 16, sh_x = NORM ? (size_t)M * (K + 8) * 2 : 0;
    const size_t sh = sh_red > sh_x ? sh_red : sh_x;           // wave reduction buffer, overlaid on the normalised rows
    if constexpr (NORM) {
        if (sh > 160 * 1024) {
            d3d_set_error_("d3d_gemm_nt_rmsnorm: rows x K does not fit the LDS");
            return D3D_EINVAL;
        }
        static std::once_flag attr_once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(attr_once, [&] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<BF16, EPI, false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (attr_err == hipSuccess)
                attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<BF16, EPI, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if constexpr (EPI != EPI_SWIGLU) {
                if (attr_err == hipSuccess)
                    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<BF16, EPI, true, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (attr_err == hipSuccess)
                    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny<BF16, EPI, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
        });
        D3D_HIP(attr_err);
    }
    // non-temporal weight loads: OFF by default -- measured SLOWER here (profiles/r05_decode.txt: 3.11 -> 3.30 ms per token; this kernel's
    // 16-row x 64-byte wave loads are not the guide's 1 KiB LDS-DMA stream).  D3D_SKINNY_NT=1 selects them (read per call: bench_decode.py).
    const char* nte = getenv("D3D_SKINNY_NT");
    const bool nt = nte && nte[0] == '1';
    // (Round 6 tried a deep-prefetch version -- a wave requests its whole K range, 8-16 steps, before the activation rows are staged: bit-identical,
    //  NOT faster (qkv 21.3 -> 20.6 us, down_proj 15.5 -> 18.3 us, token 3.10 -> 3.20 ms): the per-launch cost here is fixed ramp / prologue /
    //  drain, ~8 us, not the depth of the stream -- whose marginal rate is already 6.2 TB/s.  tools/experiments/skinny_deep/.)
#define D3D_SKINNY_LAUNCH(HALFV, NTV)                                                                                                    \
    hipLaunchKernelGGL((k_gemm_skinny<BF16, EPI, HALFV, NORM, NTV>), dim3(ntiles), dim3(nwv * 64), sh, s, (const uint16_t*)A, (const uint16_t*)W, \
                       (uint16_t*)C, (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, nw, eps)
    if (half) {
        if constexpr (EPI != EPI_SWIGLU) {
            if (nt) D3D_SKINNY_LAUNCH(true, true); else D3D_SKINNY_LAUNCH(true, false);
        }
    } else {
        if (nt) D3D_SKINNY_LAUNCH(false, true); else D3D_SKINNY_LAUNCH(false, false);
    }
#undef D3D_SKINNY_LAUNCH
    D3D_LAUNCH_CHECK();
}

int32_t skinny_dispatch(const void* A, const void* W, void* C, const void* bias, const void* residual, int32_t M, int32_t N, int32_t K,
                        int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, void* stream) {
    if (M <= 0) return D3D_OK;
    if (M > 16 || N % 32 != 0 || K % 32 != 0 || (lda & 7) || (ldw & 7) || (ldc & 3)) {
        d3d_set_error_("skinny GEMM: needs M <= 16, N % 32 == 0, K % 32 == 0, lda/ldw % 8 == 0, ldc % 4 == 0");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
#define D3D_SKINNY_CASE(E)                                                                                                    \
    case E:                                                                                                                   \
        return dtype == 0 ? launch_skinny<true, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)                         \
                          : launch_skinny<false, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);
    switch (epilogue) {
        D3D_SKINNY_CASE(EPI_NONE)
        D3D_SKINNY_CASE(EPI_BIAS)
        D3D_SKINNY_CASE(EPI_RES)
        D3D_SKINNY_CASE(EPI_SWIGLU)
    }
#undef D3D_SKINNY_CASE
    d3d_set_error_("skinny GEMM: supported epilogues are none, bias, residual, SwiGLU");
    return D3D_EINVAL;
}

}  // namespace

extern "C" {

// C[M,N] (or [M,N/2] for SwiGLU) = epi(A[M,K] W[N,K]^T).  dtype: 0 = bf16, 1 = fp16.  Requires N % 128 == 0, K % 64 == 0,
// 16-byte aligned rows (lda, ldw multiples of 8).  M is arbitrary (edge tiles are masked).
int32_t d3d_gemm_nt_tile(const void* A, const void* W, void* C, const void* bias, const void* residual, int32_t M, int32_t N, int32_t K,
                         int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, int32_t tile, void* stream);

// RMSNorm (HF Phi3RMSNorm, float32 gain) fused into the weight-streaming GEMM of <= 16 rows: C = epi(RMSNorm(A) W^T), epilogue 0 or 6.
int32_t d3d_gemm_nt_rmsnorm(const void* A, const float* norm_w, float eps, const void* W, void* C, int32_t M, int32_t N, int32_t K, int64_t lda,
                            int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, void* stream) {
    if (M <= 0) return D3D_OK;
    if (M > 16 || N % 32 != 0 || K % 512 != 0 || (lda & 7) || (ldw & 7) || (ldc & 3) || !norm_w || (epilogue != EPI_NONE && epilogue != EPI_SWIGLU)) {
        d3d_set_error_("d3d_gemm_nt_rmsnorm: needs M <= 16, N % 32 == 0, K % 512 == 0, lda/ldw % 8 == 0, ldc % 4 == 0, epilogue 0 or 6");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_NONE)
        return dtype == 0 ? launch_skinny<true, EPI_NONE, true>(A, W, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, s, norm_w, eps)
                          : launch_skinny<false, EPI_NONE, true>(A, W, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, s, norm_w, eps);
    return dtype == 0 ? launch_skinny<true, EPI_SWIGLU, true>(A, W, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, s, norm_w, eps)
                      : launch_skinny<false, EPI_SWIGLU, true>(A, W, C, nullptr, nullptr, M, N, K, lda, ldw, ldc, s, norm_w, eps);
}

int32_t d3d_gemm_reserve_workspace(void* stream) {
    SplitWorkspace* w = nullptr;
    return split_workspace((hipStream_t)stream, cu_count(), &w);
}

int32_t d3d_gemm_nt(const void* A, const void* W, void* C, const void* bias, const void* residual, int32_t M, int32_t N, int32_t K,
                    int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, void* stream) {
    // Tile choice.  The 256x256 staggered kernel runs one workgroup per CU, so it pays when its grid covers the CUs for
    // several rounds; its M remainder (M % 256 rows) goes to the 128x128 kernel instead of a mostly-empty row of 256-tiles
    // (at M = 7200 that turns 29 x N/256 workgroups = 7.25 rounds into 28 x N/256 = exactly 7 for N = 16384).
    // Between one and three rounds the partial last round decides: it is K-split when it can be cut at least three ways
    // (o_proj / down_proj at M = 6400: 300 tiles = one round + 44 tiles x 5 slices; measured 0.120 / 0.277 ms against
    // 0.132 / 0.306 ms for the 128x128 kernel and 0.148 / 0.330 ms unsplit); otherwise the finer 128x128 grid wins.
    if (M <= 16 && N % 32 == 0 && K % 32 == 0 && (epilogue == EPI_NONE || epilogue == EPI_RES || epilogue == EPI_SWIGLU || epilogue == EPI_BIAS))
        return d3d_gemm_nt_tile(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, dtype, epilogue, 16, stream);   // weight streaming
    // A partial last row of 256-tiles rides in the 256 launch when it does not add a round (the kernel clamps its loads and
    // masks its stores): ViT qkv at M = 4616 is 228 tiles in one round, 41 us, against 216 tiles + a remainder launch, 52 us.
    int64_t tm256 = M / TM;
    if (M % TM && tm256 > 0 && N % TN == 0) {
        const int64_t P = cu_count(), tn = N / TN;
        if (((tm256 + 1) * tn + P - 1) / P == (tm256 * tn + P - 1) / P) tm256 += 1;
    }
    const int64_t rows256 = tm256 * TM < M ? tm256 * TM : M;
    const int64_t blocks256 = tm256 * (N / TN);
    int tile = 128;
    // (the 256-tile kernels store 16 bytes per lane: C / residual rows must be 16-byte aligned)
    const bool ok256 = N % TN == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (!residual || ((uintptr_t)residual & 15) == 0);
    if (ok256 && (blocks256 >= 768 || (blocks256 <= cu_count() && blocks256 * 4 >= cu_count() * 3))) {
        tile = 257;               // several rounds, or one nearly full round (ViT qkv at M = 4616: 216 tiles, 41.6 us against 46.3 us)
    } else if (ok256 && blocks256 >= 256) {
        int dp_tiles, splits;
        split_plan((int)blocks256, K / BK, &dp_tiles, &splits);
        if (splits >= 3) tile = 258;
    }
    // A half-empty last round of a many-round 256-tile grid goes to the 128 x 128 kernel when it is made of whole rows of tiles:
    // gate_up at M = 6656 is 26 x 64 = 6.5 rounds -- 24 row tiles fill 6 rounds exactly, the last 512 rows are 4 x 128 = 512
    // workgroups of the 128-kernel = one round of ITS 512 slots (49 us instead of a 76 us round that idles half the CUs).
    static const bool tail128 = [] { const char* e = getenv("D3D_GEMM_TAIL128"); return !(e && e[0] == '0'); }();
    if (tile == 257 && tail128 && rows256 == M && M % TM == 0) {
        const int64_t P = cu_count(), tn = N / TN, tiles = tm256 * tn, R = tiles % P;
        if (R > 0 && R * 2 <= P && R % tn == 0 && tm256 > R / tn) {
            const int64_t m1 = (tm256 - R / tn) * TM;
            int32_t rc = d3d_gemm_nt_tile(A, W, C, bias, residual, (int32_t)m1, N, K, lda, ldw, ldc, dtype, epilogue, gemm_loop_interleaved() ? 260 : 257, stream);
            if (rc != D3D_OK) return rc;
            const char* a8 = (const char*)A + m1 * lda * 2;
            char* c8 = (char*)C + m1 * ldc * 2;
            const char* r8 = residual ? (const char*)residual + m1 * ldc * 2 : nullptr;
            return d3d_gemm_nt_tile(a8, W, c8, bias, r8, (int32_t)(M - m1), N, K, lda, ldw, ldc, dtype, epilogue, 128, stream);
        }
    }
    if (tile != 128) {
        if (gemm_loop_interleaved()) tile = tile == 257 ? 260 : 264;
        int32_t rc = d3d_gemm_nt_tile(A, W, C, bias, residual, (int32_t)rows256, N, K, lda, ldw, ldc, dtype, epilogue, tile, stream);
        if (rc != D3D_OK || rows256 == M) return rc;
        const char* a8 = (const char*)A + rows256 * lda * 2;
        char* c8 = (char*)C + rows256 * ldc * 2;
        const char* r8 = residual ? (const char*)residual + rows256 * ldc * 2 : nullptr;
        return d3d_gemm_nt_tile(a8, W, c8, bias, r8, (int32_t)(M - rows256), N, K, lda, ldw, ldc, dtype, epilogue, 128, stream);
    }
    return d3d_gemm_nt_tile(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, dtype, epilogue, 128, stream);
}

int32_t d3d_gemm_nt_tile(const void* A, const void* W, void* C, const void* bias, const void* residual, int32_t M, int32_t N, int32_t K,
                         int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, int32_t tile, void* stream) {
    if (M <= 0) return D3D_OK;
    if (tile == 16) return skinny_dispatch(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, dtype, epilogue, stream);
    if (tile != 128 && tile != 130 && tile != 132 && tile != 164 && (tile < 256 || (tile > 264 && (tile < 301 || tile > 303)))) {
        d3d_set_error_("d3d_gemm_nt_tile: tile must be 128 (130 / 132: 2 / 4 LDS stages forced), 256 (K-half steps), 257 (whole-K-tile steps) or 258 (257 + split-K tail)");
        return D3D_EINVAL;
    }
    if (tile >= 256 && (N % TN != 0 || (ldc & 7) || ((uintptr_t)C & 15) || (residual && ((uintptr_t)residual & 15)))) {
        d3d_set_error_("d3d_gemm_nt_tile: the 256 x 256 tile needs N % 256 == 0, ldc % 8 == 0 and 16-byte aligned C / residual");
        return D3D_EINVAL;
    }
    if (N % BN != 0 || K % BK != 0 || (lda & 7) || (ldw & 7) || (ldc & 3)) {
        d3d_set_error_("d3d_gemm_nt: need N % 128 == 0, K % 64 == 0, lda/ldw % 8 == 0, ldc % 4 == 0");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
#define D3D_GEMM_CASE(E)                                                                                              \
    case E:                                                                                                           \
        if (tile == 256)                                                                                              \
            return dtype == 0 ? launch256<true, E, false>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)          \
                              : launch256<false, E, false>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);        \
        if (tile == 258)                                                                                              \
            return dtype == 0 ? launch256_split<true, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)           \
                              : launch256_split<false, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);         \
        if (tile == 264)                                                                                              \
            return dtype == 0 ? launch256_split<true, E, 2>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)        \
                              : launch256_split<false, E, 2>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);      \
        if (tile == 257)                                                                                              \
            return dtype == 0 ? launch256<true, E, true>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)           \
                              : launch256<false, E, true>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);         \
        if (tile == 259)                                                                                              \
            return dtype == 0 ? launch256<true, E, true, 1>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)        \
                              : launch256<false, E, true, 1>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);      \
        if (tile == 260)                                                                                              \
            return dtype == 0 ? launch256<true, E, true, 2>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)        \
                              : launch256<false, E, true, 2>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);      \
        if (tile == 301 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 16>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);    \
        if (tile == 302 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 18>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);    \
        if (tile == 303 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 20>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);    \
        if (tile == 263 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 5>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s); \
        if (tile == 261 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 3>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s); \
        if (tile == 262 && E == EPI_NONE) return launch256<true, EPI_NONE, true, 4>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s); \
        return dtype == 0 ? launch<true, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s, tile == 128 ? 0 : (tile == 164 ? 64 : tile - 128))  \
                          : launch<false, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s, tile == 128 ? 0 : (tile == 164 ? 64 : tile - 128));
    switch (epilogue) {
        D3D_GEMM_CASE(EPI_NONE)
        D3D_GEMM_CASE(EPI_BIAS)
        D3D_GEMM_CASE(EPI_BIAS_QGELU)
        D3D_GEMM_CASE(EPI_BIAS_GELU)
        D3D_GEMM_CASE(EPI_RES)
        D3D_GEMM_CASE(EPI_BIAS_RES)
        D3D_GEMM_CASE(EPI_SWIGLU)
        D3D_GEMM_CASE(EPI_LRELU)
        D3D_GEMM_CASE(EPI_LRELU_BWD)
    }
#undef D3D_GEMM_CASE
    d3d_set_error_("d3d_gemm_nt: unknown epilogue");
    return D3D_EINVAL;
}

}  // extern "C"
