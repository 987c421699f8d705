// decode_kernels.hip -- ONE cooperative launch per generated token: the whole Phi-3 decoder stack for <= 8 sequences.
//
// Reference: the decoder layers under `llava.generate(..., max_new_tokens=20, do_sample=False)` (VLN-POL:463; HF Phi3DecoderLayer
// with use_cache).  A decode token is weight streaming: 7.6 GB of bf16 weights + the prompts' keys / values (2.7 GB at 8 x 864
// prompt rows) against a few MFLOP per byte.  Issued as 7 launches per layer (phi3_decode.cpp) every kernel runs for 8-30 us, of
// which ~5 us are ramp-up (workgroup dispatch, first bytes' latency) and drain with the HBM pipe empty: 3.45 ms per token for
// 10.2 GB = 2.95 TB/s.  Here one persistent workgroup per CU (hipLaunchCooperativeKernel: co-residency is guaranteed by the
// runtime, or the launch fails) walks through all phases of all layers:
//
//     A  RMSNorm(x) -> LDS, qkv projection                      | grid barrier
//     B  RoPE(q, k) + attention over prompt K/V + side cache    | grid barrier      one (sequence, head) per workgroup
//     C  o_proj + residual                                      | grid barrier
//     D  RMSNorm(x) -> LDS, gate_up projection + SwiGLU         | grid barrier
//     E  down_proj + residual                                   | grid barrier
//     .. final RMSNorm + lm_head
//
// GEMM phases: a workgroup owns 16-column tiles (tile = blockIdx + r * grid); its 8 waves split K, every wave loads its W
// fragments straight into MFMA layout (16 rows x 64 B per instruction, 768 B - 2 KiB contiguous per row and tile, non-temporal), the
// 8 partial accumulators meet in LDS in wave order (deterministic).  The activations (8 rows) are staged once per phase in LDS,
// rows padded by 16 B so that the 16-lane ds_read_b128 groups hit distinct banks.  The weights of the NEXT tile -- or of the next
// PHASE's first tile -- are requested before the reduction / after arriving at the grid barrier, so HBM keeps streaming through
// reductions, epilogues and barriers; only the activations wait for the barrier.
// Grid barrier: one epoch flag per workgroup (write-through store; every workgroup polls the whole flag array with one coalesced
// agent-scope load per round), no cache maintenance (cross-workgroup data is stored write-through into per-layer scratch buffers that
// are written once and first read after their barrier), bounded spin: a workgroup that waits longer than ~seconds raises the error
// flag and the kernel unwinds instead of hanging the device.
// STATUS: parity-green, 3.72 ms per token against 3.45 ms for the launch-per-op path (profiles/r02_decode.txt) -- opt-in
// (D3D_DECODE_PERSISTENT=1).  What it does not win back: barrier + activation-staging latency per phase that a one-tile register
// prefetch does not cover, the attention phase (the same ~31 us as the stand-alone kernel), idle CUs in the o_proj / down_proj phases.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

#include "gemm_epilogue.h"

constexpr int MAXL = 32;                  // layers whose pointer tables travel in the kernel argument block
constexpr int DT = 512;                   // threads per workgroup (8 waves: two per SIMD, 256 VGPRs each)
constexpr int DEC_KEYS = 4096 + 64;       // prompt + generated keys one (sequence, head) item can hold in LDS

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

struct DecP {
    int n_layers, rows, heads, vocab, t_new, t_max;
    float eps, scale;
    uint16_t *x, *logits;
    uint16_t* ws;                         // per-layer scratch: [L] qkv (8, 3H) | [L] attn (8, H) | [L] act (8, MLP) | [2L] x (8, H)
    const uint16_t* qkv_w[MAXL];
    const uint16_t* o_w[MAXL];
    const uint16_t* gu_w[MAXL];
    const uint16_t* down_w[MAXL];
    const float* n1[MAXL];
    const float* n2[MAXL];
    const uint16_t* prompt_qkv[MAXL];
    const float* norm_w;
    const uint16_t* lm_head;
    const float *cos_t, *sin_t;
    const int32_t *pos, *cu;
    uint16_t *knew, *vnew;
    int64_t cache_layer_stride;           // elements
    unsigned* bar;                        // flags[grid] arrival epochs, [BAR_ERR] error flag
    int debug;                            // D3D_DECODE_DEBUG bits (experiments; 1, 4, 8 give WRONG results): 1 no barrier wait, 2 + acquire fence, 4 no attention,
                                          // 8 no GEMM tiles, 16 agent-scope activation loads, 32 phase stamps
};

__device__ __forceinline__ void st_agent(uint16_t* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)v.x | ((uint64_t)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint16_t* p, uint32_t v) {
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool BF16>
__device__ __forceinline__ float4v mfma_w(const u32x4& a, const uint4& b, float4v c) {
    if constexpr (BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), *reinterpret_cast<const half8*>(&b), c, 0, 0, 0);
    }
}

// W fragments of one tile for this wave's K slice: KS steps of 32 (PAIR: the gate rows, then the matching up rows 16 further down)
template <int KS, bool PAIR>
__device__ __forceinline__ void issue_w(u32x4 (&wb)[32], const uint16_t* __restrict__ W, int64_t ldw, int tile, int wave, int fi, int fg) {
    const uint16_t* p = W + (int64_t)(tile * (PAIR ? 32 : 16) + fi) * ldw + wave * (KS * 32) + fg * 8;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        wb[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + u * 32));
        if constexpr (PAIR) wb[KS + u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 16 * ldw + u * 32));
    }
}

// Grid barrier over flags, not a counter.  Data that crosses workgroups (qkv, attn, x, act) is stored WRITE-THROUGH (st_agent: relaxed
// agent-scope atomic stores, sc1), so no cache maintenance is needed when arriving:
//   arrive: every wave waits for its stores (s_waitcnt), the workgroup syncs, thread 0 stores the epoch into ITS slot flags[block];
//   wait  : thread t polls flags[t] (one coalesced read of the whole array per round) until every slot carries the epoch.
// Measured on the way here (ms per token, 161 barriers): agent-scope release fence from every wave (buffer_wbl2: writes the whole L2
// back) 14.3; one counter + atomic add per workgroup + agent-scope acquire fence (buffer_inv) 7.1, of which 3.0 waiting on the
// counter (256 read-modify-writes of one address queue up at the memory side) and 2.0 in the invalidates; flags + agent-scope loads 4.2.
constexpr int BAR_ERR = 1024;             // flags[0 .. grid): arrival epochs; flags[BAR_ERR]: sticky error flag

__device__ __forceinline__ void bar_arrive(unsigned* flags, unsigned epoch) {
    // A workgroup-scope release fence emits NO vmcnt wait on gfx950 (checked in the ISA: the write-through stores were followed directly
    // by s_barrier and the flag store), so the epoch could become visible before this wave's sc1 stores had been acknowledged and a
    // workgroup on another XCD could read stale activations behind bar_wait: every wave drains its own stores explicitly.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// false = the kernel is unwinding (some workgroup gave up waiting): uniform over the workgroup
__device__ __forceinline__ bool bar_wait(unsigned* flags, unsigned epoch, int debug) {
    if (debug & 1) return true;
    const int G = gridDim.x;
    unsigned spins = 0;
    for (;;) {
        int ok = 1;
        for (int t = threadIdx.x; t < G; t += DT) ok &= __hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (__syncthreads_and(ok)) break;
        int bad = 0;
        if (threadIdx.x == 0) {
            bad = __hip_atomic_load(flags + BAR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            if (++spins > (1u << 20)) {                          // ~seconds: co-residency lost or a workgroup died
                __hip_atomic_store(flags + BAR_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = 1;
            }
        }
        if (__syncthreads_or(bad)) return false;
    }
    if (debug & 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (experiment: the invalidate the per-layer buffers make unnecessary)
    return true;
}

// Cross-workgroup activations live in PER-LAYER scratch buffers (DecP::ws): an address is written once and read only after the barrier
// that follows, so no cache of any XCD can hold an older copy of it -- the kernel's launch invalidated them, and nothing read the line
// since.  The first workgroup of an XCD that touches a line pulls it from memory, the other 31 hit that XCD's L2 (plain 16-byte loads).
// With ONE buffer per tensor (reused by every layer) the loads have to be agent-scope (8-byte, every workgroup all the way to memory):
// debug bit 16 keeps those for comparison (4.11 against 4.01 ms per token).
__device__ __forceinline__ uint4 ld_agent16(const uint16_t* p) {
    const uint64_t lo = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ uint4 ld_x16(const uint16_t* p, bool agent) {
    return agent ? ld_agent16(p) : *reinterpret_cast<const uint4*>(p);
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// RMSNorm of the <= 8 rows of x (global) into LDS rows of HID + 8: wave w = row w, the lane / chunk order of k_norm (dense_kernels.hip),
// so the normalised rows are bit-identical to d3d_norm's.  HF Phi3RMSNorm: weight * x_hat.to(dtype).
template <bool BF16, int HID>
__device__ __forceinline__ void norm_stage(const uint16_t* x, const float* __restrict__ w, uint16_t* xs, int M, float eps, bool agent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NCH = HID / 512;
    if (wave < M) {
        const uint16_t* xr = x + (int64_t)wave * HID;
        float v[NCH][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint4 raw = ld_x16(xr + c * 512 + lane * 8, agent);
            const uint16_t* h = reinterpret_cast<const uint16_t*>(&raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = to_f32<BF16>(h[j]);
                ss += v[c][j] * v[c][j];
            }
        }
        ss = wave_sum64(ss);
        const float rstd = rsqrtf(ss / (float)HID + eps);
        uint16_t* yr = xs + wave * (HID + 8);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int off = c * 512 + lane * 8;
            const float4 w0 = *reinterpret_cast<const float4*>(w + off), w1 = *reinterpret_cast<const float4*>(w + off + 4);
            const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                float a = v[c][j] * rstd, b = v[c][j + 1] * rstd;
                r16x2<BF16>(a, b);
                o[j >> 1] = pack2<BF16>(a * ww[j], b * ww[j + 1]);
            }
            *reinterpret_cast<uint4*>(yr + off) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    __syncthreads();
}

// rows of `src` (global, width W elements) -> LDS rows of W + 8 (eight 16-byte chunks per thread in flight)
template <int W>
__device__ __forceinline__ void stage_x(const uint16_t* src, uint16_t* xs, int M, bool agent) {
    constexpr int CPR = W / 8;                          // 16-byte chunks per row
    const int total = M * CPR;
    for (int q0 = threadIdx.x; q0 < total; q0 += 8 * DT) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = q0 + i * DT;
            if (q < total) v[i] = ld_x16(src + (int64_t)q * 8, agent);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = q0 + i * DT;
            if (q < total) {
                const int r = q / CPR, c = q - r * CPR;
                *reinterpret_cast<uint4*>(xs + r * (W + 8) + c * 8) = v[i];
            }
        }
    }
    __syncthreads();
}

// One GEMM phase over this workgroup's tiles.  `have`: wb already holds the first tile's fragments (requested before the barrier).
template <bool BF16, int KS, bool PAIR, int EPI>
__device__ __forceinline__ void gemm_phase(u32x4 (&wb)[32], bool& have, const uint16_t* __restrict__ W, int64_t ldw, int ntiles,
                                           const uint16_t* xs, int xstride, int M, float4v* red, uint16_t* __restrict__ C,
                                           const uint16_t* __restrict__ residual, int64_t ldc, int dbg) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fi = lane & 15, fg = lane >> 4, G = gridDim.x;
    const uint16_t* xw = xs + (fi < M ? fi : M - 1) * xstride + wave * (KS * 32) + fg * 8;     // rows >= M: a duplicate, never stored
    for (int tile = blockIdx.x; tile < (dbg & 8 ? 0 : ntiles); tile += G) {
        if (!have) issue_w<KS, PAIR>(wb, W, ldw, tile, wave, fi, fg);
        have = false;
        float4v acc0 = float4v{0.f, 0.f, 0.f, 0.f}, acc1 = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const uint4 xf = *reinterpret_cast<const uint4*>(xw + u * 32);
            acc0 = mfma_w<BF16>(wb[u], xf, acc0);
            if constexpr (PAIR) acc1 = mfma_w<BF16>(wb[KS + u], xf, acc1);
        }
        if (tile + G < ntiles) {                                  // next tile's weights fly under the reduction and the epilogue
            issue_w<KS, PAIR>(wb, W, ldw, tile + G, wave, fi, fg);
            have = true;
        }
        red[(wave * 2 + 0) * 64 + lane] = acc0;
        if constexpr (PAIR) red[(wave * 2 + 1) * 64 + lane] = acc1;
        __syncthreads();
        if (wave == 0) {
            acc0 = red[lane];
            if constexpr (PAIR) acc1 = red[64 + lane];
#pragma unroll
            for (int q = 1; q < DT / 64; ++q) {                   // wave order: deterministic
                acc0 += red[(q * 2 + 0) * 64 + lane];
                if constexpr (PAIR) acc1 += red[(q * 2 + 1) * 64 + lane];
            }
            if (fi < M) {
                const int n16 = tile * (PAIR ? 32 : 16);
                const uint2 o = epi_pack<BF16, EPI, false>(acc0, PAIR ? acc1 : acc0, nullptr, residual, fi, n16, fg, ldc);
                st_agent(C + (int64_t)fi * ldc + (PAIR ? n16 / 2 : n16) + fg * 4, o);
            }
        }
        __syncthreads();
    }
}

// Phase B for one (sequence b, head h): RoPE of the new q / k (HF apply_rotary_pos_emb on 16-bit tensors, like k_rope), k / v appended
// to the side cache, softmax(q K^T) V over prompt rows (read in place from the prefill's post-RoPE QKV buffer) + generated tokens.
template <bool BF16, int HD>
__device__ __forceinline__ void attend(const DecP& p, int layer, int b, int h, uint8_t* lds, const uint16_t* qkv_new, uint16_t* attn_out, bool agent) {
    float* qs = reinterpret_cast<float*>(lds);                          // [HD]
    uint16_t* kcur = reinterpret_cast<uint16_t*>(lds + 512);            // [HD]
    float* red = reinterpret_cast<float*>(lds + 1024);                  // [16]
    float* sc = reinterpret_cast<float*>(lds + 2048);                   // [DEC_KEYS], later [NG][HD] partial outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, H = p.heads;
    const int64_t rs = (int64_t)3 * H * HD;
    const int r0 = p.cu[b], S = p.cu[b + 1] - r0, L = S + p.t_new + 1;
    const uint16_t* qrow = qkv_new + (int64_t)b * rs + (int64_t)h * HD;
    const uint16_t* prompt = p.prompt_qkv[layer];
    uint16_t* knew = p.knew + (int64_t)layer * p.cache_layer_stride;
    uint16_t* vnew = p.vnew + (int64_t)layer * p.cache_layer_stride;
    constexpr int HALF = HD / 2;
    // this token's q, k, v rows were written by other workgroups in phase A -> LDS
    uint16_t* nq = reinterpret_cast<uint16_t*>(lds + 1152);             // [3][HD]
    if (tid < 3 * (HD / 8)) {
        const int part = tid / (HD / 8), c = tid % (HD / 8);
        *reinterpret_cast<uint4*>(nq + part * HD + c * 8) = ld_x16(qrow + (int64_t)part * H * HD + c * 8, agent);
    }
    __syncthreads();
    if (tid < HALF) {
        float q1 = to_f32<BF16>(nq[tid]), q2 = to_f32<BF16>(nq[tid + HALF]);
        float k1 = to_f32<BF16>(nq[HD + tid]), k2 = to_f32<BF16>(nq[HD + tid + HALF]);
        const float c = p.cos_t[(int64_t)p.pos[b] * HALF + tid], sn = p.sin_t[(int64_t)p.pos[b] * HALF + tid];
        auto r = [](float f) { return to_f32<BF16>((uint16_t)pack2<BF16>(f, 0.f)); };
        const uint32_t qp = pack2<BF16>(r(q1 * c) - r(q2 * sn), r(q2 * c) + r(q1 * sn));
        const uint32_t kp = pack2<BF16>(r(k1 * c) - r(k2 * sn), r(k2 * c) + r(k1 * sn));
        qs[tid] = to_f32<BF16>((uint16_t)qp) * p.scale;
        qs[tid + HALF] = to_f32<BF16>((uint16_t)(qp >> 16)) * p.scale;
        kcur[tid] = (uint16_t)kp;
        kcur[tid + HALF] = (uint16_t)(kp >> 16);
        uint16_t* kd = knew + (((int64_t)b * p.t_max + p.t_new) * H + h) * HD;
        kd[tid] = (uint16_t)kp;
        kd[tid + HALF] = (uint16_t)(kp >> 16);
    } else if (tid >= 64 && tid < 64 + HD / 8) {
        const int c = tid - 64;
        *reinterpret_cast<uint4*>(vnew + (((int64_t)b * p.t_max + p.t_new) * H + h) * HD + c * 8) = *reinterpret_cast<const uint4*>(nq + 2 * HD + c * 8);
    }
    __syncthreads();
    auto krow = [&](int j) -> const uint16_t* {                                   // keys / values 0 .. L-2 (L-1 = this token: LDS)
        if (j < S) return prompt + (int64_t)(r0 + j) * rs + (int64_t)(H + h) * HD;
        return knew + (((int64_t)b * p.t_max + (j - S)) * H + h) * HD;
    };
    auto vrow = [&](int j) -> const uint16_t* {
        if (j < S) return prompt + (int64_t)(r0 + j) * rs + (int64_t)(2 * H + h) * HD;
        return vnew + (((int64_t)b * p.t_max + (j - S)) * H + h) * HD;
    };
    // ---- ONE pass over the keys (flash-decoding): 16 lanes per key, HD/8 of them carry a 16-byte chunk of the key row AND of the value
    //      row (one row = one contiguous 2*HD-byte read), 8 keys = 16 loads in flight per lane.  Every lane group keeps a running
    //      (max, sum, output chunk) with the usual rescaling; the 32 groups' partial results are merged in LDS in group order.
    constexpr int CH = HD / 8, NGRP = DT / 16;
    const int kg = tid >> 4, kc = tid & 15;
    const bool act = kc < CH;
    float qreg[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) qreg[d] = act ? qs[kc * 8 + d] : 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fold = [&](const uint4 (&kv)[8], const uint4 (&vv)[8], int nvalid) {
        float sv[8];
        float bm = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&kv[u]);
            float t = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) t += qreg[d] * to_f32<BF16>(e[d]);
            t += __shfl_xor(t, 8);
            t += __shfl_xor(t, 4);
            t += __shfl_xor(t, 2);
            t += __shfl_xor(t, 1);
            sv[u] = u < nvalid ? t : -INFINITY;
            bm = fmaxf(bm, sv[u]);
        }
        const float m_new = fmaxf(m_run, bm);
        const float resc = __expf(m_run - m_new);               // (first block: exp(-inf) = 0)
        l_run *= resc;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] *= resc;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float pe = __expf(sv[u] - m_new);             // invalid slots: exp(-inf) = 0
            l_run += pe;
            const uint16_t* e = reinterpret_cast<const uint16_t*>(&vv[u]);
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] += pe * to_f32<BF16>(e[d]);
        }
        m_run = m_new;
    };
    for (int j0 = kg; j0 < L - 1; j0 += 8 * NGRP) {
        uint4 kv[8], vv[8];
        int nvalid = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u * NGRP;
            kv[u] = make_uint4(0, 0, 0, 0);
            vv[u] = make_uint4(0, 0, 0, 0);
            if (j < L - 1) {
                nvalid = u + 1;
                if (act) {
                    kv[u] = *reinterpret_cast<const uint4*>(krow(j) + kc * 8);
                    vv[u] = *reinterpret_cast<const uint4*>(vrow(j) + kc * 8);
                }
            }
        }
        fold(kv, vv, nvalid);
    }
    if (kg == 0) {                                                   // this token's own key / value: LDS copies
        uint4 kv[8], vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) kv[u] = vv[u] = make_uint4(0, 0, 0, 0);
        if (act) {
            kv[0] = *reinterpret_cast<const uint4*>(kcur + kc * 8);
            vv[0] = *reinterpret_cast<const uint4*>(nq + 2 * HD + kc * 8);
        }
        fold(kv, vv, 1);
    }
    // ---- merge the groups: sc = [NGRP] max | [NGRP] sum | [NGRP][HD] outputs
    float* gm = sc;
    float* gl = sc + NGRP;
    float* go = sc + 2 * NGRP;
    if (kc == 0) {
        gm[kg] = m_run;
        gl[kg] = l_run;
    }
    if (act) {
#pragma unroll
        for (int d = 0; d < 8; ++d) go[kg * HD + kc * 8 + d] = o[d];
    }
    __syncthreads();
    if (tid < HD / 2) {
        float mx = gm[0];
        for (int g = 1; g < NGRP; ++g) mx = fmaxf(mx, gm[g]);
        float lt = 0.f, a0 = 0.f, a1 = 0.f;
        for (int g = 0; g < NGRP; ++g) {                          // fixed order: deterministic
            const float wgt = __expf(gm[g] - mx);                 // (a group that saw no key: exp(-inf) = 0)
            lt += gl[g] * wgt;
            a0 += go[g * HD + 2 * tid] * wgt;
            a1 += go[g * HD + 2 * tid + 1] * wgt;
        }
        const float inv = 1.0f / lt;
        st_agent(attn_out + ((int64_t)b * H + h) * HD + 2 * tid, pack2<BF16>(a0 * inv, a1 * inv));
    }
    __syncthreads();
}

template <bool BF16, int HID, int MLP, int HD>
__global__ void __launch_bounds__(DT, 2) k_phi3_decode_token(const DecP p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int XS_BYTES = 8 * (MLP + 8) * 2;
    uint16_t* xs = reinterpret_cast<uint16_t*>(lds);                                    // staged activations (or phase B's scratch)
    float4v* red = reinterpret_cast<float4v*>(lds + XS_BYTES);                           // [8 waves][2][64] partial accumulators
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), fi = lane & 15, fg = lane >> 4;
    const int M = p.rows, G = gridDim.x, bid = blockIdx.x;
    constexpr int KS_H = HID / 32 / (DT / 64), KS_M = MLP / 32 / (DT / 64);             // K steps per wave: 12 (K = hidden), 32 (K = mlp)
    u32x4 wb[32];
    bool have = false;
    unsigned target = 0;
#define D3D_GRID_BARRIER(...)                                   \
    target += 1u;                                               \
    bar_arrive(p.bar, target);                                  \
    __VA_ARGS__;                                                \
    if (!bar_wait(p.bar, target, p.debug)) return;

    const bool agent = (p.debug & 16) != 0;
    const int L = p.n_layers;
    // debug bit 32: workgroups 0 and G-1 stamp the constant 100 MHz counter at every phase boundary of layer 5 (flags + 1280 / + 1312)
    const bool stamping = (p.debug & 32) && (bid == 0 || bid == G - 1) && threadIdx.x == 0;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.bar + 1280 + (bid == 0 ? 0 : 32));
    int stamp_i = 0;
#define D3D_STAMP() if (stamping && l == 5) stamps[stamp_i++] = __builtin_amdgcn_s_memrealtime();
    uint16_t* const ws_qkv = p.ws;
    uint16_t* const ws_attn = ws_qkv + (int64_t)L * 8 * 3 * HID;
    uint16_t* const ws_act = ws_attn + (int64_t)L * 8 * HID;
    uint16_t* const ws_x = ws_act + (int64_t)L * 8 * MLP;
    const uint16_t* xin = p.x;                                    // the residual stream entering the layer
    for (int l = 0; l < L; ++l) {
        uint16_t* const qkv = ws_qkv + (int64_t)l * 8 * 3 * HID;
        uint16_t* const attn = ws_attn + (int64_t)l * 8 * HID;
        uint16_t* const act = ws_act + (int64_t)l * 8 * MLP;
        uint16_t* const xmid = ws_x + (int64_t)(2 * l) * 8 * HID;                             // after attention
        uint16_t* const xout = l + 1 < L ? ws_x + (int64_t)(2 * l + 1) * 8 * HID : p.x;       // after the MLP (last layer: the caller's x)
        const uint16_t* const nextw = l + 1 < L ? p.qkv_w[l + 1] : p.lm_head;                // the GEMM after this layer's down_proj
        const int nextn = l + 1 < L ? 3 * HID / 16 : p.vocab / 16;
        D3D_STAMP()
        // ---- A: input RMSNorm + qkv projection
        norm_stage<BF16, HID>(xin, p.n1[l], xs, M, p.eps, agent);
        D3D_STAMP()
        gemm_phase<BF16, KS_H, false, EPI_NONE>(wb, have, p.qkv_w[l], HID, 3 * HID / 16, xs, HID + 8, M, red, qkv, nullptr, 3 * HID, p.debug);
        D3D_STAMP()
        D3D_GRID_BARRIER(if (bid < HID / 16) { issue_w<KS_H, false>(wb, p.o_w[l], HID, bid, wave, fi, fg); have = true; })
        D3D_STAMP()
        // ---- B: attention, one (sequence, head) per workgroup; pairs of heads that share 128-byte lines of a K/V row land on one XCD
        {
            const int items = M * p.heads;
            for (int it = bid; it < items; it += G) {
                int gh = it;
                if (items % 16 == 0 && G % 16 == 0) {
                    const int base = it / G * G, j = it - base, q = (j & 7) + 8 * (j >> 4);
                    gh = base + 2 * q + ((j >> 3) & 1);          // (a bijection on every round: items and G are multiples of 16)
                }
                if (!(p.debug & 4)) attend<BF16, HD>(p, l, gh / p.heads, gh % p.heads, lds, qkv, attn, agent);
            }
        }
        D3D_STAMP()
        D3D_GRID_BARRIER((void)0)
        D3D_STAMP()
        // ---- C: o_proj + residual
        stage_x<HID>(attn, xs, M, agent);
        D3D_STAMP()
        gemm_phase<BF16, KS_H, false, EPI_RES>(wb, have, p.o_w[l], HID, HID / 16, xs, HID + 8, M, red, xmid, xin, HID, p.debug);
        D3D_STAMP()
        D3D_GRID_BARRIER(if (bid < MLP / 16) { issue_w<KS_H, true>(wb, p.gu_w[l], HID, bid, wave, fi, fg); have = true; })
        D3D_STAMP()
        // ---- D: post-attention RMSNorm + gate_up projection + SwiGLU (weights interleaved per 16 rows: gate tile, up tile)
        norm_stage<BF16, HID>(xmid, p.n2[l], xs, M, p.eps, agent);
        D3D_STAMP()
        gemm_phase<BF16, KS_H, true, EPI_SWIGLU>(wb, have, p.gu_w[l], HID, MLP / 16, xs, HID + 8, M, red, act, nullptr, MLP, p.debug);
        D3D_STAMP()
        D3D_GRID_BARRIER(if (bid < HID / 16) { issue_w<KS_M, false>(wb, p.down_w[l], MLP, bid, wave, fi, fg); have = true; })
        D3D_STAMP()
        // ---- E: down_proj + residual
        stage_x<MLP>(act, xs, M, agent);
        D3D_STAMP()
        gemm_phase<BF16, KS_M, false, EPI_RES>(wb, have, p.down_w[l], MLP, HID / 16, xs, MLP + 8, M, red, xout, xmid, HID, p.debug);
        D3D_STAMP()
        D3D_GRID_BARRIER(if (bid < nextn) { issue_w<KS_H, false>(wb, nextw, HID, bid, wave, fi, fg); have = true; })
        D3D_STAMP()
        xin = xout;
    }
    // the final hidden state sits in the caller's x: this workgroup read that buffer as layer 0's input, so -- like everything a second
    // read could find stale -- it is read with agent-scope loads
    norm_stage<BF16, HID>(p.x, p.norm_w, xs, M, p.eps, true);
    gemm_phase<BF16, KS_H, false, EPI_NONE>(wb, have, p.lm_head, HID, p.vocab / 16, xs, HID + 8, M, red, p.logits, nullptr, p.vocab, p.debug);
#undef D3D_GRID_BARRIER
#undef D3D_STAMP
}

struct DecodeState {
    unsigned* bar = nullptr;              // flags[BAR_ERR + 512] (device)
    uint16_t* ws = nullptr;               // per-layer activation scratch (MAXL layers)
    int grid = 0;
};
std::mutex g_dec_mu;
std::unordered_map<hipStream_t, DecodeState> g_dec_states;

template <bool BF16>
int32_t launch_decode(const d3d_phi3_decode_args* a, DecodeState* st) {
    using K = void (*)(const DecP);
    K kern = k_phi3_decode_token<BF16, 3072, 8192, 96>;
    constexpr size_t sh = 8 * (8192 + 8) * 2 + 8 * 2 * 64 * 16 + 64;
    hipStream_t s = (hipStream_t)a->stream;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    static int per_cu = 0, cus = 0;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (attr_err != hipSuccess) return;
        attr_err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), DT, sh);
        int dev = 0;
        hipDeviceProp_t prop;
        if (attr_err == hipSuccess) attr_err = hipGetDevice(&dev);
        if (attr_err == hipSuccess) attr_err = hipGetDeviceProperties(&prop, dev);
        cus = prop.multiProcessorCount;
    });
    D3D_HIP(attr_err);
    if (per_cu < 1 || cus < 1 || cus > BAR_ERR) {
        d3d_set_error_("d3d_phi3_decode_token: the persistent decode kernel does not fit on a CU");
        return D3D_EHIP;
    }
    DecP p;
    p.n_layers = a->n_layers, p.rows = a->rows, p.heads = a->heads, p.vocab = a->vocab, p.t_new = a->t_new, p.t_max = a->t_max;
    p.eps = a->rms_eps, p.scale = 1.0f / sqrtf((float)a->head_dim);
    p.x = (uint16_t*)a->x, p.logits = (uint16_t*)a->logits, p.ws = st->ws;
    for (int l = 0; l < a->n_layers; ++l) {
        p.qkv_w[l] = (const uint16_t*)a->qkv_w[l], p.o_w[l] = (const uint16_t*)a->o_w[l], p.gu_w[l] = (const uint16_t*)a->gate_up_w[l];
        p.down_w[l] = (const uint16_t*)a->down_w[l], p.n1[l] = a->n1[l], p.n2[l] = a->n2[l], p.prompt_qkv[l] = (const uint16_t*)a->prompt_qkv[l];
    }
    p.norm_w = a->norm_w, p.lm_head = (const uint16_t*)a->lm_head_w, p.cos_t = a->cos_t, p.sin_t = a->sin_t, p.pos = a->pos, p.cu = a->cu_seqlens;
    p.knew = (uint16_t*)a->knew, p.vnew = (uint16_t*)a->vnew, p.cache_layer_stride = a->cache_layer_stride_bytes / 2;
    p.bar = st->bar;
    {
        const char* e = getenv("D3D_DECODE_DEBUG");
        p.debug = e ? atoi(e) : 0;
    }
    D3D_HIP(hipMemsetAsync(st->bar, 0, BAR_ERR * sizeof(unsigned), s));            // the arrival epochs; the error flag is sticky
    void* args[] = {&p};
    D3D_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(cus * (per_cu > 1 ? 1 : per_cu)), dim3(DT), args, sh, s));
    return D3D_OK;
}

}  // namespace

extern "C" {

// 1 = the persistent kernel takes this configuration (llava-phi-3-mini's decoder: hidden 3072, mlp 8192, head_dim 96, <= 8 rows)
int32_t d3d_phi3_decode_persistent_ok(const d3d_phi3_decode_args* a) {
    const char* e = getenv("D3D_DECODE_PERSISTENT");              // read per call: tests switch between the two paths
    const bool off = !(e && e[0] == '1');                         // opt-in while the grid barrier is being tuned
    return !off && a && a->hidden == 3072 && a->mlp == 8192 && a->head_dim == 96 && a->heads * a->head_dim == a->hidden && a->rows >= 1 &&
           a->rows <= 8 && a->n_layers >= 1 && a->n_layers <= MAXL && a->vocab % 16 == 0 && a->cos_t && a->sin_t && a->pos &&
           a->max_prompt_len + a->t_max <= DEC_KEYS - 64 && (a->dtype == 0 || a->dtype == 1);
}

int32_t d3d_phi3_decode_token_persistent(const d3d_phi3_decode_args* a) {
    DecodeState* st;
    {
        std::lock_guard<std::mutex> lock(g_dec_mu);
        st = &g_dec_states[(hipStream_t)a->stream];
        if (!st->bar) {
            D3D_HIP(hipMalloc(&st->bar, (BAR_ERR + 512) * sizeof(unsigned)));
            D3D_HIP(hipMemset(st->bar, 0, (BAR_ERR + 512) * sizeof(unsigned)));
            D3D_HIP(hipMalloc(&st->ws, (size_t)MAXL * 8 * (3 * 3072 + 3072 + 8192 + 2 * 3072) * sizeof(uint16_t)));
        }
    }
    return a->dtype == 0 ? launch_decode<true>(a, st) : launch_decode<false>(a, st);
}

// diagnostics (D3D_DECODE_DEBUG bit 32): the phase stamps of layer 5, workgroups 0 and G-1, in units of 10 ns; 16 values each
int32_t d3d_phi3_decode_stamps(void* stream, uint64_t* out32) {
    unsigned* bar = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_dec_mu);
        auto it = g_dec_states.find((hipStream_t)stream);
        if (it != g_dec_states.end()) bar = it->second.bar;
    }
    if (!bar) return D3D_EINVAL;
    D3D_HIP(hipStreamSynchronize((hipStream_t)stream));
    D3D_HIP(hipMemcpy(out32, bar + 1280, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return D3D_OK;
}

// Blocking check of the stream's sticky error flag (a grid barrier that gave up waiting): call once per generation, after the last token.
int32_t d3d_phi3_decode_status(void* stream) {
    unsigned* bar = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_dec_mu);
        auto it = g_dec_states.find((hipStream_t)stream);
        if (it != g_dec_states.end()) bar = it->second.bar;
    }
    if (!bar) return D3D_OK;
    unsigned flag = 0;
    D3D_HIP(hipStreamSynchronize((hipStream_t)stream));
    D3D_HIP(hipMemcpy(&flag, bar + BAR_ERR, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
        D3D_HIP(hipMemset(bar + BAR_ERR, 0, sizeof(unsigned)));
        d3d_set_error_("d3d_phi3_decode_token: a grid barrier of the persistent decode kernel timed out (the token's logits are invalid)");
        return D3D_EHIP;
    }
    return D3D_OK;
}

}  // extern "C"
