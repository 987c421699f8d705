// decode_attn_kernels.hip -- KV-cache decode attention for gfx950 (one query token per sequence): `d3d_decode_attention`, used by the
// greedy generation the reference's per-step call returns (`llava.generate(..., max_new_tokens=20, do_sample=False)`, VLN-POL:463; SURVEY.md
// 8 f-4).  The PREFILL attention is csrc/attn3_kernels.hip (`d3d_flash_attention_v3[_rope_q]`); the two earlier prefill kernels that used to
// live beside it (16x16x32 tiles with a pre-transposed V^T workspace, round 1; 32x32x16 tiles with register-staged K/V, round 3) were
// retired in round 4 -- their measurements are recorded in DESIGN.md section 4b.
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

// ================================================================================================
// Decode attention, second kernel (round 5): ONE pass over the keys with an online softmax, coalesced K / V rows.
//
// k_decode_attn above makes three dependent sweeps -- thread-per-key scores (a wave instruction touches 64 different rows, 16 bytes of
// each), a block softmax in LDS, then the value sum -- so the value rows are requested only after two block-wide reductions, with four
// waves per CU in flight: 29-31 us for 85 MB (2.8 TB/s), ~8 us of it the latency chain.  Here a key row is read by LPK consecutive lanes
// (16 bytes each: one 192-byte row = one contiguous segment), its K and V rows are requested TOGETHER, U keys per lane group are in
// flight before the first is used, and every lane group keeps its own running (max, sum, output) -- no synchronisation inside the
// sweep.  NWV waves per workgroup; the (NWV x KPW) partial states meet once in LDS and are merged in a fixed order (deterministic).
// Same arithmetic as k_decode_attn (float32 scores, probabilities and sums; RoPE of the new q / k as in k_rope), another summation order.
// ================================================================================================
template <bool BF16, int HD, int NWV, int U = 4>
__global__ void __launch_bounds__(NWV * 64)
k_decode_attn2(const uint16_t* __restrict__ qkv_new, const uint16_t* __restrict__ prompt, const int32_t* __restrict__ cu, uint16_t* __restrict__ knew,
               uint16_t* __restrict__ vnew, uint16_t* __restrict__ out, int H, int t_new, int Tmax, float scale, const float* __restrict__ cos_t,
               const float* __restrict__ sin_t, const int32_t* __restrict__ pos) {
    constexpr int CH = HD / 8;                          // 16-byte chunks per row: 12 / 8
    constexpr int LPK = CH <= 8 ? 8 : 16;               // lanes per key
    constexpr int KPW = 64 / LPK;                       // keys per wave instruction: 8 / 4
    // U = keys in flight per lane group
    constexpr int NP = NWV * KPW;                       // partial softmax states per workgroup
    __shared__ float qs[HD];
    __shared__ __attribute__((aligned(16))) uint16_t kcur[HD];
    __shared__ float po[NP][HD];
    __shared__ float pm[NP], pl[NP];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rs = (int64_t)3 * H * HD;
    const int r0 = cu[b], S = cu[b + 1] - r0, L = S + t_new + 1;
    const uint16_t* qrow = qkv_new + (int64_t)b * rs + (int64_t)h * HD;
    constexpr int HALF = HD / 2;
    if (tid < HALF) {                                   // (the prologue of k_decode_attn: RoPE of this token's q and k, side-cache append)
        float q1 = cvt16<BF16>(qrow[tid]), q2 = cvt16<BF16>(qrow[tid + HALF]);
        float k1 = cvt16<BF16>(qrow[(int64_t)H * HD + tid]), k2 = cvt16<BF16>(qrow[(int64_t)H * HD + tid + HALF]);
        uint16_t kr1, kr2;
        if (cos_t) {
            const float c = cos_t[(int64_t)pos[b] * HALF + tid], sn = sin_t[(int64_t)pos[b] * HALF + tid];
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
        uint16_t* kd = knew + (((int64_t)b * Tmax + t_new) * H + h) * HD;
        kd[tid] = kr1;
        kd[tid + HALF] = kr2;
    } else if (tid >= 64 && tid < 64 + HD / 8) {
        const int c = tid - 64;
        const uint4 v = *reinterpret_cast<const uint4*>(qrow + (int64_t)2 * H * HD + c * 8);
        *reinterpret_cast<uint4*>(vnew + (((int64_t)b * Tmax + t_new) * H + h) * HD + c * 8) = v;
    }
    __syncthreads();
    const int g = lane / LPK, c = lane % LPK;
    const bool lane_on = c < CH;
    const int cc = lane_on ? c : 0;                     // (idle lanes of a 16-lane group re-read chunk 0: their products are zeroed)
    float q[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = lane_on ? qs[cc * 8 + d] : 0.f;
    float m = -INFINITY, l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto dot8 = [&](const uint4& kv) -> float {
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&kv);
        float s0 = q[0] * cvt16<BF16>(e[0]) + q[4] * cvt16<BF16>(e[4]);
        float s1 = q[1] * cvt16<BF16>(e[1]) + q[5] * cvt16<BF16>(e[5]);
        float s2 = q[2] * cvt16<BF16>(e[2]) + q[6] * cvt16<BF16>(e[6]);
        float s3 = q[3] * cvt16<BF16>(e[3]) + q[7] * cvt16<BF16>(e[7]);
        float sdot = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int w = LPK / 2; w >= 1; w >>= 1) sdot += __shfl_xor(sdot, w, 64);       // the LPK lanes of one key
        return sdot;
    };
    auto fold = [&](float sc_, const uint4& vv, bool valid) {                            // one key into this lane group's running state
        if (!valid) return;
        const float m_new = fmaxf(m, sc_);
        const float alpha = __expf(m - m_new);          // (m = -inf at the first key: exp(-inf) = 0)
        const float p = __expf(sc_ - m_new);
        l = l * alpha + p;
        const uint16_t* e = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = o[d] * alpha + p * cvt16<BF16>(e[d]);
        m = m_new;
    };
    // ---- prompt keys: rows r0 .. r0 + S of the layer's prefill buffer, read in place -------------------------------------------------------
    const uint16_t* kbase = prompt + (int64_t)r0 * rs + (int64_t)(H + h) * HD + cc * 8;
    const uint16_t* vbase = prompt + (int64_t)r0 * rs + (int64_t)(2 * H + h) * HD + cc * 8;
    int j = wave * KPW + g;
    for (; j + (U - 1) * NP < S; j += U * NP) {
        uint4 kk[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kk[u] = *reinterpret_cast<const uint4*>(kbase + (int64_t)(j + u * NP) * rs);
            vv[u] = *reinterpret_cast<const uint4*>(vbase + (int64_t)(j + u * NP) * rs);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fold(dot8(kk[u]), vv[u], true);
    }
    for (; j < S; j += NP) {                             // (the shuffles of dot8 need the whole lane group: j is uniform inside it)
        const uint4 kk = *reinterpret_cast<const uint4*>(kbase + (int64_t)j * rs);
        const uint4 vv = *reinterpret_cast<const uint4*>(vbase + (int64_t)j * rs);
        fold(dot8(kk), vv, true);
    }
    // ---- generated tokens (side cache, written by EARLIER launches) and the current one (LDS / qkv_new: nothing this launch wrote is read back)
    for (int jt = S + wave * KPW + g; jt < L; jt += NP) {
        const bool cur = jt == L - 1;
        const uint16_t* kp = knew + (((int64_t)b * Tmax + (jt - S)) * H + h) * HD + cc * 8;
        const uint16_t* vp = cur ? qrow + (int64_t)2 * H * HD + cc * 8 : vnew + (((int64_t)b * Tmax + (jt - S)) * H + h) * HD + cc * 8;
        const uint4 kk = cur ? *reinterpret_cast<const uint4*>(kcur + cc * 8) : *reinterpret_cast<const uint4*>(kp);
        const uint4 vv = *reinterpret_cast<const uint4*>(vp);
        fold(dot8(kk), vv, true);
    }
    // ---- merge the NP partial states in a fixed order ----------------------------------------------------------------------------------------
    const int ps = wave * KPW + g;
    if (lane_on) {
#pragma unroll
        for (int d = 0; d < 8; ++d) po[ps][c * 8 + d] = o[d];
    }
    if (c == 0) { pm[ps] = m; pl[ps] = l; }
    __syncthreads();
    if (tid < HD / 2) {
        float mx = -INFINITY;
        for (int i = 0; i < NP; ++i) mx = fmaxf(mx, pm[i]);
        float lt = 0.f, a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < NP; ++i) {
            const float wgt = pm[i] == -INFINITY ? 0.f : __expf(pm[i] - mx);       // (a state that saw no key)
            lt += pl[i] * wgt;
            a0 += po[i][2 * tid] * wgt;
            a1 += po[i][2 * tid + 1] * wgt;
        }
        const float il = 1.0f / lt;
        *reinterpret_cast<uint32_t*>(out + ((int64_t)b * H + h) * HD + 2 * tid) = pack2<BF16>(a0 * il, a1 * il);
    }
}

}  // namespace

extern "C" {

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
    // Key ranges per (sequence, head) -- D3D_DECODE_SPLIT, default 1.  Measured (tools/experiments/bench_decode_attn.py, 8 x 32 heads x 864 keys, K/V of
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
    // D3D_DECODE_ATTN: 2 (default) = the one-pass kernel k_decode_attn2 (nsplit 1 only), 1 = the three-sweep kernel above
    const char* ae = getenv("D3D_DECODE_ATTN");                                // (read per call, like D3D_DECODE_SPLIT: the benchmarks sweep it)
    const int attn_kernel = ae ? atoi(ae) : 2;
    if (attn_kernel == 2 && nsplit == 1 && !fake_hm) {
        // D3D_DECODE_ATTN2_CFG = waves per workgroup * 100 + keys in flight per lane group (tuning knob: 804 default, 1604, 808, 1608)
        const char* ce = getenv("D3D_DECODE_ATTN2_CFG");
        const int cfg2 = ce ? atoi(ce) : 804;
        dim3 grid2(H, B);
#define D3D_DEC2K(BF, HDV, NW_, U_) hipLaunchKernelGGL((k_decode_attn2<BF, HDV, NW_, U_>), grid2, dim3(NW_ * 64), 0, s, (const uint16_t*)qkv_new, (const uint16_t*)prompt_qkv, \
                                             cu_seqlens, (uint16_t*)knew, (uint16_t*)vnew, (uint16_t*)out, H, t_new, Tmax, scale, cos_t, sin_t, pos)
#define D3D_DEC2(BF, HDV) do { if (cfg2 == 1604) D3D_DEC2K(BF, HDV, 16, 4); else if (cfg2 == 808) D3D_DEC2K(BF, HDV, 8, 8); else if (cfg2 == 1608) D3D_DEC2K(BF, HDV, 16, 8); \
                               else if (cfg2 == 404) D3D_DEC2K(BF, HDV, 4, 4); else D3D_DEC2K(BF, HDV, 8, 4); } while (0)
        if (dtype == 0) { if (head_dim == 96) D3D_DEC2(true, 96); else D3D_DEC2(true, 64); }
        else { if (head_dim == 96) D3D_DEC2(false, 96); else D3D_DEC2(false, 64); }
#undef D3D_DEC2
#undef D3D_DEC2K
        D3D_LAUNCH_CHECK();
    }
    dim3 grid(H, B, nsplit), block(256);
#define D3D_DEC(BF, HDV) hipLaunchKernelGGL((k_decode_attn<BF, HDV>), grid, block, 0, s, (const uint16_t*)qkv_new, (const uint16_t*)prompt_qkv, cu_seqlens, \
                                            (uint16_t*)knew, (uint16_t*)vnew, (uint16_t*)out, H, t_new, Tmax, scale, cos_t, sin_t, pos, nsplit, part, counters, fake_hm)
    if (dtype == 0) { if (head_dim == 96) D3D_DEC(true, 96); else D3D_DEC(true, 64); }
    else { if (head_dim == 96) D3D_DEC(false, 96); else D3D_DEC(false, 64); }
#undef D3D_DEC
    D3D_LAUNCH_CHECK();
}


}  // extern "C"
