// skinny_kernels.hip -- the weight-streaming GEMM of a KV-cache decode token (M <= 16 rows), deep-prefetch version (round 6).
//
// Same work split, same arithmetic order and therefore the SAME BITS as k_gemm_skinny (gemm_kernels.hip): one workgroup owns 32 (HALF: 16)
// output columns, its 4 / 8 / 16 waves split K into contiguous step ranges, a step is one 16 x 32 W fragment (two when not HALF) loaded
// straight from global memory in MFMA operand layout, the waves' accumulators meet in LDS and wave 0 sums them in wave order.
//
// What changes is WHEN the weights are requested.  k_gemm_skinny keeps two batches of 4 steps in flight (4-8 KiB per wave) and starts its
// stream with one batch before the RMSNorm prologue; a decode projection gives a wave only 6-24 steps, so the kernel is all ramp: 22 us for
// qkv's 56.6 MB (2.6 TB/s), 25.6 us for gate_up's 100.7 MB (3.9 TB/s), 11.9 us average for o_proj / down_proj (2.9 TB/s).  Here a wave
// requests its first R = 16 (HALF) / 8 steps -- for qkv, o_proj, down_proj that is its WHOLE K range -- before anything else, then the
// workgroup stages the <= 16 activation rows in LDS (normalised for the NORM variants, raw otherwise: the activation fragments come from
// LDS on the lgkm counter instead of queueing behind the weight stream on the in-order vm counter), and every consumed slot is
// refilled at once.  Nothing but weights is on the vector-memory queue while the stream runs.
//
// Registers: a[16] (HALF) or a[8] + b[8] = 64 VGPRs of fragments under the 128-register budget of a 1024-thread workgroup; no scratch
// (tests/test_kernel_resources.py).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include <mutex>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

#include "gemm_epilogue.h"

template <bool BF16, int EPI, bool HALF, bool NORM>
__global__ void __launch_bounds__(1024)
k_gemm_skinny_deep(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W, uint16_t* __restrict__ C, const uint16_t* __restrict__ bias,
                   const uint16_t* __restrict__ residual, int M, int N, int K, int64_t ldx, int64_t ldw, int64_t ldc,
                   const float* __restrict__ nw, float eps, int x_in_lds) {
    extern __shared__ __attribute__((aligned(16))) float sk_lds[];      // activation rows [M][K + 8] 16-bit, later overlaid by [NW][2][64] float4
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int ct = blockIdx.x;
    const int nsteps = K / 32;
    const int s0 = (int)((int64_t)wave * nsteps / NW), s1 = (int)((int64_t)(wave + 1) * nsteps / NW);   // (k_gemm_skinny's split: same bits)
    float4v* red = reinterpret_cast<float4v*>(sk_lds);
    constexpr int COLS = HALF ? 16 : 32;
    constexpr int R = HALF ? 16 : 8;                                    // steps in flight per wave
    const uint16_t* w0 = W + (int64_t)(ct * COLS + fi) * ldw + fg * 8;
    const uint16_t* w1 = HALF ? w0 : w0 + 16 * ldw;
    uint4 a[R], b[HALF ? 1 : R];
    // ---- the weight stream starts here: the first R steps of this wave's K range ----
#pragma unroll
    for (int u = 0; u < R; ++u) {
        if (s0 + u < s1) {
            a[u] = *reinterpret_cast<const uint4*>(w0 + (s0 + u) * 32);
            if constexpr (!HALF) b[u] = *reinterpret_cast<const uint4*>(w1 + (s0 + u) * 32);
        }
    }
    const int xrow = fi < M ? fi : M - 1;                               // rows >= M: a duplicate, never stored
    const int XS = K + 8;                                               // LDS row stride in elements (16 B pad: rows 4 banks apart)
    uint16_t* xs = reinterpret_cast<uint16_t*>(sk_lds);
    if constexpr (NORM) {
        // HF Phi3RMSNorm of the <= 16 rows, once per workgroup, in k_norm's lane / chunk order (bit-identical to d3d_norm and to k_gemm_skinny)
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
                    float p = to_f32<BF16>(h[j]) * rstd, q = to_f32<BF16>(h[j + 1]) * rstd;
                    r16x2<BF16>(p, q);
                    o[j >> 1] = pack2<BF16>(p * ww[j], q * ww[j + 1]);
                }
                *reinterpret_cast<uint4*>(xs + r * XS + off) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
    } else if (x_in_lds) {
        const int cpr = K / 8;                                          // 16-byte chunks per row
        for (int i = threadIdx.x; i < M * cpr; i += blockDim.x) {
            const int r = i / cpr, c = i - r * cpr;
            *reinterpret_cast<uint4*>(xs + r * XS + c * 8) = *reinterpret_cast<const uint4*>(X + (int64_t)r * ldx + c * 8);
        }
        __syncthreads();
    }
    const uint16_t* xr = X + (int64_t)xrow * ldx + fg * 8;
    const bool lds_x = NORM || x_in_lds;
    auto xfrag = [&](int stp) -> uint4 {
        if (lds_x) return *reinterpret_cast<const uint4*>(xs + xrow * XS + stp * 32 + fg * 8);
        return *reinterpret_cast<const uint4*>(xr + stp * 32);
    };
    float4v acc0 = float4v{0.f, 0.f, 0.f, 0.f}, acc1 = float4v{0.f, 0.f, 0.f, 0.f};
    for (int st = s0; st < s1; st += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (st + u < s1) {                                          // (wave-uniform)
                const uint4 x = xfrag(st + u);
                acc0 = mfma16<BF16>(a[u], x, acc0);
                if constexpr (!HALF) acc1 = mfma16<BF16>(b[u], x, acc1);
                if (st + u + R < s1) {                                  // the slot is free: its next step goes out at once
                    a[u] = *reinterpret_cast<const uint4*>(w0 + (st + u + R) * 32);
                    if constexpr (!HALF) b[u] = *reinterpret_cast<const uint4*>(w1 + (st + u + R) * 32);
                }
            }
        }
    }
    if (lds_x) __syncthreads();                                         // every wave is done reading the rows the buffer below overlays
    red[(wave * 2 + 0) * 64 + lane] = acc0;
    red[(wave * 2 + 1) * 64 + lane] = acc1;
    __syncthreads();
    if (wave != 0 || fi >= M) return;
    acc0 = red[lane];
    acc1 = red[64 + lane];
    for (int q = 1; q < NW; ++q) {                                      // wave order: deterministic
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

template <bool BF16, int EPI, bool HALF, bool NORM>
int32_t launch_one(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda, int64_t ldw,
                   int64_t ldc, hipStream_t s, const float* nw, float eps, int ntiles, int nwv) {
    const size_t sh_red = (size_t)nwv * 2 * 64 * 16, sh_x = (size_t)M * (K + 8) * 2;
    const bool fits = sh_x <= 160 * 1024;
    if (NORM && !fits) {
        d3d_set_error_("d3d_gemm_nt_rmsnorm: rows x K does not fit the LDS");
        return D3D_EINVAL;
    }
    const size_t sh = (fits && sh_x > sh_red) ? sh_x : sh_red;
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny_deep<BF16, EPI, HALF, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    D3D_HIP(attr_err);
    hipLaunchKernelGGL((k_gemm_skinny_deep<BF16, EPI, HALF, NORM>), dim3(ntiles), dim3(nwv * 64), sh, s, (const uint16_t*)A, (const uint16_t*)W,
                       (uint16_t*)C, (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, nw, eps, fits ? 1 : 0);
    D3D_LAUNCH_CHECK();
}

template <bool BF16, int EPI, bool NORM>
int32_t launch_epi(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda, int64_t ldw,
                   int64_t ldc, hipStream_t s, const float* nw, float eps, int half, int ntiles, int nwv) {
    if constexpr (EPI != EPI_SWIGLU) {
        if (half) return launch_one<BF16, EPI, true, NORM>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s, nw, eps, ntiles, nwv);
    }
    return launch_one<BF16, EPI, false, NORM>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s, nw, eps, ntiles, nwv);
}

}  // namespace

// Called by launch_skinny (gemm_kernels.hip) with ITS launch geometry (half / ntiles / nwv): same work split, same bits.
// epilogue: 0 none | 1 bias | 4 residual | 6 SwiGLU; norm: the RMSNorm-fused variants (epilogue 0 or 6).
int32_t d3d_skinny_deep_launch_(int bf16, int epilogue, int norm, const void* A, const void* W, void* C, const void* bias, const void* res, int M,
                                int N, int K, int64_t lda, int64_t ldw, int64_t ldc, void* stream, const float* nw, float eps, int half,
                                int ntiles, int nwv) {
    hipStream_t s = (hipStream_t)stream;
#define D3D_DEEP(E, NORMV)                                                                                                              \
    return bf16 ? launch_epi<true, E, NORMV>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s, nw, eps, half, ntiles, nwv)                 \
                : launch_epi<false, E, NORMV>(A, W, C, bias, res, M, N, K, lda, ldw, ldc, s, nw, eps, half, ntiles, nwv)
    if (norm) {
        if (epilogue == EPI_NONE) { D3D_DEEP(EPI_NONE, true); }
        if (epilogue == EPI_SWIGLU) { D3D_DEEP(EPI_SWIGLU, true); }
    } else {
        if (epilogue == EPI_NONE) { D3D_DEEP(EPI_NONE, false); }
        if (epilogue == EPI_BIAS) { D3D_DEEP(EPI_BIAS, false); }
        if (epilogue == EPI_RES) { D3D_DEEP(EPI_RES, false); }
        if (epilogue == EPI_SWIGLU) { D3D_DEEP(EPI_SWIGLU, false); }
    }
#undef D3D_DEEP
    d3d_set_error_("skinny GEMM (deep prefetch): unsupported epilogue");
    return D3D_EINVAL;
}
