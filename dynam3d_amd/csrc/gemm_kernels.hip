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

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;

using short8 = __attribute__((ext_vector_type(8))) short;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float4v = __attribute__((ext_vector_type(4))) float;

enum Epi : int { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_QGELU = 2, EPI_BIAS_GELU = 3, EPI_RES = 4, EPI_BIAS_RES = 5, EPI_SWIGLU = 6 };

template <bool BF16>
__device__ __forceinline__ float4v mfma16(const uint4& a, const uint4& b, float4v c) {
    if constexpr (BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&a), *reinterpret_cast<const half8*>(&b), c, 0, 0, 0);
    }
}

template <bool BF16>
__device__ __forceinline__ float to_f32(uint16_t v) {
    if constexpr (BF16) {
        return __uint_as_float((uint32_t)v << 16);
    } else {
        return __half2float(*reinterpret_cast<const __half*>(&v));
    }
}

template <bool BF16>
__device__ __forceinline__ uint16_t from_f32(float f) {
    if constexpr (BF16) {
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
        u += 0x7fffu + ((u >> 16) & 1u);                                          // round to nearest even
        return (uint16_t)(u >> 16);
    } else {
        __half h = __float2half_rn(f);
        return *reinterpret_cast<uint16_t*>(&h);
    }
}

__device__ __forceinline__ float act_qgelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }

// One 128x64 operand tile -> LDS (lane-linear image, source-side swizzle).  `rows_valid` clamps the
// row index (edge tiles re-read the last valid row; their outputs are never stored).
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ g, int64_t ld, int row0, int rows_valid, int k0,
                                           uint16_t* lds_tile, int wave, int lane) {
    // 16 KiB tile = 16 pieces of 1 KiB (8 rows x 128 B); wave w issues pieces w, w+4, w+8, w+12
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int piece = wave + p * 4;
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (r & 7);                 // swizzled source chunk for LDS chunk (lane&7)
        int gr = row0 + r;
        gr = gr < rows_valid ? gr : rows_valid - 1;
        const uint16_t* src = g + (int64_t)gr * ld + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + piece * 512), 16, 0, 0);
    }
}

template <bool BF16, int EPI>
__global__ void __launch_bounds__(NTHREADS, 2)
k_gemm_nt(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C,
          const uint16_t* __restrict__ bias, const uint16_t* __restrict__ residual, int M, int N, int K, int64_t lda,
          int64_t ldw, int64_t ldc, int tiles_m, int tiles_n) {
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
    const int row0 = tm * BM, col0 = tn * BN;

    const int wr = wave >> 1, wc = wave & 1;                // wave -> 64x64 sub-tile
    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    // LDS: buffer b -> A tile at smem + b*2*BM*BK, B tile right behind it
    stage_tile(A, lda, row0, M, 0, smem, wave, lane);
    stage_tile(W, ldw, col0, N, 0, smem + BM * BK, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fi = lane & 15, fg = lane >> 4;               // fragment row, k-group
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            uint16_t* nxt = smem + (cur ^ 1) * (2 * BM * BK);
            stage_tile(A, lda, row0, M, (t + 1) * BK, nxt, wave, lane);
            stage_tile(W, ldw, col0, N, (t + 1) * BK, nxt + BM * BK, wave, lane);
        }
        const uint16_t* la = smem + cur * (2 * BM * BK);
        const uint16_t* lb = la + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wr * 64 + i * 16 + fi;
                af[i] = *reinterpret_cast<const uint4*>(la + ra * BK + (((kk * 4 + fg) ^ (ra & 7)) << 3));
                const int rb = wc * 64 + i * 16 + fi;
                bf[i] = *reinterpret_cast<const uint4*>(lb + rb * BK + (((kk * 4 + fg) ^ (rb & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<BF16>(bf[j], af[i], acc[i][j]);   // D = C^T tile: rows n, cols m
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds, for tile (i,j): row m = row0+wr*64+i*16+fi, cols n = col0+wc*64+j*16+fg*4 .. +3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = row0 + wr * 64 + i * 16 + fi;
        if (m >= M) continue;
        if constexpr (EPI == EPI_SWIGLU) {
            // columns interleaved per 64: [gate 64 | up 64] inside every 128-wide tile; wc==0 waves hold gate, wc==1 up.
            // exchange through LDS is avoided by pairing tiles j of the two wave columns via __shfl is not possible
            // (different waves) -> the host lays gate/up out per 16: [g16|u16|g16|u16...], so tiles j=0,2 are gate and
            // j=1,3 the matching up columns of the SAME wave.
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int n_out = (col0 + wc * 64 + j * 16) / 2 + fg * 4;
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<BF16>(acc[i][j + 1][r] * act_silu(acc[i][j][r]));
                *reinterpret_cast<uint2*>(C + (int64_t)m * ldc + n_out) = *reinterpret_cast<const uint2*>(o);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = col0 + wc * 64 + j * 16 + fg * 4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
                    const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
                    const uint16_t* bp = reinterpret_cast<const uint16_t*>(&bb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32<BF16>(bp[r]);
                }
                if constexpr (EPI == EPI_BIAS_QGELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_qgelu(v[r]);
                }
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = act_gelu(v[r]);
                }
                if constexpr (EPI == EPI_RES || EPI == EPI_BIAS_RES) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(residual + (int64_t)m * ldc + n);
                    const uint16_t* rp = reinterpret_cast<const uint16_t*>(&rr);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32<BF16>(rp[r]);
                }
                uint16_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<BF16>(v[r]);
                *reinterpret_cast<uint2*>(C + (int64_t)m * ldc + n) = *reinterpret_cast<const uint2*>(o);
            }
        }
    }
}

template <bool BF16, int EPI>
int32_t launch(const void* A, const void* W, void* C, const void* bias, const void* res, int M, int N, int K, int64_t lda,
               int64_t ldw, int64_t ldc, hipStream_t s) {
    const int tm = (M + BM - 1) / BM, tn = N / BN;
    const size_t sh = 4 * BM * BK * sizeof(uint16_t);
    static bool attr_set = false;
    if (!attr_set) {
        D3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt<BF16, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gemm_nt<BF16, EPI>), dim3(tm * tn), dim3(NTHREADS), sh, s, (const uint16_t*)A, (const uint16_t*)W, (uint16_t*)C,
                       (const uint16_t*)bias, (const uint16_t*)res, M, N, K, lda, ldw, ldc, tm, tn);
    D3D_LAUNCH_CHECK();
}

}  // namespace

extern "C" {

// C[M,N] (or [M,N/2] for SwiGLU) = epi(A[M,K] W[N,K]^T).  dtype: 0 = bf16, 1 = fp16.  Requires N % 128 == 0, K % 64 == 0,
// 16-byte aligned rows (lda, ldw multiples of 8).  M is arbitrary (edge tiles are masked).
int32_t d3d_gemm_nt(const void* A, const void* W, void* C, const void* bias, const void* residual, int32_t M, int32_t N, int32_t K,
                    int64_t lda, int64_t ldw, int64_t ldc, int32_t dtype, int32_t epilogue, void* stream) {
    if (M <= 0) return D3D_OK;
    if (N % BN != 0 || K % BK != 0 || (lda & 7) || (ldw & 7) || (ldc & 3)) {
        d3d_set_error_("d3d_gemm_nt: need N % 128 == 0, K % 64 == 0, lda/ldw % 8 == 0, ldc % 4 == 0");
        return D3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
#define D3D_GEMM_CASE(E)                                                                                              \
    case E:                                                                                                           \
        return dtype == 0 ? launch<true, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s)                        \
                          : launch<false, E>(A, W, C, bias, residual, M, N, K, lda, ldw, ldc, s);
    switch (epilogue) {
        D3D_GEMM_CASE(EPI_NONE)
        D3D_GEMM_CASE(EPI_BIAS)
        D3D_GEMM_CASE(EPI_BIAS_QGELU)
        D3D_GEMM_CASE(EPI_BIAS_GELU)
        D3D_GEMM_CASE(EPI_RES)
        D3D_GEMM_CASE(EPI_BIAS_RES)
        D3D_GEMM_CASE(EPI_SWIGLU)
    }
#undef D3D_GEMM_CASE
    d3d_set_error_("d3d_gemm_nt: unknown epilogue");
    return D3D_EINVAL;
}

}  // extern "C"
