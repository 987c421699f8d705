// f32x3_kernels.hip -- float32 GEMM of the 3D-token builder on the 16-BIT matrix cores, at float32 accuracy.
//
//   d3d_gemm_nt_f32x3   C[M,N] = epi(A[M,K] W[N,K]^T), float32 in, float32 out, same contract as d3d_gemm_nt_f32 (f32_kernels.hip).
//
// The set encoders / merge discriminator / prefix MLPs (VLN-FF:134-161, VLN-POL:83-111) are float32 modules, and v_mfma_f32_16x16x4_f32
// runs at 1/16 of the 16-bit MFMA rate (157 TFLOP/s peak; the float32 kernel reached ~50).  Here every float32 operand element x is
// split while its tile is staged into LDS
//       hi = fp16(x)   (round to nearest even)        lo = fp16(x - hi)      (x - hi is exact in float32)
// and the product is taken as three fp16 MFMAs with float32 accumulation:   a w  ~=  a_hi w_hi + a_hi w_lo + a_lo w_hi.
// hi + lo carries 22 significand bits of x (|x - hi - lo| <= 2^-22 |x|, or <= 2^-25 absolute once lo is subnormal: the fp16 MFMA
// does not flush subnormal INPUTS -- tools/probe/denorm_probe.py), the dropped a_lo w_lo term is <= 2^-22 |a w|, products of fp16 values
// are exact in the float32 accumulator: the result differs from a float32 GEMM by ~1e-6 relative, the size of a float32 summation-
// order effect (tests/test_gpu_f32x3.py measures it against float64).  3 MFMAs at 16x the rate = 5.3x the float32 matrix peak.
// Range: none for finite operands, with the ROW EXPONENTS (round 5): every operand row is scaled by a power of two (exact) so that its
// largest element lies in [1, 2) before the split -- `a_exp[m]` / `w_exp[n]` = floor(log2(max |row|)), computed by d3d_row_exponents
// (once per weight, once per call for the activations) -- and the accumulator is scaled back by 2^(a_exp[m] + w_exp[n]) in the epilogue.
// No finite row can overflow fp16 any more (|x| >= 65504 used to become inf), and a row of uniformly small values (1e-6: fp16 subnormals,
// 4 significant bits) keeps its 22 bits.  Within a row, an element below 2^-14 of the row's maximum keeps only its hi half: an absolute
// error <= 2^-25 x (row max) per element, invisible next to the row's leading terms in the same dot product.  Without exponent arrays
// (null) the kernel is the unscaled round-3 one.  `status` (nullable): bit 0 is OR-ed in when an output element is not finite -- a device
// word the caller reads when it pleases (the token builder: once per update, one step late, no synchronisation: f32_ops.py).
//
// Tile = (32 WT) x (32 WT) outputs per 256-thread workgroup, K step 32, every wave (16 WT)^2 = WT x WT tiles of v_mfma_f32_16x16x32_f16;
// global float4 prefetch of the next K step in asm (see f32_kernels.hip for why), converted at the LDS write; LDS rows are
// [32 hi | 32 lo | 8 pad] halves = 144 B: the ds_write_b64 of 16 lanes and the ds_read_b128 of 8 lanes each cover 128 distinct bytes of
// the bank space.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

using float4v = __attribute__((ext_vector_type(4))) float;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using half8 = __attribute__((ext_vector_type(8))) _Float16;

enum EpiF32 : int { F_NONE = 0, F_BIAS = 1, F_BIAS_GELU = 2, F_BIAS_RES = 3 };

constexpr int XBK = 32, XPITCH = 72;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ const float* row_ptr(const float* __restrict__ base, int64_t ld, int row, int rows_valid) {
    row = row < rows_valid ? row : rows_valid - 1;
    return base + (int64_t)row * ld;
}

__device__ __forceinline__ void gload4(float4v& d, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

__device__ __forceinline__ float pow2f(int e) {          // 2^e, e in [-126, 127]
    return __uint_as_float((uint32_t)(127 + e) << 23);
}

__device__ __forceinline__ void split_store(_Float16* dst, float4v v, const float sc) {
    half4 hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] *= sc;                                       // power of two: exact
        hi[r] = (_Float16)v[r];
        lo[r] = (_Float16)(v[r] - (float)hi[r]);
    }
    *reinterpret_cast<half4*>(dst) = hi;
    *reinterpret_cast<half4*>(dst + XBK) = lo;
}

template <int EPI, int WT>
__global__ void __launch_bounds__(256, 2)
k_gemm_f32x3(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, const float* __restrict__ bias,
             const float* __restrict__ residual, int M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc, int tiles_m, int tiles_n,
             const int32_t* __restrict__ a_exp, const int32_t* __restrict__ w_exp, int32_t* __restrict__ status) {
    constexpr int FBM = 32 * WT, FBN = 32 * WT, WAVE_T = 16 * WT;
    constexpr int NLD = FBM / 32;                                                 // float4 loads per thread and operand tile (rows lr + 32 j)
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_raw[];          // [buffer][A | W][FBM * XPITCH]: 72 KiB at WT = 4 (dynamic: > 64 KiB)
    auto smem = reinterpret_cast<_Float16(*)[2][FBM * XPITCH]>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups walk the N tiles of one group of 4 M tiles: the W panels stay hot in L2 (as k_gemm_f32)
    constexpr int GM = 4;
    const int wg = blockIdx.x;
    const int group = wg / (GM * tiles_n);
    const int gm0 = group * GM;
    const int gsz = min(GM, tiles_m - gm0);
    const int tm = gm0 + (wg % (GM * tiles_n)) % gsz;
    const int tn = (wg % (GM * tiles_n)) / gsz;
    const int row0 = tm * FBM, col0 = tn * FBN;
    const int wr = wave >> 1, wc = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;

    float4v acc[WT][WT];      // D = W_frag x A_frag: lane holds C[m = i-tile row fi][n = j-tile rows 4 fg .. +3]
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = K / XBK;
    const int lr = tid >> 3, lc = (tid & 7) * 4;             // this thread's first row / first float of the staged tiles
    const int lds_off = lr * XPITCH + lc;
    const float* pa[NLD];
    const float* pw[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        pa[j] = row_ptr(A, lda, row0 + lr + 32 * j, M) + lc;
        pw[j] = row_ptr(W, ldw, col0 + lr + 32 * j, N) + lc;
    }
    float sa[NLD], sw[NLD];                                  // 2^-exponent of this thread's staged rows (1 without exponent arrays)
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int ra_ = min(row0 + lr + 32 * j, M - 1), rw_ = min(col0 + lr + 32 * j, N - 1);
        sa[j] = a_exp ? pow2f(-a_exp[ra_]) : 1.0f;
        sw[j] = w_exp ? pow2f(-w_exp[rw_]) : 1.0f;
    }
    float4v ra[NLD], rw[NLD];
    auto issue = [&](int k) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            gload4(ra[j], pa[j] + k);
            gload4(rw[j], pw[j] + k);
        }
    };
    auto land = [&](int buf) {
        if constexpr (NLD == 4) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rw[0]), "+v"(rw[1]), "+v"(rw[2]), "+v"(rw[3])::"memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(rw[0]), "+v"(rw[1])::"memory");
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            split_store(smem[buf][0] + lds_off + 32 * j * XPITCH, ra[j], sa[j]);
            split_store(smem[buf][1] + lds_off + 32 * j * XPITCH, rw[j], sw[j]);
        }
    };
    issue(0);
    land(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        issue((t + 1 < nk ? t + 1 : t) * XBK);               // next K step in flight under this step's MFMAs (the last step re-loads its own: never used)
        __builtin_amdgcn_sched_barrier(0);
        const _Float16* la = smem[cur][0] + (wr * WAVE_T + fi) * XPITCH + fg * 8;
        const _Float16* lw = smem[cur][1] + (wc * WAVE_T + fi) * XPITCH + fg * 8;
        // lane group fg supplies k = 8 fg .. 8 fg + 7 of this 32-deep step -- the same bijection for both operands and for hi / lo
        half8 ah[WT], al[WT], wh[WT], wl[WT];
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            ah[i] = *reinterpret_cast<const half8*>(la + i * 16 * XPITCH);
            al[i] = *reinterpret_cast<const half8*>(la + i * 16 * XPITCH + XBK);
            wh[i] = *reinterpret_cast<const half8*>(lw + i * 16 * XPITCH);
            wl[i] = *reinterpret_cast<const half8*>(lw + i * 16 * XPITCH + XBK);
        }
        // the two cross terms first, the leading term last; WT^2 independent accumulators between two MFMAs of one chain
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        land(cur ^ 1);
        __syncthreads();
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int m = row0 + wr * WAVE_T + i * 16 + fi;
        if (m >= M) continue;
        const float back_a = a_exp ? pow2f(a_exp[m]) : 1.0f;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = col0 + wc * WAVE_T + j * 16 + fg * 4;
            if (n >= N) continue;
            float4v v = acc[i][j];
            if (a_exp || w_exp) {                             // undo the row scales: two exact multiplications (no intermediate overflow of 2^(ea + ew))
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (v[r] * back_a) * (w_exp ? pow2f(w_exp[n + r]) : 1.0f);
            }
            if constexpr (EPI != F_NONE) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == F_BIAS_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if constexpr (EPI == F_BIAS_RES) {
                const float4 rr = *reinterpret_cast<const float4*>(residual + (int64_t)m * ldc + n);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            bad |= !(fabsf(v[0]) <= 3.4028235e38f) | !(fabsf(v[1]) <= 3.4028235e38f) | !(fabsf(v[2]) <= 3.4028235e38f) | !(fabsf(v[3]) <= 3.4028235e38f);
            *reinterpret_cast<float4*>(C + (int64_t)m * ldc + n) = float4{v[0], v[1], v[2], v[3]};
        }
    }
    if (status && __any(bad) && lane == 0) atomicOr(status, 1);
}

// floor(log2(max |row|)) of every row (0 for an all-zero row; clamped to [-126, 126] = the exponents pow2f(+-e) can build: the scaled maximum
// of ANY finite row then lies in [2^-23, 4) -- no finite row overflows fp16); a non-finite element ORs bit 1 into `status` and is
// left out of the maximum.  One wave per row.
__global__ void __launch_bounds__(256)
k_row_exponents(const float* __restrict__ X, int M, int K, int64_t ld, int32_t* __restrict__ out, int32_t* __restrict__ status) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* x = X + (int64_t)row * ld;
    float mx = 0.f;
    bool bad = false;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        const float a[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (a[r] <= 3.4028235e38f) mx = fmaxf(mx, a[r]);
            else bad = true;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (status && __any(bad) && lane == 0) atomicOr(status, 2);
    if (lane == 0) {
        int e = mx > 0.f ? (int)((__float_as_uint(mx) >> 23) & 0xff) - 127 : 0;      // (a subnormal maximum reads as -127: clamped below)
        out[row] = max(-126, min(126, e));
    }
}

}  // namespace

extern "C" {

int32_t d3d_row_exponents(const float* X, int32_t M, int32_t K, int64_t ld, int32_t* out, int32_t* status, void* stream) {
    if (M <= 0) return D3D_OK;
    if (K % 4 != 0 || K <= 0 || (ld & 3) || !X || !out) {
        d3d_set_error_("d3d_row_exponents: need K % 4 == 0, ld % 4 == 0 and non-null buffers");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_row_exponents, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, M, K, ld, out, status);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_gemm_nt_f32x3(const float* A, const float* W, float* C, const float* bias, const float* residual, int32_t M, int32_t N, int32_t K,
                          int64_t lda, int64_t ldw, int64_t ldc, int32_t epilogue, const int32_t* a_exp, const int32_t* w_exp, int32_t* status,
                          void* stream) {
    if (M <= 0) return D3D_OK;
    if (N % 4 != 0 || K % XBK != 0 || K <= 0 || (lda & 3) || (ldw & 3) || (ldc & 3)) {
        d3d_set_error_("d3d_gemm_nt_f32x3: need N % 4 == 0, K % 32 == 0 (zero-pad), lda / ldw / ldc % 4 == 0");
        return D3D_EINVAL;
    }
    if ((epilogue != F_NONE && !bias) || (epilogue == F_BIAS_RES && !residual)) {
        d3d_set_error_("d3d_gemm_nt_f32x3: epilogue needs bias (1, 2, 3) / residual (3)");
        return D3D_EINVAL;
    }
    // 128 x 128 tiles when they cover at least half the CUs, 64 x 64 tiles otherwise
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool big = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * 2 >= cus;
    const int T = big ? 128 : 64;
    const int tm = (M + T - 1) / T, tn = (N + T - 1) / T;
    dim3 grid(tm * tn), block(256);
    hipStream_t s = (hipStream_t)stream;
    const size_t sh = (size_t)2 * 2 * T * XPITCH * sizeof(_Float16);
    hipError_t attr_err = hipSuccess;
#define D3D_F32X3_CASE(E)                                                                                                           \
    case E:                                                                                                                         \
        if (big) {                                                                                                                  \
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_f32x3<E, 4>),                  \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 2 * 128 * XPITCH * sizeof(_Float16))); \
            attr_err = once;                                                                                                        \
            if (attr_err == hipSuccess)                                                                                             \
                hipLaunchKernelGGL((k_gemm_f32x3<E, 4>), grid, block, sh, s, A, W, C, bias, residual, M, N, K, lda, ldw, ldc, tm, tn, a_exp, w_exp, status); \
        } else {                                                                                                                    \
            hipLaunchKernelGGL((k_gemm_f32x3<E, 2>), grid, block, sh, s, A, W, C, bias, residual, M, N, K, lda, ldw, ldc, tm, tn, a_exp, w_exp, status);  \
        }                                                                                                                           \
        break;
    switch (epilogue) {
        D3D_F32X3_CASE(F_NONE)
        D3D_F32X3_CASE(F_BIAS)
        D3D_F32X3_CASE(F_BIAS_GELU)
        D3D_F32X3_CASE(F_BIAS_RES)
        default:
            d3d_set_error_("d3d_gemm_nt_f32x3: epilogue 0 none, 1 bias, 2 bias + GELU, 3 bias + residual");
            return D3D_EINVAL;
    }
#undef D3D_F32X3_CASE
    if (attr_err != hipSuccess) {
        d3d_set_error_("d3d_gemm_nt_f32x3: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        return D3D_EHIP;
    }
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
