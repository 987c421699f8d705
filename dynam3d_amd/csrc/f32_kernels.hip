// f32_kernels.hip -- float32 dense kernels of the 3D-token builder (a7, a9, a11, a14): the set encoders, the merge discriminator and
// the prefix MLPs (VLN-FF:134-161, VLN-POL:83-111) are float32 modules whose decisions (merge = an argmax) are pinned bit for bit by
// golden trajectories, so they stay in float32 -- on the matrix cores:
//
//   d3d_gemm_nt_f32      C[M,N] = epi(A[M,K] W[N,K]^T): v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bitwise a k-ordered fmaf chain,
//                        157 TFLOP/s peak = 1/16 of bf16 MFMA; MI355X_MICROARCH.md), 128 x 128 x 16 tiles, 4 waves of 64 x 64,
//                        register-staged double-buffered LDS, fused bias / GELU(erf) / residual epilogues, float4 stores.
//   d3d_linear_smallk_f32  y = x W^T + b for K <= 8 (the 3- / 4- / 6- / 7-wide geometry inputs of the position-embedding MLPs)
//   d3d_linear_smalln_f32  y = x W^T + b for N <= 8 (the 2 logits of the merge discriminator): one wave per row
//   d3d_layer_norm_f32     y = [gelu](LayerNorm(x [+ residual])) -- the LN / GELU / residual-add glue of nn.TransformerEncoderLayer
//                          (post-LN) and of nn.Sequential(Linear, LayerNorm, GELU, Linear) in one pass per row
//
// Replaces hipBLASLt sgemm + ~10 PyTorch element-wise launches per encoder layer (round 1: 2.1 ms of sgemm + ~2 ms of glue per step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

using float4v = __attribute__((ext_vector_type(4))) float;

enum EpiF32 : int { F_NONE = 0, F_BIAS = 1, F_BIAS_GELU = 2, F_BIAS_RES = 3, F_BIAS_QGELU = 4, F_RES = 5 };    // 4 / 5: the float32 verification towers (QuickGELU; bias-free residual)

constexpr int FBK = 16, FPITCH = 24;   // LDS row pitch 24 floats (96 B): ds_read_b128 of lane (row i, k-group g) hits 16-byte slot
                                       // (6 i + g) % 16 -- conflict-free for every 16-lane service group
// Tile = (32 WT) x (32 WT) outputs per 256-thread workgroup, every wave (16 WT) x (16 WT) = WT x WT MFMA tiles.  WT = 4: 128 x 128, 64
// MFMAs per wave and K step -- the throughput shape.  WT = 2: 64 x 64, 16 MFMAs per step -- four times as many workgroups and a
// quarter of the serial MFMA chain per K step, for the small-M launches (the CLS-only second encoder layer, zone sets, merge
// discriminator: M = 100-500 rows) whose time is K/16 dependent steps, not FLOPs.

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// One 128 x 16 operand tile: thread t moves rows (t >> 2) and (t >> 2) + 64, floats [4 (t & 3), +4).  Rows beyond the operand's last
// row re-read that row (their outputs are never stored).
//
// The prefetch loads are INLINE ASM.  hipcc sinks ordinary loads to their first use: written as C++ (one- or two-step prefetch distance,
// named registers, alternating register sets -- all tried) the loads of the next tile were always issued behind the 64 MFMAs, right in
// front of the ds_write that consumes them, and their latency was exposed on every K step (a struct / array of staged values was even
// demoted to scratch memory).  An asm load is invisible to that pass; it is waited for by hand (vmcnt(0) tied to the four registers).
__device__ __forceinline__ const float* row_ptr(const float* __restrict__ base, int64_t ld, int row, int rows_valid) {
    row = row < rows_valid ? row : rows_valid - 1;
    return base + (int64_t)row * ld;
}

__device__ __forceinline__ void gload4(float4v& d, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}

template <int EPI, int WT>
__global__ void __launch_bounds__(256, 2)
k_gemm_f32(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, const float* __restrict__ bias,
           const float* __restrict__ residual, int M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc, int tiles_m, int tiles_n) {
    constexpr int FBM = 32 * WT, FBN = 32 * WT, WAVE_T = 16 * WT;
    constexpr int NLD = FBM / 64;                                                 // float4 loads per thread and operand tile
    __shared__ __attribute__((aligned(16))) float smem[2][2][FBM * FPITCH];      // [buffer][A | W]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // consecutive workgroups walk the N tiles of one group of 4 M tiles: the W panels stay hot in L2
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

    float4v acc[WT][WT];      // [i: M tile][j: N tile]; D = W_frag x A_frag -> lane holds C[m = i-tile row fi][n = j-tile rows 4 fg .. +3]
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = K / FBK;
    const int lr = tid >> 2, lc = (tid & 3) * 4;             // this thread's row (and row + 64 at WT = 4) / first float of the staged tiles
    const int lds_off = lr * FPITCH + lc;
    const float* pa0 = row_ptr(A, lda, row0 + lr, M) + lc;
    const float* pa1 = row_ptr(A, lda, row0 + lr + 64, M) + lc;
    const float* pw0 = row_ptr(W, ldw, col0 + lr, N) + lc;
    const float* pw1 = row_ptr(W, ldw, col0 + lr + 64, N) + lc;
    float4v a0, a1, w0, w1;
    auto issue = [&](int k) {
        gload4(a0, pa0 + k);
        gload4(w0, pw0 + k);
        if constexpr (NLD == 2) {
            gload4(a1, pa1 + k);
            gload4(w1, pw1 + k);
        }
    };
    auto land = [&](int buf) {
        if constexpr (NLD == 2) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(w0), "+v"(w1)::"memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(w0)::"memory");
        }
        *reinterpret_cast<float4v*>(smem[buf][0] + lds_off) = a0;
        *reinterpret_cast<float4v*>(smem[buf][1] + lds_off) = w0;
        if constexpr (NLD == 2) {
            *reinterpret_cast<float4v*>(smem[buf][0] + lds_off + 64 * FPITCH) = a1;
            *reinterpret_cast<float4v*>(smem[buf][1] + lds_off + 64 * FPITCH) = w1;
        }
    };
    issue(0);
    land(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        issue((t + 1 < nk ? t + 1 : t) * FBK);               // next tile in flight under this step's MFMAs (the last step re-loads its own: never used)
        __builtin_amdgcn_sched_barrier(0);
        const float* la = smem[cur][0] + (wr * WAVE_T + fi) * FPITCH + fg * 4;
        const float* lw = smem[cur][1] + (wc * WAVE_T + fi) * FPITCH + fg * 4;
        float4 af[WT], wf[WT];
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            af[i] = *reinterpret_cast<const float4*>(la + i * 16 * FPITCH);
            wf[i] = *reinterpret_cast<const float4*>(lw + i * 16 * FPITCH);
        }
        // lane group fg supplies k = 4 fg + c of this 16-deep tile to the c-th MFMA of a (row tile, column tile) pair -- the same
        // bijection for both operands, so every k is multiplied exactly once
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) {
                    const float a = c == 0 ? af[i].x : c == 1 ? af[i].y : c == 2 ? af[i].z : af[i].w;
                    const float w = c == 0 ? wf[j].x : c == 1 ? wf[j].y : c == 2 ? wf[j].z : wf[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, a, acc[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
        land(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < WT; ++i) {
        const int m = row0 + wr * WAVE_T + i * 16 + fi;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = col0 + wc * WAVE_T + j * 16 + fg * 4;
            if (n >= N) continue;
            float4v v = acc[i][j];
            if constexpr (EPI != F_NONE && EPI != F_RES) {
                const float4 b = *reinterpret_cast<const float4*>(bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if constexpr (EPI == F_BIAS_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            if constexpr (EPI == F_BIAS_QGELU) {      // clip/model.py:162-164: x * sigmoid(1.702 x)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-1.702f * v[r]));
            }
            if constexpr (EPI == F_BIAS_RES || EPI == F_RES) {
                const float4 rr = *reinterpret_cast<const float4*>(residual + (int64_t)m * ldc + n);
                v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
            *reinterpret_cast<float4*>(C + (int64_t)m * ldc + n) = float4{v[0], v[1], v[2], v[3]};
        }
    }
}

// y[m, n] = b[n] + sum_k x[m, k] W[n, k], K <= 8.  A lane owns 4 output columns and keeps their 4 x 8 weights in registers; a wave walks
// down the rows of its block (row r, r + 4, ...: the row's K inputs are one wave-uniform read), storing 1 KiB of consecutive outputs per
// instruction.  (The first version re-read the weights for every output: 50 us for 4 600 x 768 outputs; the write alone is ~3 us.)
constexpr int SMALLK_ROWS = 64;           // rows per workgroup (4 waves x 16)
template <int EPI>
__global__ void __launch_bounds__(256)
k_linear_smallk(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y, int M,
                int N, int K, int64_t ldx, int64_t ldy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + lane) * 4;
    const bool live = n < N;
    float w[4][8], bb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bb[r] = live ? bias[n + r] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) w[r][k] = (live && k < K) ? W[(int64_t)(n + r) * K + k] : 0.f;
    }
    const int m_end = min(M, (int)(blockIdx.y + 1) * SMALLK_ROWS);
    for (int m = blockIdx.y * SMALLK_ROWS + wave; m < m_end; m += 4) {
        const float* xr = x + (int64_t)m * ldx;
        float xv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = k < K ? xr[k] : 0.f;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s = fmaf(xv[k], w[r][k], s);      // (k >= K: + 0 * 0, exact)
            o[r] = s + bb[r];
            if constexpr (EPI == F_BIAS_GELU) o[r] = gelu_erf(o[r]);
        }
        if (live) *reinterpret_cast<float4*>(y + (int64_t)m * ldy + n) = float4{o[0], o[1], o[2], o[3]};
    }
}

// y[m, n] = b[n] + sum_k x[m, k] W[n, k], N <= 8: one wave per row
__global__ void __launch_bounds__(256)
k_linear_smalln(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y, int M, int N, int K,
                int64_t ldx, int64_t ldy) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float s[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) s[n] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + k);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (n < N) {
                const float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)n * K + k);
                s[n] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, s[n]))));
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[n] += __shfl_xor(s[n], o);
    if (lane < N) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) v = lane == n ? s[n] : v;
        y[(int64_t)row * ldy + lane] = v + bias[lane];
    }
}

// y = [gelu](LN(x [+ res])): one wave per row, D <= 3072 (D % 4 == 0), float32 statistics (two-pass variance, biased)
template <int NCH>
__global__ void __launch_bounds__(256)
k_layer_norm_f32(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                 int rows, int D, int64_t ldx, int64_t ldr, int64_t ldy, float eps, int gelu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 256 + lane * 4;
        if (off < D) {
            float4 a = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + off);
            if (res) {
                const float4 r = *reinterpret_cast<const float4*>(res + (int64_t)row * ldr + off);
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
            v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
            s += (a.x + a.y) + (a.z + a.w);
        } else {
            v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c * 256 + lane * 4 < D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[c][j] - mean;
                q += d * d;
            }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int off = c * 256 + lane * 4;
        if (off < D) {
            const float4 ww = *reinterpret_cast<const float4*>(w + off), bb = *reinterpret_cast<const float4*>(b + off);
            float o[4] = {(v[c][0] - mean) * rstd * ww.x + bb.x, (v[c][1] - mean) * rstd * ww.y + bb.y, (v[c][2] - mean) * rstd * ww.z + bb.z,
                          (v[c][3] - mean) * rstd * ww.w + bb.w};
            if (gelu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = gelu_erf(o[j]);
            }
            *reinterpret_cast<float4*>(y + (int64_t)row * ldy + off) = float4{o[0], o[1], o[2], o[3]};
        }
    }
}

}  // namespace

extern "C" {

int32_t d3d_gemm_nt_f32(const float* A, const float* W, float* C, const float* bias, const float* residual, int32_t M, int32_t N, int32_t K,
                        int64_t lda, int64_t ldw, int64_t ldc, int32_t epilogue, void* stream) {
    if (M <= 0) return D3D_OK;
    if (N % 4 != 0 || K % FBK != 0 || K <= 0 || (lda & 3) || (ldw & 3) || (ldc & 3)) {
        d3d_set_error_("d3d_gemm_nt_f32: need N % 4 == 0, K % 16 == 0 (zero-pad), lda / ldw / ldc % 4 == 0");
        return D3D_EINVAL;
    }
    if ((epilogue != F_NONE && epilogue != F_RES && !bias) || ((epilogue == F_BIAS_RES || epilogue == F_RES) && !residual)) {
        d3d_set_error_("d3d_gemm_nt_f32: epilogue needs bias (1, 2, 3, 4) / residual (3, 5)");
        return D3D_EINVAL;
    }
    // 128 x 128 tiles when they cover at least half the CUs, 64 x 64 tiles otherwise (see k_gemm_f32)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool big = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * 2 >= cus;
    const int T = big ? 128 : 64;
    const int tm = (M + T - 1) / T, tn = (N + T - 1) / T;
    dim3 grid(tm * tn), block(256);
    hipStream_t s = (hipStream_t)stream;
#define D3D_F32_CASE(E)                                                                                                             \
    case E:                                                                                                                         \
        if (big)                                                                                                                    \
            hipLaunchKernelGGL((k_gemm_f32<E, 4>), grid, block, 0, s, A, W, C, bias, residual, M, N, K, lda, ldw, ldc, tm, tn);      \
        else                                                                                                                        \
            hipLaunchKernelGGL((k_gemm_f32<E, 2>), grid, block, 0, s, A, W, C, bias, residual, M, N, K, lda, ldw, ldc, tm, tn);      \
        break;
    switch (epilogue) {
        D3D_F32_CASE(F_NONE)
        D3D_F32_CASE(F_BIAS)
        D3D_F32_CASE(F_BIAS_GELU)
        D3D_F32_CASE(F_BIAS_RES)
        D3D_F32_CASE(F_BIAS_QGELU)
        D3D_F32_CASE(F_RES)
        default:
            d3d_set_error_("d3d_gemm_nt_f32: epilogue 0 none, 1 bias, 2 bias + GELU, 3 bias + residual, 4 bias + QuickGELU, 5 residual");
            return D3D_EINVAL;
    }
#undef D3D_F32_CASE
    D3D_LAUNCH_CHECK();
}

int32_t d3d_linear_smallk_f32(const float* x, const float* W, const float* bias, float* y, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldy,
                              int32_t gelu, void* stream) {
    if (M <= 0) return D3D_OK;
    if (K < 1 || K > 8 || N % 4 != 0 || (ldy & 3)) {
        d3d_set_error_("d3d_linear_smallk_f32: 1 <= K <= 8, N % 4 == 0, ldy % 4 == 0");
        return D3D_EINVAL;
    }
    dim3 grid((unsigned)((N / 4 + 63) / 64), (unsigned)((M + SMALLK_ROWS - 1) / SMALLK_ROWS)), block(256);
    if (gelu)
        hipLaunchKernelGGL((k_linear_smallk<F_BIAS_GELU>), grid, block, 0, (hipStream_t)stream, x, W, bias, y, M, N, K, ldx, ldy);
    else
        hipLaunchKernelGGL((k_linear_smallk<F_BIAS>), grid, block, 0, (hipStream_t)stream, x, W, bias, y, M, N, K, ldx, ldy);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_linear_smalln_f32(const float* x, const float* W, const float* bias, float* y, int32_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldy,
                              void* stream) {
    if (M <= 0) return D3D_OK;
    if (N < 1 || N > 8 || K % 4 != 0 || (ldx & 3)) {
        d3d_set_error_("d3d_linear_smalln_f32: 1 <= N <= 8, K % 4 == 0, ldx % 4 == 0");
        return D3D_EINVAL;
    }
    hipLaunchKernelGGL(k_linear_smalln, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, bias, y, M, N, K, ldx, ldy);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_layer_norm_f32(const float* x, const float* residual, const float* w, const float* b, float* y, int32_t rows, int32_t D, int64_t ldx,
                           int64_t ldr, int64_t ldy, float eps, int32_t gelu, void* stream) {
    if (rows <= 0) return D3D_OK;
    if (D % 4 != 0 || D > 3072 || (ldx & 3) || (ldy & 3) || (residual && (ldr & 3))) {
        d3d_set_error_("d3d_layer_norm_f32: D % 4 == 0, D <= 3072, row strides % 4 == 0");
        return D3D_EINVAL;
    }
    const int nch = (D + 255) / 256;
    dim3 grid((rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define D3D_LN_CASE(N)                                                                                                                      \
    if (nch <= N) {                                                                                                                         \
        hipLaunchKernelGGL((k_layer_norm_f32<N>), grid, block, 0, s, x, residual, w, b, y, rows, D, ldx, ldr, ldy, eps, gelu);                 \
        D3D_LAUNCH_CHECK();                                                                                                                 \
    }
    D3D_LN_CASE(3)
    D3D_LN_CASE(6)
    D3D_LN_CASE(12)
#undef D3D_LN_CASE
    d3d_set_error_("d3d_layer_norm_f32: unsupported width");
    return D3D_EINVAL;
}

}  // extern "C"
