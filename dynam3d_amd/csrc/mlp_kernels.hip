// mlp_kernels.hip -- the tinycudann-style small MLP as ONE kernel per network (north_star: "a fused ... MLP that replaces tinycudann";
// reference: `tcnn.Network(otype="CutlassMLP")`, PRE-FF:221-243, evaluated at PRE-FF:484, 488 on 1 152 rows per rendered view).
//
// A workgroup (8 waves) owns a slab of 64 rows and walks ALL layers with the slab's activations resident in LDS (64 x <= 896 values,
// 16 bit); only the weights stream (from L2: 1.2 MB per 768 x 768 layer, shared by every workgroup) and only what the caller asks for
// is written to HBM.  Per layer a wave owns a strip of output columns (N / 16 column tiles dealt to the 8 waves) for all 64 rows:
//   D^T = W X^T on v_mfma_f32_16x16x32: A = a W fragment (16 output columns x 32 k) straight from global memory in MFMA layout (16
//   bytes per lane, 64-byte runs per weight row), B = an X fragment (16 rows x 32 k) from LDS (ds_read_b128); a lane ends with 4
//   CONSECUTIVE output columns of one row -> the activation is applied on the float32 accumulators, the result is rounded ONCE to
//   16 bit (the rounding points of the unfused path: d3d_gemm_nt epilogues 0 / 7 / 8) and written back to the slab with 8-byte stores.
// The same kernel runs the DATA-GRADIENT chain of the backward pass: "weights" = the transposed matrices, epilogue = the LeakyReLU
// derivative taken from the saved forward activation of the layer below (mode 2), every layer's gradient also stored for the weight-
// gradient GEMMs.  K order of the accumulation = ascending 32-steps, like the GEMM kernel's: fused and unfused results are bit-identical.
//
// MEASURED (tools/bench_render.py, MI355X): the 768-768-768-769 network on 1 152 rows: 119 us fused against 37 us as three d3d_gemm_nt
// launches; on 9 216 rows 125 against 50 us.  Keeping the slab on chip makes every workgroup stream ALL the weights -- 3.5 MB per 64 rows
// (LDS holds no more rows at 16 bit) -- and one CU pulls ~30 GB/s through its L1 with these 64-byte row pieces (the same time with 8 or
// 16 waves per workgroup, with 18 or 144 workgroups: neither latency hiding nor L2 bandwidth is the limit), while the GEMM launches
// share each weight tile among 256 rows.  The fused kernel is therefore opt-in (D3D_MLP_FUSED=1 / tcnn.FUSED = True); the default is the
// per-layer GEMM path with the activations written once per layer (2.7 MB at 1 152 rows: cache-resident).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <mutex>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float4v = __attribute__((ext_vector_type(4))) float;

constexpr int ROWS = 64, NTH = 512, NWAVE = 8, MAXW = 896, LDX = MAXW + 8, MAXL = 4, MAXCT = (MAXW / 16 + NWAVE - 1) / NWAVE;   // 7 column tiles per wave

struct Layer {
    const uint16_t* W;       // (N, K) row-major, rows zero-padded to N
    const uint16_t* aux;     // mode 2: saved forward activation (rows, >= N), else null
    uint16_t* out;           // optional HBM copy of this layer's output (rows, >= N)
    int64_t ld_aux, ld_out;
    int K, N, mode;          // mode 0: none, 1: LeakyReLU(0.01), 2: multiply by LeakyReLU'(aux)
};
struct Args {
    Layer L[MAXL];
    int n_layers;
};

template <bool BF16>
__device__ __forceinline__ float4v mfma16(const uint4& a, const uint4& b, float4v c) {
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&a), *reinterpret_cast<const half8*>(&b), c, 0, 0, 0);
}
template <bool BF16>
__device__ __forceinline__ float to_f32(uint16_t v) {
    if constexpr (BF16) return __uint_as_float((uint32_t)v << 16);
    else return __half2float(*reinterpret_cast<const __half*>(&v));
}
template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (BF16) {
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
        return *reinterpret_cast<const uint32_t*>(&r);
    } else {
        const __half2 h = __floats2half2_rn(lo, hi);
        return *reinterpret_cast<const uint32_t*>(&h);
    }
}

template <bool BF16>
__global__ void __launch_bounds__(NTH)
k_mlp_fused(const uint16_t* __restrict__ x, int64_t ldx, int n_rows, int n_in, Args a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t X[];            // [ROWS][LDX]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    // ---- the slab's input rows -> LDS (zero rows behind n_rows) ------------------------------------------------------------------------------
    {
        const int ch = n_in / 8;
        for (int c = tid; c < ROWS * ch; c += NTH) {
            const int r = c / ch, p = c - r * ch;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row0 + r < n_rows) v = *reinterpret_cast<const uint4*>(x + (int64_t)(row0 + r) * ldx + p * 8);
            *reinterpret_cast<uint4*>(X + r * LDX + p * 8) = v;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
        const Layer L = a.L[l];
        const int K = L.K, N = L.N;
        const int n_ct = N / 16, cpw = (n_ct + NWAVE - 1) / NWAVE;
        const int ct0 = wave * cpw;
        const int my_ct = max(0, min(cpw, n_ct - ct0));                                            // wave-uniform
        float4v acc[4][MAXCT];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int c = 0; c < MAXCT; ++c) acc[rt][c] = float4v{0.f, 0.f, 0.f, 0.f};
        // W fragment of column tile c, k-step k0: lane holds W[(ct0 + c) * 16 + fi][k0 + fg * 8 .. + 7]
        const uint16_t* wp = L.W + (int64_t)(ct0 * 16 + fi) * K + fg * 8;
        const uint16_t* xp = X + fi * LDX + fg * 8;
        uint4 wf[MAXCT], wn[MAXCT];
#pragma unroll
        for (int c = 0; c < MAXCT; ++c) wf[c] = c < my_ct ? *reinterpret_cast<const uint4*>(wp + (int64_t)c * 16 * K) : make_uint4(0, 0, 0, 0);
        for (int k0 = 0; k0 < K; k0 += 32) {
            const bool more = k0 + 32 < K;
#pragma unroll
            for (int c = 0; c < MAXCT; ++c) wn[c] = (more && c < my_ct) ? *reinterpret_cast<const uint4*>(wp + (int64_t)c * 16 * K + k0 + 32) : make_uint4(0, 0, 0, 0);
            uint4 xf[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) xf[rt] = *reinterpret_cast<const uint4*>(xp + rt * 16 * LDX + k0);
#pragma unroll
            for (int c = 0; c < MAXCT; ++c) {
                if (c < my_ct) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) acc[rt][c] = mfma16<BF16>(wf[c], xf[rt], acc[rt][c]);
                }
            }
#pragma unroll
            for (int c = 0; c < MAXCT; ++c) wf[c] = wn[c];
        }
        __syncthreads();                                       // every wave has read the slab for this layer
        // ---- epilogue: lane holds out[row rt*16 + fi][col (ct0 + c)*16 + fg*4 + r] -------------------------------------------------------
#pragma unroll
        for (int c = 0; c < MAXCT; ++c) {
            if (c < my_ct) {
                const int col = (ct0 + c) * 16 + fg * 4;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int r_ = rt * 16 + fi, grow = row0 + r_;
                    float4v v = acc[rt][c];
                    if (L.mode == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
                    } else if (L.mode == 2) {
                        uint2 hh = make_uint2(0, 0);
                        if (grow < n_rows) hh = *reinterpret_cast<const uint2*>(L.aux + (int64_t)grow * L.ld_aux + col);
                        const uint16_t* hp = reinterpret_cast<const uint16_t*>(&hh);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = to_f32<BF16>(hp[r]) > 0.f ? v[r] : 0.01f * v[r];
                    }
                    const uint2 o = make_uint2(pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3]));
                    *reinterpret_cast<uint2*>(X + r_ * LDX + col) = o;
                    if (L.out && grow < n_rows) *reinterpret_cast<uint2*>(L.out + (int64_t)grow * L.ld_out + col) = o;
                }
            }
        }
        __syncthreads();                                       // the slab now holds this layer's output
    }
}

}  // namespace

extern "C" {

// Fused evaluation of a chain of up to 4 bias-free layers y_l = f_l(y_{l-1} W_l^T) over rows of 16-bit activations (see the file header).
// widths[0 .. n_layers]: input width, then every layer's (padded) output width; all % 16 == 0, <= 896, contraction widths % 32 == 0.
// weights[l]: device pointer, (widths[l+1], widths[l]) row-major.  modes[l]: 0 none, 1 LeakyReLU(0.01), 2 multiply by LeakyReLU'(aux[l])
// (aux[l]: (rows, ld_aux[l]) saved activation whose SIGN selects the slope).  outs[l]: optional (rows, ld_outs[l]) HBM copy of layer l's
// output (the last layer's is required).  Host arrays are read during the call only.
int32_t d3d_mlp_fused(const void* x, int64_t ldx, int64_t n_rows, int32_t n_layers, const int32_t* widths, const void* const* weights,
                      const int32_t* modes, const void* const* aux, const int64_t* ld_aux, void* const* outs, const int64_t* ld_outs, int32_t dtype,
                      void* stream) {
    if (n_rows <= 0) return D3D_OK;
    if (n_layers < 1 || n_layers > MAXL || n_rows > INT32_MAX || !outs[n_layers - 1]) {
        d3d_set_error_("d3d_mlp_fused: 1..4 layers, the last layer's output pointer is required");
        return D3D_EINVAL;
    }
    Args a;
    a.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        const int K = widths[l], N = widths[l + 1];
        if (K % 32 || N % 16 || K > MAXW || N > MAXW || K <= 0 || N <= 0 || (modes[l] < 0 || modes[l] > 2) || (modes[l] == 2 && !aux[l]) ||
            (outs[l] && (ld_outs[l] & 3)) || (modes[l] == 2 && (ld_aux[l] & 3))) {
            d3d_set_error_("d3d_mlp_fused: widths must be multiples of 32 (inputs) / 16 (outputs) and <= 896; mode 2 needs aux; row strides % 4 == 0");
            return D3D_EINVAL;
        }
        a.L[l] = Layer{(const uint16_t*)weights[l], (const uint16_t*)aux[l], (uint16_t*)outs[l], aux[l] ? ld_aux[l] : 0, outs[l] ? ld_outs[l] : 0, K, N, modes[l]};
    }
    if (ldx & 7) {
        d3d_set_error_("d3d_mlp_fused: input row stride must be a multiple of 8 elements");
        return D3D_EINVAL;
    }
    const size_t sh = (size_t)ROWS * LDX * sizeof(uint16_t);                    // 113 KiB
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    });
    D3D_HIP(attr_err);
    const dim3 grid((unsigned)((n_rows + ROWS - 1) / ROWS)), block(NTH);
    if (dtype == 0) hipLaunchKernelGGL(k_mlp_fused<true>, grid, block, sh, (hipStream_t)stream, (const uint16_t*)x, ldx, (int)n_rows, widths[0], a);
    else hipLaunchKernelGGL(k_mlp_fused<false>, grid, block, sh, (hipStream_t)stream, (const uint16_t*)x, ldx, (int)n_rows, widths[0], a);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
