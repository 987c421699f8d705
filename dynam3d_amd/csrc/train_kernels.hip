// train_kernels.hip -- the small kernels around the GEMMs of the tcnn-MLP backward pass (first slice):
//
//   d3d_transpose_pad16   (R, C) 16-bit row-major -> (C, Rp) with Rp >= R a multiple of 64, columns [R, Rp) zero.  The weight
//                         gradient dW[N,K] = sum_m dz[m,n] h[m,k] is the NT GEMM  dz^T[N,Mp] . (h^T[K,Mp])^T  (both operands
//                         contiguous along the reduction dimension m), so dz and h are transposed once, through LDS.
//   d3d_lrelu_bwd         dz = dy * (y > 0 ? 1 : 0.01): the LeakyReLU(0.01) of a layer whose OUTPUT y is kept (sign(y) = sign(z)).
//                         (Between layers the same factor is the GEMM epilogue 8 of the data-gradient GEMM.)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

// 64 x 64 tile per 256-thread workgroup through a padded LDS tile (row pitch 66 elements: conflict-free column reads).
__global__ void __launch_bounds__(256)
k_transpose_pad16(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int C, int64_t ld_in, int Rp) {
    __shared__ uint16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < C) ? in[(int64_t)r * ld_in + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(int64_t)c * Rp + r] = tile[tx][ty * 16 + i];
    }
}

template <bool BF16>
__global__ void k_lrelu_bwd(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y, uint16_t* __restrict__ dz, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const uint4 a = reinterpret_cast<const uint4*>(dy)[i], b = reinterpret_cast<const uint4*>(y)[i];
    const uint16_t* ah = reinterpret_cast<const uint16_t*>(&a);
    const uint16_t* bh = reinterpret_cast<const uint16_t*>(&b);
    uint16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float g, v;
        if constexpr (BF16) {
            g = __uint_as_float((uint32_t)ah[j] << 16);
            v = __uint_as_float((uint32_t)bh[j] << 16);
        } else {
            g = __half2float(*reinterpret_cast<const __half*>(&ah[j]));
            v = __half2float(*reinterpret_cast<const __half*>(&bh[j]));
        }
        const float r = v > 0.f ? g : 0.01f * g;
        if constexpr (BF16) {
            uint32_t u = __float_as_uint(r);
            u += 0x7fffu + ((u >> 16) & 1u);
            o[j] = (uint16_t)(u >> 16);
        } else {
            const __half h = __float2half_rn(r);
            o[j] = *reinterpret_cast<const uint16_t*>(&h);
        }
    }
    reinterpret_cast<uint4*>(dz)[i] = *reinterpret_cast<const uint4*>(o);
}

}  // namespace

extern "C" {

int32_t d3d_transpose_pad16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int32_t Rp, void* stream) {
    if (R <= 0 || C <= 0) return D3D_OK;
    if (Rp < R || Rp % 64 != 0) {
        d3d_set_error_("d3d_transpose_pad16: Rp must be a multiple of 64 and >= R");
        return D3D_EINVAL;
    }
    dim3 grid((C + 63) / 64, Rp / 64), block(256);
    hipLaunchKernelGGL(k_transpose_pad16, grid, block, 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out, R, C, ld_in, Rp);
    D3D_LAUNCH_CHECK();
}

int32_t d3d_lrelu_bwd(const void* dy, const void* y, void* dz, int64_t n, int32_t dtype, void* stream) {
    if (n <= 0) return D3D_OK;
    if (n % 8) {
        d3d_set_error_("d3d_lrelu_bwd: element count must be a multiple of 8");
        return D3D_EINVAL;
    }
    const int64_t n8 = n / 8;
    dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    if (dtype == 0)
        hipLaunchKernelGGL(k_lrelu_bwd<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dz, n8);
    else
        hipLaunchKernelGGL(k_lrelu_bwd<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dz, n8);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
