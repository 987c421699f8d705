// segment_kernels.hip -- an ON-DEVICE class-agnostic segmenter for `Feature_Fields.get_patch_segm` (VLN-FF:400-430).
//
// The reference segments every RGB frame with FastSAM (a YOLOv8-seg network whose weights, FastSAM.pt, are not available offline) and
// only consumes the MASKS: "last mask wins" label image -> nearest 24 x 24 resize -> dense relabel (d3d_patch_segm_from_masks, a6).
// Until a trained network can be validated, this kernel is the mask generator that keeps the whole RGB-D -> tokens step on the
// device behind the same callable: grid-seeded colour + position clustering (SLIC-style k-means over all pixels of the frame, K = gx * gy
// seeds, a few Lloyd iterations), one workgroup per image.  Everything is DETERMINISTIC and restated bit for bit by oracle/segment_ref.py:
// cluster sums are INTEGER atomics in LDS (order-independent), centres = float32(sum) / float32(count), the distance is evaluated in a
// fixed float32 operation order (compiled with -ffp-contract=off), ties go to the lowest cluster index.
// Output: one {0,1} mask per cluster (clusters that lost all their pixels give an all-zero mask, which the relabel step drops) and,
// optionally, the label image.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dynam3d_hip.h"
#include "d3d_common.h"

namespace {

constexpr int SEG_MAXK = 64, SEG_NT = 1024;

__global__ void __launch_bounds__(SEG_NT)
k_segment_slic(const uint8_t* __restrict__ rgb, int H, int W, int gx, int gy, int iters, float w_xy, uint8_t* __restrict__ masks,
               int32_t* __restrict__ labels) {
    __shared__ float cen[SEG_MAXK][5];                  // r, g, b, x, y
    __shared__ int sums[SEG_MAXK][6];                   // r, g, b, x, y, count
    const int img = blockIdx.x, tid = threadIdx.x, K = gx * gy, HW = H * W;
    const uint8_t* im = rgb + (int64_t)img * HW * 3;
    if (tid < K) {                                       // seeds: cell centres of a gy x gx grid, colour of the pixel under the seed
        const int i = tid / gx, j = tid % gx;
        const float sx = ((float)j + 0.5f) * (float)W / (float)gx, sy = ((float)i + 0.5f) * (float)H / (float)gy;
        const int px = min(W - 1, (int)sx), py = min(H - 1, (int)sy);
        const uint8_t* p = im + ((int64_t)py * W + px) * 3;
        cen[tid][0] = (float)p[0]; cen[tid][1] = (float)p[1]; cen[tid][2] = (float)p[2];
        cen[tid][3] = sx; cen[tid][4] = sy;
    }
    __syncthreads();
    for (int it = 0; it <= iters; ++it) {                // `iters` Lloyd updates, then the final assignment
        const bool last = it == iters;
        if (tid < K) {
#pragma unroll
            for (int c = 0; c < 6; ++c) sums[tid][c] = 0;
        }
        __syncthreads();
        for (int p = tid; p < HW; p += SEG_NT) {
            const int y = p / W, x = p - y * W;
            const float r = (float)im[p * 3], g = (float)im[p * 3 + 1], b = (float)im[p * 3 + 2];
            const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
            float best = 3.4e38f;
            int bk = 0;
            for (int k = 0; k < K; ++k) {
                const float dr = r - cen[k][0], dg = g - cen[k][1], db = b - cen[k][2], dx = fx - cen[k][3], dy = fy - cen[k][4];
                const float d = ((dr * dr + dg * dg) + db * db) + w_xy * (dx * dx + dy * dy);
                if (d < best) { best = d; bk = k; }
            }
            if (!last) {
                atomicAdd(&sums[bk][0], (int)im[p * 3]); atomicAdd(&sums[bk][1], (int)im[p * 3 + 1]); atomicAdd(&sums[bk][2], (int)im[p * 3 + 2]);
                atomicAdd(&sums[bk][3], x); atomicAdd(&sums[bk][4], y); atomicAdd(&sums[bk][5], 1);
            } else {
                if (labels) labels[(int64_t)img * HW + p] = bk;
                for (int k = 0; k < K; ++k) masks[((int64_t)img * K + k) * HW + p] = (uint8_t)(k == bk);
            }
        }
        __syncthreads();
        if (!last && tid < K && sums[tid][5] > 0) {
            const float n = (float)sums[tid][5];
            cen[tid][0] = (float)sums[tid][0] / n; cen[tid][1] = (float)sums[tid][1] / n; cen[tid][2] = (float)sums[tid][2] / n;
            cen[tid][3] = (float)sums[tid][3] / n + 0.5f; cen[tid][4] = (float)sums[tid][4] / n + 0.5f;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int32_t d3d_segment_slic(const uint8_t* rgb, int32_t n_img, int32_t H, int32_t W, int32_t gx, int32_t gy, int32_t iters, float compactness,
                         uint8_t* masks, int32_t* labels, void* stream) {
    if (n_img <= 0) return D3D_OK;
    if (gx < 1 || gy < 1 || gx * gy > SEG_MAXK || iters < 0 || H < 1 || W < 1 || (int64_t)H * W * 255 > 2000000000ll) {
        d3d_set_error_("d3d_segment_slic: 1 <= gx * gy <= 64 seeds, iters >= 0, H * W <= 7.8 M pixels (integer cluster sums)");
        return D3D_EINVAL;
    }
    const float S = 0.5f * ((float)W / (float)gx + (float)H / (float)gy);           // seed spacing
    const float w_xy = (compactness * compactness) / (S * S);
    hipLaunchKernelGGL(k_segment_slic, dim3(n_img), dim3(SEG_NT), 0, (hipStream_t)stream, rgb, H, W, gx, gy, iters, w_xy, masks, labels);
    D3D_LAUNCH_CHECK();
}

}  // extern "C"
