// mfma_valu_probe.hip -- can VALU work (v_exp_f32, v_fma_f32) issue on a SIMD while its matrix pipe is busy?
//
// Question behind it: a flash-attention key tile is ~900 cycles of v_mfma_f32_32x32x16 and ~900 cycles of softmax
// VALU per wave; with two waves per SIMD a wave-tile measures ~2 250 cycles.  If MFMA and VALU instructions of DIFFERENT waves (or of
// one wave, interleaved) overlap, the floor is max(MFMA, VALU) ~ 900; if they serialise on the SIMD's issue port it is their sum.
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD), 256 workgroups, s_memtime around each wave's loop, slowest wave of
// each role in shader cycles per step.  A step = 16 MFMAs 32x32x16 bf16 (16 x 32 = 512 pipe cycles) and / or NV VALU instructions.
//
//   mode 0: waves 0-3 MFMA steps, 4-7 idle                       mode 1: waves 4-7 VALU steps (64 v_exp_f32), 0-3 idle
//   mode 2: waves 0-3 MFMA  ||  waves 4-7 v_exp (partners on the same SIMD)
//   mode 3: every wave: 16 MFMAs then 64 v_exp per step (not interleaved in the source; independent of each other)
//   mode 4: every wave: MFMA, 4 v_exp, MFMA, 4 v_exp ... (interleaved by sched_group_barrier)
//   mode 5 / 6 / 7: as 1 / 2 / 4 with v_fma_f32 (full-rate VALU) instead of v_exp_f32 (quarter rate): 128 per step
//
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip ; run: ./mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float16v = __attribute__((ext_vector_type(16))) float;

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__);  \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <int MODE>
__global__ void __launch_bounds__(512, 2) k_probe(float* out, unsigned long long* cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr bool EXP = MODE <= 4;
    constexpr int NV = EXP ? 64 : 128;
    float16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = make_uint4(0x3c003c00u + lane, 0x3c013c00u + i * 77u, 0x3c003c05u + lane * 3u, 0x3c033c00u);
        fb[i] = make_uint4(0x3c003c00u + lane * 5u, 0x3c023c00u + i * 31u, 0x3c003c07u + lane, 0x3c013c02u);
    }
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = -0.001f * (float)(lane + i);
    const bool do_mfma = MODE == 3 || MODE == 4 || MODE == 7 || ((MODE == 0 || MODE == 2 || MODE == 6) && wave < 4);
    const bool do_valu = MODE == 3 || MODE == 4 || MODE == 7 || ((MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6) && wave >= 4);
    constexpr bool INTER = MODE == 4 || MODE == 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (INTER) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[m & 3]), *reinterpret_cast<const bf16x8*>(&fb[(m >> 2) & 3]),
                                                                      acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV / 16; ++q) {
                    const int i = (m * (NV / 16) + q) & 15;
                    if (EXP) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0001f;       // (the subtraction keeps the chain from saturating; it is a VALU op too)
                    else v[i] = __builtin_fmaf(v[i], 0.999f, 0.0001f);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, EXP ? 2 * (NV / 16) : NV / 16, 0);
            }
        } else {
            if (do_mfma) {
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[m & 3]), *reinterpret_cast<const bf16x8*>(&fb[(m >> 2) & 3]),
                                                                          acc[m & 3], 0, 0, 0);
            }
            if (do_valu) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const int i = q & 15;
                    if (EXP) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0001f;
                    else v[i] = __builtin_fmaf(v[i], 0.999f, 0.0001f);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
int run(const char* what, float* out, unsigned long long* cyc, int iters) {
    const int G = 256;
    hipLaunchKernelGGL(k_probe<MODE>, dim3(G), dim3(512), 0, 0, out, cyc, 8);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(G), dim3(512), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(G * 8);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long lo = 0, hi = 0;       // slowest wave of waves 0-3 / 4-7
    for (int g = 0; g < G; ++g)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) = std::max(w < 4 ? lo : hi, h[g * 8 + w]);
    printf("mode %d  %-72s waves 0-3: %7.1f cycles/step   waves 4-7: %7.1f cycles/step   (kernel %.3f ms)\n", MODE, what, (double)lo / iters, (double)hi / iters, ms);
    return 0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    const int iters = 2000;
    printf("step = 16 x v_mfma_f32_32x32x16_bf16 (512 pipe cycles) and / or 64 x (v_exp_f32 + v_sub_f32) resp. 128 x v_fma_f32; s_memtime-free: shader cycles\n");
    if (run<0>("waves 0-3 MFMA, 4-7 idle", out, cyc, iters)) return 1;
    if (run<1>("waves 4-7 v_exp, 0-3 idle", out, cyc, iters)) return 1;
    if (run<2>("waves 0-3 MFMA || waves 4-7 v_exp (same SIMDs)", out, cyc, iters)) return 1;
    if (run<3>("every wave: 16 MFMA then 64 v_exp (source order), 2 waves / SIMD", out, cyc, iters)) return 1;
    if (run<4>("every wave: MFMA / 4 v_exp interleaved (sched_group_barrier), 2 waves / SIMD", out, cyc, iters)) return 1;
    if (run<5>("waves 4-7 v_fma, 0-3 idle", out, cyc, iters)) return 1;
    if (run<6>("waves 0-3 MFMA || waves 4-7 v_fma (same SIMDs)", out, cyc, iters)) return 1;
    if (run<7>("every wave: MFMA / 8 v_fma interleaved, 2 waves / SIMD", out, cyc, iters)) return 1;
    return 0;
}
