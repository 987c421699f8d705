// mfma_lds_probe.hip -- does LDS -> VGPR traffic overlap with MFMA issue on the SAME SIMD?
//
// Question behind it ("where the other half goes"): the 256x256x64 GEMM tile needs 24 ds_read_b128 per wave
// and K tile next to 64 v_mfma_f32_16x16x32_bf16.  Per SIMD that is 2 x 64 MFMAs and 2 x 24 KiB of LDS returns per K tile.  If the
// two overlap the loop is MFMA-bound; if they ADD, the matrix pipe can at best be busy t_mfma / (t_mfma + t_lds) of the time.
// One workgroup of 8 waves per CU (two waves per SIMD: wave w and w + 4, like the GEMM), 256 workgroups, every wave timed with
// s_memtime around its loop; the table prints the SLOWEST wave of each role in shader cycles per step (one step = 64 MFMAs and / or
// 24 ds_read_b128 issued back to back and consumed after one s_waitcnt):
//
//   mode 0: waves 0-3: MFMA steps, waves 4-7 idle                      mode 1: waves 4-7: LDS steps, waves 0-3 idle
//   mode 2: waves 0-3 MFMA steps  ||  waves 4-7 LDS steps (partners)   mode 3: every wave: 24 reads then 64 MFMAs per step
//   mode 4: all 8 waves MFMA steps (two waves share each SIMD's matrix pipe)
//   mode 5: every wave: 64 MFMAs with the 24 reads of the next (half) step interleaved one per ~3 MFMAs
//   mode 6 / 7: pure MFMA streams (two waves per SIMD) on operands with RANDOM mantissas, v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16: what
//               the matrix pipe sustains on real data under the chip's power limit (clock from s_memtime / wall time)
//
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe mfma_lds_probe.hip ; run: ./mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float4v = __attribute__((ext_vector_type(4))) float;

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
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 1024 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    float4v acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = float4v{0.f, 0.f, 0.f, 0.f};
    uint4 frag[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) frag[i] = make_uint4(0x3f803f80u + lane, 0x3f813f80u + i * 77u, 0x3f803f85u + lane * 3u, 0x3f833f80u);
    // conflict-free ds_read_b128: lane-linear 16-byte chunks, 1 KiB per instruction, 24 KiB per wave and step
    const uint16_t* base = smem + lane * 8 + (wave & 3) * 12288;
    const bool do_mfma = MODE == 3 || MODE == 4 || ((MODE == 0 || MODE == 2) && wave < 4);
    const bool do_lds = MODE == 3 || ((MODE == 1 || MODE == 2) && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (do_lds) {
            uint4 v[24];
#pragma unroll
            for (int i = 0; i < 24; ++i) v[i] = *reinterpret_cast<const uint4*>(base + i * 512);
#pragma unroll
            for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));   // all 24 in flight, ONE wait
#pragma unroll
            for (int i = 0; i < 24; ++i) frag[i].w ^= v[i].x & 1u;
        }
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&frag[(i + r) % 24]),
                                                                     *reinterpret_cast<const bf16x8*>(&frag[(i * 5 + r) % 24]), acc[i], 0, 0, 0);
        }
        if constexpr (MODE == 5) {
            // software-pipelined half steps: 32 MFMAs on fragment set `cur` (12 registers of 16 B) while the 12 ds_read_b128 of the
            // NEXT half step land in the other set -- one read issued after every ~3rd MFMA, no burst
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4* cur = frag + (h ? 12 : 0);
                uint4* nxt = frag + (h ? 0 : 12);
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&cur[i % 12]),
                                                                          *reinterpret_cast<const bf16x8*>(&cur[(i * 5 + 1) % 12]), acc[i & 15], 0, 0, 0);
                    if (i % 3 == 0 && i / 3 < 12) nxt[i / 3] = *reinterpret_cast<const uint4*>(base + (h * 12 + i / 3) * 512);   // next half step's operands
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                    // 1 MFMA ...
                    if (i % 3 == 0 && i / 3 < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // ... then 1 DS read
                }
            }
        }
    }
    if constexpr (MODE == 6 || MODE == 7) {
        // operands with random mantissas (|x| in [0.5, 2)): the toggling of real data, for the power-limited clock
        uint4 rf[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            uint32_t h = (uint32_t)(tid * 2654435761u) ^ (uint32_t)(i * 40503u) ^ (uint32_t)(blockIdx.x * 97u);
            auto nx = [&]() { h = h * 1664525u + 1013904223u; const uint32_t a = 0x3f00u | ((h >> 9) & 0x00ffu) | ((h >> 3) & 0x8000u);
                              h = h * 1664525u + 1013904223u; const uint32_t b = 0x3f00u | ((h >> 9) & 0x00ffu) | ((h >> 3) & 0x8000u); return a | (b << 16); };
            rf[i] = make_uint4(nx(), nx(), nx(), nx());
        }
        using float16v = __attribute__((ext_vector_type(16))) float;
        float16v big[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 6) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&rf[(i + r) % 24]),
                                                                         *reinterpret_cast<const bf16x8*>(&rf[(i * 5 + r) % 24]), acc[i], 0, 0, 0);
            } else {
                // the same 64 x 16384 flop per step as 32 instructions of 32x32x16
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        big[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&rf[(i + r) % 24]),
                                                                         *reinterpret_cast<const bf16x8*>(&rf[(i * 5 + r) % 24]), big[i], 0, 0, 0);
            }
            if ((it & 63) == 63) {          // keep the accumulators bounded
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] *= 1e-3f;
#pragma unroll
                for (int i = 0; i < 8; ++i) big[i] *= 1e-3f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 15][0] += big[i][0] + big[i][7] + big[i][15];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 24; ++i) s += (float)(frag[i].w & 3u);
    asm volatile("" : "+v"(s));
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (s == 12345.678f) out[blockIdx.x * 512 + tid] = s;
}

struct Res {
    double ms, cyc_mfma, cyc_lds;      // kernel wall time; slowest wave of each role, shader cycles per step
};

template <int MODE>
int run(const char* name, float* out, unsigned long long* cyc_d, int iters, Res* res) {
    const size_t sh = 128 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(256), dim3(512), sh, 0, out, cyc_d, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(256), dim3(512), sh, 0, out, cyc_d, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(256 * 8);
    CHECK(hipMemcpy(h.data(), cyc_d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost));
    unsigned long long m_lo = 0, m_hi = 0;
    for (int b2 = 0; b2 < 256; ++b2)
        for (int w = 0; w < 8; ++w) {
            if (w < 4) m_lo = std::max(m_lo, h[b2 * 8 + w]);
            else m_hi = std::max(m_hi, h[b2 * 8 + w]);
        }
    res->ms = best;
    res->cyc_mfma = (MODE == 0 || MODE == 2) ? (double)m_lo / iters : ((MODE >= 3) ? (double)std::max(m_lo, m_hi) / iters : 0.0);
    res->cyc_lds = (MODE == 1 || MODE == 2) ? (double)m_hi / iters : (MODE == 3 ? (double)std::max(m_lo, m_hi) / iters : 0.0);
    printf("%-36s %8.3f ms   cycles/step: mfma waves %8.1f   lds waves %8.1f   (counter rate %.3f GHz)\n", name, best, res->cyc_mfma, res->cyc_lds,
           (double)std::max(m_lo, m_hi) / (best * 1e-3) / 1e9);
    if (MODE >= 6) printf("   -> %.0f TFLOP/s sustained (256 CUs x 8 waves x 64 x 16384 flop per step)\n", 256.0 * 8 * 64 * 16384.0 * iters / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CHECK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
    const int iters = 4000;
    Res r0, r1, r2, r3, r4, r5;
    if (run<0>("0: MFMA, one wave per SIMD", out, cyc, iters, &r0)) return 1;
    if (run<1>("1: LDS reads, one wave per SIMD", out, cyc, iters, &r1)) return 1;
    if (run<2>("2: MFMA wave || LDS wave per SIMD", out, cyc, iters, &r2)) return 1;
    if (run<3>("3: 2 waves/SIMD, reads then MFMAs", out, cyc, iters, &r3)) return 1;
    if (run<4>("4: MFMA, two waves per SIMD", out, cyc, iters, &r4)) return 1;
    if (run<5>("5: 2 waves/SIMD, reads INTERLEAVED", out, cyc, iters, &r5)) return 1;
    Res r6, r7;
    if (run<6>("6: 16x16x32, RANDOM operands", out, cyc, 20000, &r6)) return 1;
    if (run<7>("7: 32x32x16, RANDOM operands", out, cyc, 20000, &r7)) return 1;
    if (run<6>("6: 16x16x32, RANDOM operands (again)", out, cyc, 20000, &r6)) return 1;
    if (run<7>("7: 32x32x16, RANDOM operands (again)", out, cyc, 20000, &r7)) return 1;
    printf("\nwall time per step (64 MFMAs, 24 KiB of ds_read_b128 per wave), ns -- ratios are clock independent:\n");
    const double n0 = r0.ms * 1e6 / iters, n1 = r1.ms * 1e6 / iters, n2 = r2.ms * 1e6 / iters, n3 = r3.ms * 1e6 / iters, n4 = r4.ms * 1e6 / iters;
    printf("  one MFMA wave per SIMD alone   %7.1f ns = %.2f ns per MFMA\n", n0, n0 / 64);
    printf("  two MFMA waves per SIMD        %7.1f ns for 128 MFMAs = %.2f ns per MFMA (the matrix pipe's own rate)\n", n4, n4 / 128);
    printf("  one LDS wave per SIMD alone    %7.1f ns for 4 x 24 KiB per CU = %.0f GB/s per CU\n", n1, 4 * 24576.0 / n1);
    printf("  partners (mode 2)              %7.1f ns: overlap = %.2f (1 = the shorter one is hidden, 0 = the times add)\n", n2, (n0 + n1 - n2) / std::min(n0, n1));
    printf("  GEMM-like step (mode 3)        %7.1f ns for 2 x (24 reads + 64 MFMAs) per SIMD; matrix-pipe-only time %.1f ns -> pipe busy at most %.0f %%;\n"
           "                                 pipe + LDS-alone times: %.1f ns (additive model) vs max %.1f ns (perfect overlap)\n",
           n3, n4, 100 * n4 / n3, n4 + 2 * n1, std::max(n4, 2 * n1));
    const double n5 = r5.ms * 1e6 / iters;
    printf("  interleaved reads (mode 5)     %7.1f ns for 2 x (24 reads + 64 MFMAs) per SIMD, one read behind every 3rd MFMA, fragments double-buffered\n"
           "                                 by half steps -> pipe busy at most %.0f %% (burst reads, mode 3: %.0f %%)\n", n5, 100 * n4 / n5, 100 * n4 / n3);
    return 0;
}
