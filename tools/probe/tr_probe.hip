// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/tr_probe tools/probe/tr_probe.hip ; run on the GPU box: tools/probe/tr_probe
// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds value = row*256 + col (16-bit), 64 rows x 64 cols row-major.
// Lane l (group g = l>>4, j = l&15) reads at row = 4*g + (j>>2), col = (j&3)*4  -> prints the 4 values each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = (uint16_t)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const uint32_t addr = (uint32_t)(uintptr_t)(tile) + ((4 * g + (j >> 2)) * 64 + (j & 3) * 4) * 2;
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[l * 4 + 0] = (uint16_t)(r.x & 0xffff);
    out[l * 4 + 1] = (uint16_t)(r.x >> 16);
    out[l * 4 + 2] = (uint16_t)(r.y & 0xffff);
    out[l * 4 + 3] = (uint16_t)(r.y >> 16);
}
int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int i = 0; i < 4; ++i) printf(" (r%2d,c%2d)", h[l * 4 + i] / 256, h[l * 4 + i] % 256);
        printf("\n");
    }
    return 0;
}
