// gemm_epilogue.h -- MFMA wrappers, 16-bit conversions, LDS-DMA staging helpers and the fused GEMM epilogues shared by
// gemm_kernels.hip (prefill GEMMs) and decode_kernels.hip (the persistent decode-token kernel).  Included INSIDE the
// including file's anonymous namespace.
#pragma once

using short8 = __attribute__((ext_vector_type(8))) short;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using float4v = __attribute__((ext_vector_type(4))) float;

enum Epi : int { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_QGELU = 2, EPI_BIAS_GELU = 3, EPI_RES = 4, EPI_BIAS_RES = 5, EPI_SWIGLU = 6, EPI_LRELU = 7,
                 EPI_LRELU_BWD = 8 };

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

// Operand tile (ROWS x 64, 16-bit) -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave
// instruction = 8 rows x 128 B, no VGPR round trip).  The image is lane-linear, so the ds_read bank-conflict
// swizzle is applied to the per-lane SOURCE chunk (chunk ^= row & 7) and undone in the read address.
// `rows_valid` clamps the row index: edge tiles re-read the last valid row (their outputs are never stored).
//
// Issued from inline asm on purpose: hipcc counts a builtin LDS-DMA as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY buffer, which serialises prefetch and compute.
// In asm the DMA is invisible to that bookkeeping; the kernels below wait for it explicitly (counted in tiles)
// and order it against the ds_reads with s_barrier.  M0 (LDS destination base) is saved/restored around the
// statement (cdna_hip_programming.md 5.7).
// Addressing: SGPR base (tile origin + K offset, advanced with scalar adds) + one loop-invariant 32-bit VGPR byte offset per
// piece (row * ld + swizzled chunk), so issuing a tile costs no vector ALU work inside the K loop.
template <int WAVES>
struct TileLanes {
    uint32_t off[4];     // wave w owns pieces w, w+WAVES, w+2*WAVES, w+3*WAVES  (4*WAVES pieces of 8 rows = 32*WAVES rows)
    __device__ __forceinline__ void init(int64_t ld, int row0, int rows_valid, int wave, int lane) {
        const int base_row = row0 < rows_valid ? row0 : rows_valid - 1;         // (offsets are relative to the clamped origin)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = (wave + p * WAVES) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (r & 7);
            int gr = row0 + r;
            gr = (gr < rows_valid ? gr : rows_valid - 1) - base_row;
            off[p] = (uint32_t)(((int64_t)gr * ld + c * 8) * 2);
        }
    }
};

// two pieces per wave instead of four: a 16 * WAVES-row operand tile (the 64-row W tile of the 128 x 64 kernel)
template <int WAVES>
__device__ __forceinline__ void stage_tile_dma2(const TileLanes<WAVES>& tl, const uint16_t* __restrict__ base, uint32_t lds_byte_addr, int wave) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr + (uint32_t)wave * 1024u);
    uint32_t keep;
    constexpr int STEP = WAVES * 1024;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_add_u32 m0, m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(tl.off[0]), "v"(tl.off[1]), "s"(dst), "i"(STEP), "s"(base)
        : "memory");
}

template <int WAVES>
__device__ __forceinline__ void stage_tile_dma(const TileLanes<WAVES>& tl, const uint16_t* __restrict__ base, uint32_t lds_byte_addr, int wave) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr + (uint32_t)wave * 1024u);
    uint32_t keep;
    constexpr int STEP = WAVES * 1024;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %7\n\t"
        "s_add_u32 m0, m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %7\n\t"
        "s_add_u32 m0, m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %7\n\t"
        "s_add_u32 m0, m0, %6\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %7\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(tl.off[0]), "v"(tl.off[1]), "v"(tl.off[2]), "v"(tl.off[3]), "s"(dst), "i"(STEP), "s"(base)
        : "memory");
}

// One 1 KiB piece (8 rows x 128 B) of an operand tile; `dst` = wave-uniform LDS byte address of the piece.
__device__ __forceinline__ void stage_piece_dma(uint32_t voff, const uint16_t* __restrict__ base, uint32_t dst) {
    const uint32_t d = __builtin_amdgcn_readfirstlane(dst);
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(d), "s"(base)
        : "memory");
}

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// Epilogue for one 16-column MFMA tile of one output row: this lane owns columns n16 + fg*4 .. +3.
// SWIGLU: `a` is the gate tile, `b` the matching up tile (weights interleaved per 16 rows); output column n16/2.
// Rounding points = the reference's module boundaries (include/dynam3d_hip.h "ROUNDING POINTS").  16-bit stores go through the
// hardware converters (v_cvt_pk_bf16_f32: two values per instruction, round-to-nearest-even, NaN-safe; v_cvt_f16_f32): with one
// workgroup per CU nothing overlaps the epilogue, and a hand-rolled integer bf16 rounding (6 VALU operations per value, three stores
// per SwiGLU output) cost gate_up_proj 7 % (543 against 505 us at M = 6912).
template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (BF16) {
        // fptrunc <2 x float> -> <2 x bfloat> selects v_cvt_pk_bf16_f32 (RNE, NaN-safe); NOT inline asm: the hazard recogniser does not
        // see through asm, and a conversion scheduled right behind the MFMA that produced its operand reads a stale register
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
        return *reinterpret_cast<const uint32_t*>(&r);
    } else {
        const __half2 h = __floats2half2_rn(lo, hi);
        return *reinterpret_cast<const uint32_t*>(&h);
    }
}

// store-and-reload of a pair of fp32 registers in the 16-bit dtype
template <bool BF16>
__device__ __forceinline__ void r16x2(float& a, float& b) {
    const uint32_t u = pack2<BF16>(a, b);
    if constexpr (BF16) {
        a = __uint_as_float(u << 16);
        b = __uint_as_float(u & 0xffff0000u);
    } else {
        const __half2 h = *reinterpret_cast<const __half2*>(&u);
        a = __low2float(h);
        b = __high2float(h);
    }
}

template <bool BF16>
__device__ __forceinline__ void r16x4(float* v) {
    r16x2<BF16>(v[0], v[1]);
    r16x2<BF16>(v[2], v[3]);
}

// epi_pack: the epilogue arithmetic of four neighbouring outputs of one row, packed as 4 x 16 bit.  DEFER (the LDS-transposed epilogue of
// the 256 x 256 kernel): the residual of EPI_RES / EPI_BIAS_RES is NOT added here -- the value returned is the linear's own 16-bit output
// R(acc [+ bias]), the residual is added after the transposition with row-contiguous 16-byte loads (same rounding points).
template <bool BF16, int EPI, bool DEFER>
__device__ __forceinline__ uint2 epi_pack(const float4v& a, const float4v& b, const uint16_t* __restrict__ bias,
                                          const uint16_t* __restrict__ residual, int m, int n16, int fg, int64_t ldc) {
    if constexpr (EPI == EPI_SWIGLU) {
        // HF Phi3MLP: gate_up = linear(x) [16-bit]; up * silu(gate) with silu's result and the product stored 16-bit
        float g[4] = {a[0], a[1], a[2], a[3]}, u[4] = {b[0], b[1], b[2], b[3]};
        r16x4<BF16>(g);
        r16x4<BF16>(u);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = act_silu(g[r]);
        r16x4<BF16>(g);
        uint2 o;
        o.x = pack2<BF16>(u[0] * g[0], u[1] * g[1]);
        o.y = pack2<BF16>(u[2] * g[2], u[3] * g[3]);
        return o;
    } else {
        const int n = n16 + fg * 4;
        float v[4] = {a[0], a[1], a[2], a[3]};
        if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
            const uint2 bb = *reinterpret_cast<const uint2*>(bias + n);
            const uint16_t* bp = reinterpret_cast<const uint16_t*>(&bb);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32<BF16>(bp[r]);
        }
        if constexpr (EPI == EPI_BIAS_QGELU) {          // x * sigmoid(1.702 * x) on 16-bit tensors (clip/model.py:162-164)
            r16x4<BF16>(v);
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = 1.702f * v[r];
            r16x4<BF16>(t);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = 1.0f / (1.0f + __expf(-t[r]));
            r16x4<BF16>(t);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= t[r];
        }
        if constexpr (EPI == EPI_BIAS_GELU) {
            r16x4<BF16>(v);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_gelu(v[r]);
        }
        if constexpr (EPI == EPI_LRELU) {          // tcnn CutlassMLP hidden activation (slope 0.01), PRE-FF:221-243
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.01f * v[r];
        }
        if constexpr (EPI == EPI_LRELU_BWD) {      // data gradient through the layer BELOW's LeakyReLU: dz = (dz_above W) * act'(h), `residual` = that layer's output h
            const uint2 hh = *reinterpret_cast<const uint2*>(residual + (int64_t)m * ldc + n);
            const uint16_t* hp = reinterpret_cast<const uint16_t*>(&hh);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = to_f32<BF16>(hp[r]) > 0.f ? v[r] : 0.01f * v[r];
        }
        if constexpr ((EPI == EPI_RES || EPI == EPI_BIAS_RES) && !DEFER) {
            const uint2 rr = *reinterpret_cast<const uint2*>(residual + (int64_t)m * ldc + n);
            const uint16_t* rp = reinterpret_cast<const uint16_t*>(&rr);
            r16x4<BF16>(v);                                                          // x + linear(...): the linear's output is a 16-bit tensor
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f32<BF16>(rp[r]);
        }
        uint2 o;
        o.x = pack2<BF16>(v[0], v[1]);
        o.y = pack2<BF16>(v[2], v[3]);
        return o;
    }
}

template <bool BF16, int EPI>
__device__ __forceinline__ void store4(const float4v& a, const float4v& b, uint16_t* __restrict__ C, const uint16_t* __restrict__ bias,
                                       const uint16_t* __restrict__ residual, int m, int n16, int fg, int64_t ldc) {
    const uint2 o = epi_pack<BF16, EPI, false>(a, b, bias, residual, m, n16, fg, ldc);
    const int n = EPI == EPI_SWIGLU ? n16 / 2 + fg * 4 : n16 + fg * 4;
    *reinterpret_cast<uint2*>(C + (int64_t)m * ldc + n) = o;
}

// a + r on two packed 16-bit values each, rounded once (the residual add of the transposed epilogue)
template <bool BF16>
__device__ __forceinline__ uint32_t add2_16(uint32_t a, uint32_t r) {
    return pack2<BF16>(to_f32<BF16>((uint16_t)(a & 0xffffu)) + to_f32<BF16>((uint16_t)(r & 0xffffu)),
                       to_f32<BF16>((uint16_t)(a >> 16)) + to_f32<BF16>((uint16_t)(r >> 16)));
}
