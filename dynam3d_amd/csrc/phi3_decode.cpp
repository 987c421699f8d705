// phi3_decode.cpp -- host side of one KV-cache decode token of the Phi-3 decoder stack: the seven launches of every layer
// (RMSNorm, qkv GEMM, RoPE + decode attention, o_proj + residual, RMSNorm, gate_up + SwiGLU, down_proj + residual) are issued
// from C++ in one call.  At 8 rows a launch runs for 5-60 us, so issuing ~260 of them per token from Python (ctypes + tensor
// allocation, ~15 us each) left the GPU idle most of the time.  Reference: the decoder layers under
// `llava.generate(..., do_sample=False)` (VLN-POL:463; HF Phi3DecoderLayer with use_cache).
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dynam3d_hip.h"

extern "C" void d3d_set_error_(const char* msg);
// decode_kernels.hip: the whole token as ONE cooperative launch (persistent workgroups, grid barriers between the phases)
extern "C" int32_t d3d_phi3_decode_persistent_ok(const d3d_phi3_decode_args* a);
extern "C" int32_t d3d_phi3_decode_token_persistent(const d3d_phi3_decode_args* a);

extern "C" int32_t d3d_phi3_decode_token(const d3d_phi3_decode_args* a) {
    if (!a || a->n_layers <= 0 || a->rows <= 0) return D3D_OK;
    const int32_t B = a->rows, Hd = a->hidden, H = a->heads, hd = a->head_dim, I = a->mlp, dt = a->dtype;
    const int64_t qkv_w = (int64_t)3 * H * hd;
    if (H * hd != Hd || a->rows > 16) {
        d3d_set_error_("d3d_phi3_decode_token: needs heads * head_dim == hidden (no grouped-query attention) and rows <= 16");
        return D3D_EINVAL;
    }
    if (d3d_phi3_decode_persistent_ok(a)) return d3d_phi3_decode_token_persistent(a);      // (D3D_DECODE_PERSISTENT=0: the launch-per-op path below)
    // RMSNorm inside the projection that follows it (d3d_gemm_nt_rmsnorm: 5 launches per layer instead of 7), bit-identical to norm + GEMM.
    // History (tools/bench_decode.py, ms per token against 3.46-3.50 unfused): normalising every activation fragment inside the K loop 3.76
    // (round 2); the rows normalised once per workgroup into LDS 3.67 (the extra 49 KB pushed the 16-wave workgroups to one per CU);
    // with that buffer overlaid on the wave-reduction buffer (two per CU again) 3.38 -- the default since round 4.  D3D_DECODE_FUSE_NORM=0
    // selects the seven-launch layer.  Rows 9-16 need twice the LDS for the normalised rows (98 KB at K = 3072): one workgroup per CU
    // again, the regime measured SLOWER than the seven-launch layer -- they take the seven launches.
    const char* fe = getenv("D3D_DECODE_FUSE_NORM");
    const bool fuse = !(fe && fe[0] == '0') && B <= 8 && Hd % 512 == 0 && qkv_w % 32 == 0 && (2 * I) % 32 == 0;
    void* s = a->stream;
    void* x = a->x;                       // (rows, hidden) in / out: the residual stream
    int32_t rc;
#define D3D_TRY(call)         \
    do {                      \
        rc = (call);          \
        if (rc != D3D_OK) return rc; \
    } while (0)
    for (int32_t l = 0; l < a->n_layers; ++l) {
        if (fuse) {
            D3D_TRY(d3d_gemm_nt_rmsnorm(x, a->n1[l], a->rms_eps, a->qkv_w[l], a->qkv, B, (int32_t)qkv_w, Hd, Hd, Hd, qkv_w, dt, 0, s));   // RMSNorm inside
        } else {
            D3D_TRY(d3d_norm(x, a->n1[l], nullptr, a->h, B, Hd, Hd, Hd, a->rms_eps, 1, dt, s));
            D3D_TRY(d3d_gemm_nt(a->h, a->qkv_w[l], a->qkv, nullptr, nullptr, B, (int32_t)qkv_w, Hd, Hd, Hd, qkv_w, dt, 0, s));
        }
        D3D_TRY(d3d_decode_attention(a->qkv, a->prompt_qkv[l], a->cu_seqlens, (char*)a->knew + (int64_t)l * a->cache_layer_stride_bytes,
                                     (char*)a->vnew + (int64_t)l * a->cache_layer_stride_bytes, a->attn, B, H, hd, a->t_new, a->t_max,
                                     a->max_prompt_len, a->cos_t, a->sin_t, a->pos, dt, s));                      // RoPE of q, k inside
        D3D_TRY(d3d_gemm_nt(a->attn, a->o_w[l], x, nullptr, x, B, Hd, Hd, Hd, Hd, Hd, dt, 4, s));                 // + residual, in place
        if (fuse) {
            D3D_TRY(d3d_gemm_nt_rmsnorm(x, a->n2[l], a->rms_eps, a->gate_up_w[l], a->act, B, 2 * I, Hd, Hd, Hd, I, dt, 6, s));            // + SwiGLU
        } else {
            D3D_TRY(d3d_norm(x, a->n2[l], nullptr, a->h, B, Hd, Hd, Hd, a->rms_eps, 1, dt, s));
            D3D_TRY(d3d_gemm_nt(a->h, a->gate_up_w[l], a->act, nullptr, nullptr, B, 2 * I, Hd, Hd, Hd, I, dt, 6, s));  // SwiGLU (interleaved rows)
        }
        D3D_TRY(d3d_gemm_nt(a->act, a->down_w[l], x, nullptr, x, B, Hd, I, I, I, Hd, dt, 4, s));                   // + residual, in place
    }
    if (fuse && a->vocab % 32 == 0) {
        D3D_TRY(d3d_gemm_nt_rmsnorm(x, a->norm_w, a->rms_eps, a->lm_head_w, a->logits, B, a->vocab, Hd, Hd, Hd, a->vocab, dt, 0, s));
    } else {
        D3D_TRY(d3d_norm(x, a->norm_w, nullptr, a->h, B, Hd, Hd, Hd, a->rms_eps, 1, dt, s));
        D3D_TRY(d3d_gemm_nt(a->h, a->lm_head_w, a->logits, nullptr, nullptr, B, a->vocab, Hd, Hd, Hd, a->vocab, dt, 0, s));
    }
#undef D3D_TRY
    return D3D_OK;
}
