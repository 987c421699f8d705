// mlp_forward.cpp -- tinycudann `Network(otype="CutlassMLP")` forward as ONE C call (SURVEY.md 8 b2/b3 `d3d_mlp768_forward`):
// bias-free layers y = act(x W^T), fp16 weights and activations, fp32 accumulation, fp16 store per layer: one d3d_gemm_nt launch per
// layer with the activation in the epilogue (7 = LeakyReLU(0.01), 0 = none), or -- D3D_MLP_FUSED=1, networks of <= 4 layers and widths
// <= 896 -- ONE launch with the activations resident in LDS (csrc/mlp_kernels.hip, d3d_mlp_fused: bit-identical results, measured
// 2.5-3x slower on this chip, hence opt-in).  Reference call sites: PRE-FF:221-243 (construction), 484 / 488
// (nerf_encoder 768->768->768->769, nerf_decoder 768->768->768->768).
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dynam3d_hip.h"

extern "C" void d3d_set_error_(const char* msg);

extern "C" int32_t d3d_mlp768_forward(const void* x, int64_t n_rows, int32_t n_in, const void* const* weights, int32_t n_hidden,
                                      int32_t n_neurons, int32_t n_out_padded, int32_t act, int32_t out_act, void* scratch_a,
                                      void* scratch_b, void* y, void* stream) {
    if (n_rows <= 0) return D3D_OK;
    if (n_hidden < 1 || (act != 0 && act != 1) || (out_act != 0 && out_act != 1) || n_rows > INT32_MAX) {
        d3d_set_error_("d3d_mlp768_forward: n_hidden >= 1, activations 0 (none) / 1 (LeakyReLU 0.01)");
        return D3D_EINVAL;
    }
    if (n_in % 64 || n_neurons % 128 || n_out_padded % 128) {
        d3d_set_error_("d3d_mlp768_forward: n_in % 64, n_neurons % 128, n_out_padded % 128 (zero-pad the last layer's rows)");
        return D3D_EINVAL;
    }
    static const bool fused = [] { const char* e = getenv("D3D_MLP_FUSED"); return e && e[0] == '1'; }();      // measured slower: see mlp_kernels.hip
    if (fused && n_hidden + 1 <= 4 && n_in <= 896 && n_neurons <= 896 && n_out_padded <= 896 && n_in % 32 == 0 && n_neurons % 32 == 0) {
        int32_t widths[5], modes[4];
        const void* aux[4] = {nullptr, nullptr, nullptr, nullptr};
        void* outs[4] = {nullptr, nullptr, nullptr, nullptr};
        int64_t ld_aux[4] = {0, 0, 0, 0}, ld_outs[4] = {0, 0, 0, 0};
        widths[0] = n_in;
        for (int32_t l = 0; l <= n_hidden; ++l) {
            widths[l + 1] = l == n_hidden ? n_out_padded : n_neurons;
            modes[l] = (l == n_hidden ? out_act : act) ? 1 : 0;
        }
        outs[n_hidden] = y;
        ld_outs[n_hidden] = n_out_padded;
        return d3d_mlp_fused(x, n_in, n_rows, n_hidden + 1, widths, weights, modes, aux, ld_aux, outs, ld_outs, /*fp16*/ 1, stream);
    }
    const int32_t M = (int32_t)n_rows;
    const void* in = x;
    int32_t k = n_in;
    void* bufs[2] = {scratch_a, scratch_b};
    for (int32_t l = 0; l <= n_hidden; ++l) {
        const bool last = l == n_hidden;
        const int32_t n = last ? n_out_padded : n_neurons;
        void* out = last ? y : bufs[l & 1];
        const int32_t rc = d3d_gemm_nt(in, weights[l], out, nullptr, nullptr, M, n, k, k, k, n, /*fp16*/ 1,
                                       (last ? out_act : act) ? 7 : 0, stream);
        if (rc != D3D_OK) return rc;
        in = out;
        k = n;
    }
    return D3D_OK;
}
