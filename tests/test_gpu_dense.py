"""GPU: hand-written dense kernels (GEMM + fused epilogues, LayerNorm/RMSNorm, RoPE, SwiGLU, bicubic front-end)
against plain PyTorch float32 references of the same op, evaluated WITH THE REFERENCE'S 16-BIT STORES: a fused kernel must
return what the module sequence it replaces returns in bf16 / fp16 (include/dynam3d_hip.h "ROUNDING POINTS"), e.g.
bias+residual = R(R(acc + bias) + residual).  `epi_ref` below spells every epilogue out; with the stores in the same places
what is left is float32 summation order flipping an occasional store, so the tolerances are a third of a one-rounding bound
(bf16 1e-3, fp16 2e-4 relative L2; unit round-offs are 3.9e-3 / 4.9e-4) -- a kernel that stored once at the end would fail."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hd():
    from dynam3d_amd.hip_dense import HipDense
    return HipDense()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def epi_ref(name, y32, b, r, dt):
    """The module sequence each fused epilogue replaces, float32 arithmetic with a store in `dt` after every module."""
    R = lambda t: t.to(dt).float()
    N = y32.shape[1]
    bf = None if b is None else b.float()
    rf = None if r is None else r.float()
    if name == "none":
        return R(y32)
    if name == "bias":
        return R(y32 + bf)
    if name == "qgelu":                                   # x * sigmoid(1.702 * x) on 16-bit tensors (clip/model.py:162-164)
        y = R(y32 + bf)
        return R(y * R(torch.sigmoid(R(1.702 * y))))
    if name == "gelu":
        return R(F.gelu(R(y32 + bf)))
    if name == "res":
        return R(R(y32) + rf)
    if name == "bias_res":
        return R(R(y32 + bf) + rf)
    if name == "swiglu":                                  # HF Phi3MLP: up * silu(gate)
        return R(R(y32[:, N // 2:]) * R(F.silu(R(y32[:, :N // 2]))))
    raise KeyError(name)


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-3), (torch.float16, 2e-4)])
def test_gemm_epilogues_asymmetric(hd, dt, tol):
    from dynam3d_amd.hip_dense import interleave_gate_up
    torch.manual_seed(1)
    for M, N, K in ((300, 256, 192), (1000, 384, 1024), (77, 128, 64), (513, 1024, 3072)):
        x = (torch.randn(M, K, device="cuda") * 0.7).to(dt)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        w[3] *= 4.0                                   # asymmetric operands: transposes cannot cancel
        x[:, 5] += 1.0
        b = (torch.randn(N, device="cuda") * 0.3).to(dt)
        r = torch.randn(M, N, device="cuda").to(dt)
        y32 = x.float() @ w.float().t()
        cases = {
            "none": (hd.linear(x, w, None, None), epi_ref("none", y32, None, None, dt)),
            "bias": (hd.linear(x, w, b, None), epi_ref("bias", y32, b, None, dt)),
            "qgelu": (hd.linear(x, w, b, "quick_gelu"), epi_ref("qgelu", y32, b, None, dt)),
            "gelu": (hd.linear(x, w, b, "gelu"), epi_ref("gelu", y32, b, None, dt)),
            "res": (hd.linear(x, w, None, None, r), epi_ref("res", y32, None, r, dt)),
            "bias_res": (hd.linear(x, w, b, None, r), epi_ref("bias_res", y32, b, r, dt)),
            "swiglu": (hd.linear_swiglu(x, interleave_gate_up(w)), epi_ref("swiglu", y32, None, None, dt)),
        }
        for name, (got, exp) in cases.items():
            assert got.dtype == dt and got.shape == exp.shape
            assert rel(got.float(), exp) < tol, (name, M, N, K, rel(got.float(), exp))
        # strided A (a column slice of a wider buffer)
        big = (torch.randn(M, K + 64, device="cuda")).to(dt)
        assert rel(hd.linear(big[:, :K], w, None, None).float(), epi_ref("none", big[:, :K].float() @ w.float().t(), None, None, dt)) < tol


@pytest.mark.parametrize("tile", [130, 132, 164, 256, 257, 259, 260])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-3), (torch.float16, 2e-4)])
def test_gemm_forced_tile_kernels(hd, dt, tol, tile):
    """Every kernel variant forced explicitly -- 128x128 with a 2- / 4-deep LDS ring (130 / 132), 256x256x64 staggered in
    both step sizes (256 = K-half steps, 257 = whole-K-tile steps): ragged M, fewer K tiles than ring slots, many K tiles,
    every fused epilogue."""
    from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
    torch.manual_seed(2)
    try:
        HipDense.TILE = tile
        for M, N, K in ((300, 256, 64), (512, 512, 128), (1000, 768, 1024), (2304, 3072, 192)):
            x = (torch.randn(M, K, device="cuda") * 0.7).to(dt)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
            w[3] *= 4.0
            x[:, 5] += 1.0
            b = (torch.randn(N, device="cuda") * 0.3).to(dt)
            r = torch.randn(M, N, device="cuda").to(dt)
            y32 = x.float() @ w.float().t()
            assert rel(hd.linear(x, w, None, None).float(), epi_ref("none", y32, None, None, dt)) < tol, (M, N, K, "none")
            assert rel(hd.linear(x, w, b, "quick_gelu").float(), epi_ref("qgelu", y32, b, None, dt)) < tol, (M, N, K, "qgelu")
            assert rel(hd.linear(x, w, b, None, r).float(), epi_ref("bias_res", y32, b, r, dt)) < tol, (M, N, K, "bias_res")
            assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), epi_ref("swiglu", y32, None, None, dt)) < tol, (M, N, K, "swiglu")
    finally:
        HipDense.TILE = 0


@pytest.mark.parametrize("tile", [258, 264])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-3), (torch.float16, 2e-4)])
def test_gemm_split_k_tail(hd, dt, tol, tile):
    """Tile codes 258 (staggered K loop) / 264 (interleaved K loop): whole rounds of 256x256 tiles data-parallel + the remainder tiles cut into K-slices whose fp32 partials a
    second launch sums in slice order.  Shapes: tail only (4, 44, 100 tiles -> 8, 5, 2 slices), rounds + tail, slices > K tiles,
    no tail at all; repeated launches reuse the workspace; results are bit-identical from launch to launch."""
    from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
    torch.manual_seed(3)
    try:
        HipDense.TILE = tile
        for M, N, K in ((512, 512, 1024), (2816, 1024, 2048), (6400, 1024, 512), (6400, 3072, 3072), (512, 512, 128), (4096, 4096, 256)):
            x = (torch.randn(M, K, device="cuda") * 0.7).to(dt)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
            w[3] *= 4.0
            x[:, 5] += 1.0
            b = (torch.randn(N, device="cuda") * 0.3).to(dt)
            r = torch.randn(M, N, device="cuda").to(dt)
            y32 = x.float() @ w.float().t()
            for rep in range(2):
                assert rel(hd.linear(x, w, None, None).float(), epi_ref("none", y32, None, None, dt)) < tol, (M, N, K, "none")
                assert rel(hd.linear(x, w, b, None, r).float(), epi_ref("bias_res", y32, b, r, dt)) < tol, (M, N, K, "bias_res")
                assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), epi_ref("swiglu", y32, None, None, dt)) < tol, (M, N, K, "swiglu")
            assert torch.equal(hd.linear(x, w, None, None, r), hd.linear(x, w, None, None, r))
    finally:
        HipDense.TILE = 0


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-3), (torch.float16, 2e-4)])
def test_gemm_skinny_decode_rows(hd, dt, tol):
    """M <= 16 rows (KV-cache decode) take the weight-streaming kernel: one wave per 32 output columns and K slice, slices reduced
    in slice order by the last arriver.  1 / 8 / 16 rows, the Phi-3 decode shapes (scaled), a vocabulary-sized N, K not a
    multiple of the slice count, all supported epilogues; repeated launches reuse the arrival counters; bit-identical reruns."""
    from dynam3d_amd.hip_dense import interleave_gate_up
    torch.manual_seed(4)
    for M, N, K in ((8, 9216, 3072), (8, 3072, 8192), (1, 3072, 3072), (16, 16384, 1024), (8, 32064, 768), (5, 96, 416)):
        x = (torch.randn(M, K, device="cuda") * 0.7).to(dt)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        w[3] *= 4.0
        x[:, 5] += 1.0
        b = (torch.randn(N, device="cuda") * 0.3).to(dt)
        r = torch.randn(M, N, device="cuda").to(dt)
        y32 = x.float() @ w.float().t()
        for rep in range(2):
            assert rel(hd.linear(x, w, None, None).float(), epi_ref("none", y32, None, None, dt)) < tol, (M, N, K, "none")
            assert rel(hd.linear(x, w, b, None).float(), epi_ref("bias", y32, b, None, dt)) < tol, (M, N, K, "bias")
            assert rel(hd.linear(x, w, None, None, r).float(), epi_ref("res", y32, None, r, dt)) < tol, (M, N, K, "res")
            if N % 32 == 0 and (N // 2) % 16 == 0:
                assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), epi_ref("swiglu", y32, None, None, dt)) < tol, (M, N, K, "swiglu")
        assert torch.equal(hd.linear(x, w, None, None, r), hd.linear(x, w, None, None, r))


def test_gemm_output_row_stride_not_16_byte_aligned(hd):
    """The 256-tile kernels and the transposed epilogue of the 128-tile kernel store 16 bytes per lane: an output whose row stride is
    a multiple of 4 but not of 8 elements (rows not 16-byte aligned) must take the 8-byte store path and still be right -- d3d_gemm_nt
    routes it to the 128 x 128 kernel with direct stores; forcing a 256-tile kernel on it is refused."""
    import ctypes as C
    from dynam3d_amd import _lib
    lib = _lib.load()
    torch.manual_seed(12)
    M, N, K = 2304, 3072, 512
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    ldc = N + 4
    out = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
    res = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
    res[:, :N] = r
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.d3d_gemm_nt(p(x), p(w), p(out), None, p(res), M, N, K, K, K, ldc, 0, 4, st)
    assert rc == 0
    ref = epi_ref("res", x.float() @ w.float().t(), None, r, torch.bfloat16)
    assert rel(out[:, :N].float(), ref) < 1e-3
    assert float(out[:, N:].abs().max()) == 0.0                                   # nothing written past the row
    rc = lib.d3d_gemm_nt_tile(p(x), p(w), p(out), None, p(res), M, N, K, K, K, ldc, 0, 4, 260, st)
    assert rc != 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_skinny_fused_rmsnorm(hd, dt):
    """d3d_gemm_nt_rmsnorm (decode: RMSNorm applied to the raw residual stream inside the weight-streaming GEMM) is BIT-identical to
    d3d_norm followed by d3d_gemm_nt: plain and SwiGLU epilogues, 1 / 8 / 16 rows, the Phi-3 decode shapes."""
    from dynam3d_amd.hip_dense import interleave_gate_up
    torch.manual_seed(9)
    for M, N, K in ((8, 9216, 3072), (1, 3072, 3072), (16, 16384, 1024), (8, 32064, 3072), (5, 96, 512)):
        x = (torch.randn(M, K, device="cuda") * 1.7 + 0.2).to(dt)
        g = (torch.randn(K, device="cuda") * 0.2 + 1).float()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        h = hd.rms_norm(x, g, 1e-5)
        assert torch.equal(hd.linear_rmsnorm(x, g, 1e-5, w), hd.linear(h, w, None, None)), (M, N, K)
        if (N // 2) % 16 == 0:
            wi = interleave_gate_up(w)
            assert torch.equal(hd.linear_rmsnorm(x, g, 1e-5, wi, swiglu=True), hd.linear_swiglu(h, wi)), (M, N, K, "swiglu")


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 1e-3), (torch.float16, 2e-4)])
def test_norms_rope_swiglu(hd, dt, tol):
    torch.manual_seed(2)
    R = lambda t: t.to(dt).float()
    for D in (768, 1024, 3072, 4096, 128):
        x = (torch.randn(1001, D, device="cuda") * 2 + 0.3).to(dt)
        w, b = torch.randn(D, device="cuda") * 0.2 + 1, torch.randn(D, device="cuda") * 0.1
        assert rel(hd.layer_norm(x, w, b, 1e-5).float(), R(F.layer_norm(x.float(), (D,), w, b, 1e-5))) < tol
        xf = x.float()                                        # HF Phi3RMSNorm: weight * x_hat.to(dtype)
        assert rel(hd.rms_norm(x, w, 1e-5).float(), R(R(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)) * w)) < tol
    B, S, H, hdm = 3, 50, 6, 96
    qkv = torch.randn(B, S, 3 * H, hdm, device="cuda").to(dt)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hdm, 2, device="cuda").float() / hdm))
    ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    ref = qkv.clone().float()
    x1, x2 = ref[:, :, :2 * H, :hdm // 2].clone(), ref[:, :, :2 * H, hdm // 2:].clone()
    c, s = cos[None, :, None], sin[None, :, None]
    ref[:, :, :2 * H] = torch.cat([R(R(x1 * c) - R(x2 * s)), R(R(x2 * c) + R(x1 * s))], -1)      # HF apply_rotary_pos_emb on 16-bit tensors
    got = qkv.clone()
    hd.rope_inplace(got.view(B * S, -1), cos, sin, S, 2 * H, hdm)
    assert rel(got.float(), ref) < tol and torch.equal(got[:, :, 2 * H:], qkv[:, :, 2 * H:])
    gu = torch.randn(333, 2 * 8192, device="cuda").to(dt)
    assert rel(hd.swiglu(gu).float(), R(gu[:, 8192:].float() * R(F.silu(gu[:, :8192].float())))) < tol


def test_resize_normalize_matches_torch_bicubic(hd):
    from dynam3d_amd import dense_ops as D
    from dynam3d_amd.towers import CLIP_MEAN, CLIP_STD
    rng = np.random.default_rng(4)
    for hw in (224, 336, 97):
        rgb = torch.from_numpy(rng.integers(0, 256, (3, hw, hw, 3), dtype=np.uint8)).cuda()
        got = hd.resize_normalize(rgb, 336, CLIP_MEAN, CLIP_STD)
        exp = D.resize_normalize(rgb.cpu(), 336, CLIP_MEAN, CLIP_STD).cuda()      # torch CPU bicubic (the oracle's op)
        # identical except where the float32 interpolant sits within rounding noise of a .5 boundary: <= 1 LSB (1/255/std)
        diff = (got - exp).abs()
        assert float(diff.max()) <= 1.0 / 255 / min(CLIP_STD) + 1e-5
        assert float((diff > 1e-5).float().mean()) < 2e-3


def test_set_attention_varlen(hd):
    """d3d_set_attention (fp32 varlen self-attention inside packed sets) vs a per-set torch reference."""
    torch.manual_seed(3)
    H, lens = 12, [1, 37, 2, 64, 65, 300, 5]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    T = int(off[-1])
    qkv = torch.randn(T, 3 * H * 64, device="cuda")
    set_off = torch.from_numpy(off).cuda()
    out = hd.set_attention(qkv, set_off, len(lens), H, max(lens))
    cls = hd.set_attention(qkv, set_off, len(lens), H, max(lens), q_rows=1)
    for g, L in enumerate(lens):
        blk = qkv[off[g]:off[g + 1]].view(L, 3, H, 64)
        q, k, v = (blk[:, j].transpose(0, 1) for j in range(3))
        ref = F.scaled_dot_product_attention(q[None], k[None], v[None])[0].transpose(0, 1).reshape(L, H * 64)
        assert torch.allclose(out[off[g]:off[g + 1]], ref, atol=2e-5, rtol=1e-4)
        assert torch.allclose(cls[off[g]], ref[0], atol=2e-5, rtol=1e-4)
        if L > 1:
            assert float(cls[off[g] + 1:off[g + 1]].abs().sum()) == 0.0


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 4e-3), (torch.float16, 6e-4)])
def test_flash_attention_vs_fp32_reference(hd, dt, tol):
    """d3d_flash_attention_v3 (head_dim 64/96, causal/full, ragged S, masked tail) vs float32 SDPA on the same 16-bit inputs."""
    torch.manual_seed(4)
    for (B, H, S, d, causal) in [(2, 3, 577, 64, False), (2, 4, 900, 96, True), (1, 2, 130, 96, True), (3, 2, 64, 64, True), (1, 1, 1, 96, True), (2, 2, 333, 96, False)]:
        qkv = (torch.randn(B, S, 3 * H, d, device="cuda") * 1.5).to(dt)
        qkv[:, :, H:2 * H] += 0.5                                  # asymmetric q/k
        q, k, v = (qkv[:, :, i * H:(i + 1) * H].transpose(1, 2).float() for i in range(3))
        ref = F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2)
        got = hd.attention_qkv(qkv, H, causal).float()
        assert got.shape == ref.shape
        assert rel(got, ref) < tol, (B, H, S, d, causal, rel(got, ref))
    # a spike that forces the deferred-max rescale path (one key dominating late in the sequence)
    B, H, S, d = 1, 1, 256, 96
    qkv = (torch.randn(B, S, 3 * H, d, device="cuda") * 0.3).to(dt)
    qkv[0, 200, 1] = qkv[0, 255, 0] * 40                            # k[200] aligned with q[255] -> huge score at a late tile
    q, k, v = (qkv[:, :, i * H:(i + 1) * H].transpose(1, 2).float() for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2)
    assert rel(hd.attention_qkv(qkv, H, True).float(), ref) < tol


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
def test_flash_attention_packed_ragged(hd, dt, tol, causal):
    """Packed variable-length batch (cu_seqlens): sequences of 1 token, below / at / just above a 128-row query block and a
    64-key tile, odd and even block counts (the causal kernel pairs the longest block of a sequence with its shortest), padding
    rows after the last sequence.  Reference: fp32 softmax attention per sequence.  Tolerance = 16-bit output rounding + 16-bit P."""
    torch.manual_seed(5)
    H, d = 4, 96
    lens = [1, 63, 128, 129, 200, 385, 640, 705]
    T = sum(lens)
    Tp = (T + 255) // 256 * 256
    qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.8).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    out = hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens)).float()
    o = 0
    for n in lens:
        x = qkv[o:o + n].float()
        q, k, v = x[:, :H].transpose(0, 1), x[:, H:2 * H].transpose(0, 1), x[:, 2 * H:].transpose(0, 1)      # (H,n,d)
        ref = F.scaled_dot_product_attention(q[None], k[None], v[None], is_causal=causal)[0].transpose(0, 1)
        assert rel(out[o:o + n], ref) < tol, (n, rel(out[o:o + n], ref))
        o += n
    assert float(out[T:].abs().max()) == 0.0                     # padding rows untouched


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_flash_attention_bitwise_repeatable(hd, dt):
    """The same launch twenty times gives the same bits.  (Round 3: a three-input maximum written as inline asm read the score
    accumulators without the wait states a VALU needs after the MFMA that wrote them -- the hazard recogniser does not look inside
    asm -- and the last bits of every query block after the first changed from launch to launch while every tolerance test passed.)"""
    torch.manual_seed(9)
    for H, d, lens in ((32, 96, [37, 211, 129, 64, 5, 90, 300, 17]), (8, 96, [828, 826, 1072]), (16, 64, [577, 577])):
        T = sum(lens)
        qkv = (torch.randn((T + 255) // 256 * 256, 3 * H, d, device="cuda") * 0.5).to(dt)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        for causal in (True, False):
            first = hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens)).clone()
            for _ in range(20):
                assert torch.equal(hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens)), first), (H, d, lens, causal)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_flash_attention_fused_query_rope_equals_separate_rope(hd, dt):
    """d3d_flash_attention_v3_rope_q (un-rotated q in the buffer, rotated inside the kernel; k rotated in place) == d3d_rope_inplace over
    q and k followed by d3d_flash_attention_v3, bit for bit: packed ragged prompts, head_dim 96 and 64, with and without the window."""
    torch.manual_seed(12)
    for H, d, lens, window in ((8, 96, [1, 63, 129, 200, 385, 705], 0), (4, 64, [300, 77], 0), (2, 96, [700, 40], 257)):
        T = sum(lens)
        Tp = (T + 255) // 256 * 256
        qkv = (torch.randn(Tp, 3 * H * d, device="cuda") * 0.8).to(dt)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        pos = torch.zeros(Tp, dtype=torch.int32, device="cuda")
        for b, n in enumerate(lens):
            pos[int(cu[b]):int(cu[b]) + n] = torch.arange(n, dtype=torch.int32, device="cuda")
        S = max(lens)
        inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32, device="cuda") / d))
        ang = torch.arange(S + 5, dtype=torch.float32, device="cuda")[:, None] * inv[None]
        cos, sin = ang.cos().to(dt).float().contiguous(), ang.sin().to(dt).float().contiguous()
        a = qkv.clone()
        hd.rope_inplace(a, cos, sin, 1, 2 * H, d, pos)                                  # q and k heads
        ref = hd.attention_packed(a.view(Tp, 3 * H, d), H, True, cu, len(lens), S, n_valid=T, window=window)
        b_ = qkv.clone()
        hd.rope_inplace(b_[:, H * d:], cos, sin, 1, H, d, pos)                           # k heads only
        assert torch.equal(b_[:, H * d:], a[:, H * d:]) and torch.equal(b_[:, :H * d], qkv[:, :H * d])
        out = hd.attention_packed(b_.view(Tp, 3 * H, d), H, True, cu, len(lens), S, n_valid=T, window=window, rope_q=(cos, sin))
        assert torch.equal(out, ref), (H, d, lens, window, float((out.float() - ref.float()).abs().max()))
        # the SCHEDULED launch (d3d_flash_attention_v3_sched: one query block per workgroup, heaviest first) computes every block exactly
        # as the paired launch does: same bits, with and without the fused query rotation
        tab = torch.from_numpy(hd.attention_schedule(lens, H)).cuda()
        assert tab.numel() == sum((n + 127) // 128 for n in lens) * H
        ref3 = ref
        o1 = hd.attention_packed(b_.view(Tp, 3 * H, d), H, True, cu, len(lens), S, n_valid=T, window=window, rope_q=(cos, sin), sched=tab)
        o2 = hd.attention_packed(a.view(Tp, 3 * H, d), H, True, cu, len(lens), S, n_valid=T, window=window, sched=tab)
        assert torch.equal(o1, ref3) and torch.equal(o2, ref3), (H, d, lens, window)


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
def test_flash_attention_v4_software_pipelined_kernel(hd, dt, tol, monkeypatch):
    """The round-6 experiment kernel (csrc/attn4_kernels.hip; D3D_ATTN_V4=1: MFMAs of key blocks b-1 / b+1 interleaved with the softmax of
    block b inside the wave) against float32 attention and against the product kernel: ragged packed causal prompts with and without the
    fused query RoPE, dense ViT shape; every edge of its slot loop (1-block sequences, diagonal blocks, partial last tiles)."""
    torch.manual_seed(21)
    for H, d, lens, causal in ((8, 96, [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 385, 705, 830], True), (4, 64, [300, 77, 577], False),
                               (4, 64, [5, 640], True), (2, 96, [257, 96], False)):
        T = sum(lens)
        Tp = (T + 255) // 256 * 256
        qkv = (torch.randn(Tp, 3 * H, d, device="cuda") * 0.7).to(dt)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        v3 = hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens), n_valid=T)
        monkeypatch.setenv("D3D_ATTN_V4", "1")
        v4 = hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens), n_valid=T)
        for _ in range(3):
            assert torch.equal(hd.attention_packed(qkv, H, causal, cu, len(lens), max(lens), n_valid=T), v4)       # repeatable
        monkeypatch.delenv("D3D_ATTN_V4")
        o = 0
        for n in lens:
            x = qkv[o:o + n].float()
            q, k, v = (x[:, i * H:(i + 1) * H].transpose(0, 1) for i in range(3))
            ref = F.scaled_dot_product_attention(q[None], k[None], v[None], is_causal=causal)[0].transpose(0, 1)
            assert rel(v4[o:o + n].float(), ref) < tol, (H, d, n, causal)
            assert rel(v4[o:o + n].float(), v3[o:o + n].float()) < tol
            o += n
        assert float(v4[T:].abs().max()) == 0.0
    qv = (torch.randn(2, 577, 48, 64, device="cuda") * 0.5).to(dt)
    monkeypatch.setenv("D3D_ATTN_V4", "1")
    got = hd.attention_qkv(qv, 16, False).float()
    monkeypatch.delenv("D3D_ATTN_V4")
    q, k, v = (qv[:, :, i * 16:(i + 1) * 16].float().transpose(1, 2) for i in range(3))
    assert rel(got, F.scaled_dot_product_attention(q, k, v).transpose(1, 2)) < tol


def test_flash_attention_sliding_window(hd):
    """The v2 kernel's sliding window (HF Phi-3-mini-4k: a query attends to its last `window` keys, itself included): packed ragged
    prompts longer than the window and a dense batch, windows that cut inside a tile / at a tile edge / before the first query block,
    against an explicitly masked float32 softmax."""
    torch.manual_seed(8)
    H, d = 2, 96
    for dt, tol in ((torch.bfloat16, 6e-3), (torch.float16, 1e-3)):
        for window in (1, 50, 64, 257, 300):
            lens = [40, 333, 700]
            T = sum(lens)
            qkv = (torch.randn((T + 255) // 256 * 256, 3 * H, d, device="cuda") * 0.8).to(dt)
            cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
            out = hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), window=window).float()
            o = 0
            for n in lens:
                x = qkv[o:o + n].float()
                q, k, v = x[:, :H].transpose(0, 1), x[:, H:2 * H].transpose(0, 1), x[:, 2 * H:].transpose(0, 1)      # (H,n,d)
                i = torch.arange(n, device="cuda")
                ok = (i[None, :] <= i[:, None]) & (i[None, :] > i[:, None] - window)
                ref = F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=ok[None, None])[0].transpose(0, 1)
                assert rel(out[o:o + n], ref) < tol, (window, n, rel(out[o:o + n], ref))
                o += n
        qkv = (torch.randn(2, 400, 3 * H, 64, device="cuda")).to(dt)
        q, k, v = (qkv[:, :, i * H:(i + 1) * H].transpose(1, 2).float() for i in range(3))
        i = torch.arange(400, device="cuda")
        ok = (i[None, :] <= i[:, None]) & (i[None, :] > i[:, None] - 130)
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=ok[None, None]).transpose(1, 2)
        assert rel(hd.attention_qkv(qkv, H, True, window=130).float(), ref) < tol


@pytest.mark.parametrize("split", [None, "2", "attn1"])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
def test_decode_attention_kv_cache(hd, dt, tol, split, monkeypatch):
    """(default: the one-pass kernel k_decode_attn2; "attn1" = D3D_DECODE_ATTN=1: the three-sweep kernel it replaced, kept as a knob;
    split = D3D_DECODE_SPLIT: the keys of a (sequence, head) cut into ranges whose partial softmax results the last workgroup
    to finish merges -- off by default, kept as a tested knob.)
    d3d_decode_attention: one query per sequence over [prompt keys read in place from a packed QKV buffer | side cache | the
    current token], ragged prompts, several decode steps; vs fp32 softmax attention over the same (16-bit) keys and values.  The
    fused-RoPE form (un-rotated q, k + cos/sin/pos) must give what rope_inplace followed by the plain form gives."""
    if split == "attn1":
        monkeypatch.setenv("D3D_DECODE_ATTN", "1")
    elif split is not None:
        monkeypatch.setenv("D3D_DECODE_SPLIT", split)
    torch.manual_seed(6)
    H, d, Tmax = 4, 96, 5
    lens = [1, 63, 300, 129]
    B, T = len(lens), sum(lens)
    prompt = (torch.randn(T + 7, 3 * H, d, device="cuda") * 0.8).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    inv = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32, device="cuda") / d))
    ang = torch.arange(max(lens) + Tmax + 1, dtype=torch.float32, device="cuda")[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
    kn_a, vn_a = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    kn_b, vn_b = (torch.zeros(B, Tmax, H, d, dtype=dt, device="cuda") for _ in range(2))
    new_k, new_v = [], []
    for t in range(Tmax):
        raw = (torch.randn(B, 3 * H, d, device="cuda") * 0.8).to(dt)
        pos = (lens_d + t).contiguous()
        rot = raw.clone().view(B, 3 * H * d)
        hd.rope_inplace(rot, cos, sin, 1, 2 * H, d, pos)
        rot = rot.view(B, 3 * H, d)
        out_a = hd.decode_attention(rot.view(B, -1), prompt.view(T + 7, -1), cu, kn_a, vn_a, H, t, max(lens)).float().view(B, H, d)
        out_b = hd.decode_attention(raw.view(B, -1), prompt.view(T + 7, -1), cu, kn_b, vn_b, H, t, max(lens), rope=(cos, sin, pos)).float().view(B, H, d)
        new_k.append(rot[:, H:2 * H].float())
        new_v.append(rot[:, 2 * H:].float())
        for b, n in enumerate(lens):
            o = int(cu[b])
            k = torch.cat([prompt[o:o + n, H:2 * H].float()] + [x[b][None] for x in new_k], 0)            # (L,H,d)
            v = torch.cat([prompt[o:o + n, 2 * H:].float()] + [x[b][None] for x in new_v], 0)
            q = rot[b, :H].float()                                                                        # (H,d)
            p = torch.softmax(torch.einsum("hd,lhd->hl", q, k) / d ** 0.5, -1)
            ref = torch.einsum("hl,lhd->hd", p, v)
            assert rel(out_a[b], ref) < tol, (t, b, rel(out_a[b], ref))
        assert rel(out_b, out_a) < 1e-3 and torch.equal(vn_a[:, :t + 1], vn_b[:, :t + 1])
        # the rotated keys of the two forms are the same arithmetic in two kernels; the compiler may fuse `convert(x * c)` into one
        # mixed-precision instruction in one of them (a single rounding of the exact product instead of float32-then-16-bit), which
        # moves a value sitting on a 16-bit rounding tie by one unit in the last place: at most a few elements in ten thousand
        ka, kb = kn_a[:, :t + 1].float(), kn_b[:, :t + 1].float()
        ne = ka != kb
        assert float(ne.float().mean()) < 1e-3 and float(((ka - kb).abs() / ka.abs().clamp_min(1e-3))[ne].max() if ne.any() else 0.0) < (1e-2 if dt == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dt,tol", [(torch.float16, 1e-3), (torch.bfloat16, 6e-3)])
def test_vit_shape_attention_five_wave_workgroups(hd, dt, tol, monkeypatch):
    """The ViT towers' shape -- 8 images x 16 heads x 577 rows, head_dim 64, non-causal -- can be launched with 5-wave workgroups (160 query
    rows: 512 workgroups = one round of the chip instead of 640 = two; D3D_ATTN_NW5=1 -- measured slower in the step, so a knob).  Same
    arithmetic per query: bit-identical to the default 4-wave launch, and both against float32 softmax attention."""
    torch.manual_seed(21)
    B, S, H, d = 8, 577, 16, 64
    qkv = (torch.randn(B, S, 3 * H, d, device="cuda") * 0.7).to(dt)
    out4 = hd.attention_qkv(qkv, H, False)
    monkeypatch.setenv("D3D_ATTN_NW5", "1")
    out5 = hd.attention_qkv(qkv, H, False)
    monkeypatch.delenv("D3D_ATTN_NW5")
    assert torch.equal(out5, out4)
    q, k, v = (qkv[:, :, i * H:(i + 1) * H].transpose(1, 2).float() for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2)
    assert rel(out5.float(), ref) < tol
    # a shape where 160-row blocks do not save a round keeps the 4-wave kernel (nothing to compare, just runs): 2 images
    assert torch.isfinite(hd.attention_qkv(qkv[:2].contiguous(), H, False).float()).all()
