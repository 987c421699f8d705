"""GPU: hand-written dense kernels (GEMM + fused epilogues, LayerNorm/RMSNorm, RoPE, SwiGLU, bicubic front-end)
against plain PyTorch float32 references of the same op.  Tolerances: outputs are bf16/fp16, i.e. one final
rounding (unit roundoff 3.9e-3 / 4.9e-4) on top of fp32 accumulation -> relative L2 <= 3e-3 (bf16) / 6e-4 (fp16)."""
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


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-3), (torch.float16, 6e-4)])
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
            "none": (hd.linear(x, w, None, None), y32),
            "bias": (hd.linear(x, w, b, None), y32 + b.float()),
            "qgelu": (hd.linear(x, w, b, "quick_gelu"), (lambda t: t * torch.sigmoid(1.702 * t))(y32 + b.float())),
            "gelu": (hd.linear(x, w, b, "gelu"), F.gelu(y32 + b.float())),
            "res": (hd.linear(x, w, None, None, r), y32 + r.float()),
            "bias_res": (hd.linear(x, w, b, None, r), y32 + b.float() + r.float()),
            "swiglu": (hd.linear_swiglu(x, interleave_gate_up(w)), y32[:, N // 2:] * F.silu(y32[:, :N // 2])),
        }
        for name, (got, exp) in cases.items():
            assert got.dtype == dt and got.shape == exp.shape
            assert rel(got.float(), exp) < tol, (name, M, N, K, rel(got.float(), exp))
        # strided A (a column slice of a wider buffer)
        big = (torch.randn(M, K + 64, device="cuda")).to(dt)
        assert rel(hd.linear(big[:, :K], w, None, None).float(), big[:, :K].float() @ w.float().t()) < tol


@pytest.mark.parametrize("tile", [130, 132, 256, 257])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-3), (torch.float16, 6e-4)])
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
            assert rel(hd.linear(x, w, None, None).float(), y32) < tol, (M, N, K, "none")
            assert rel(hd.linear(x, w, b, "quick_gelu").float(), (lambda t: t * torch.sigmoid(1.702 * t))(y32 + b.float())) < tol, (M, N, K, "qgelu")
            assert rel(hd.linear(x, w, b, None, r).float(), y32 + b.float() + r.float()) < tol, (M, N, K, "bias_res")
            assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), y32[:, N // 2:] * F.silu(y32[:, :N // 2])) < tol, (M, N, K, "swiglu")
    finally:
        HipDense.TILE = 0


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-3), (torch.float16, 6e-4)])
def test_gemm_split_k_tail(hd, dt, tol):
    """Tile code 258: whole rounds of 256x256 tiles data-parallel + the remainder tiles cut into K-slices whose last arriver
    reduces them in slice order.  Shapes: tail only (4, 44, 100 tiles -> 8, 5, 2 slices), rounds + tail, slices > K tiles,
    no tail at all; repeated launches reuse the counters; results are bit-identical from launch to launch."""
    from dynam3d_amd.hip_dense import HipDense, interleave_gate_up
    torch.manual_seed(3)
    try:
        HipDense.TILE = 258
        for M, N, K in ((512, 512, 1024), (2816, 1024, 2048), (6400, 1024, 512), (6400, 3072, 3072), (512, 512, 128), (4096, 4096, 256)):
            x = (torch.randn(M, K, device="cuda") * 0.7).to(dt)
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
            w[3] *= 4.0
            x[:, 5] += 1.0
            b = (torch.randn(N, device="cuda") * 0.3).to(dt)
            r = torch.randn(M, N, device="cuda").to(dt)
            y32 = x.float() @ w.float().t()
            for rep in range(2):
                assert rel(hd.linear(x, w, None, None).float(), y32) < tol, (M, N, K, "none")
                assert rel(hd.linear(x, w, b, None, r).float(), y32 + b.float() + r.float()) < tol, (M, N, K, "bias_res")
                assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), y32[:, N // 2:] * F.silu(y32[:, :N // 2])) < tol, (M, N, K, "swiglu")
            assert torch.equal(hd.linear(x, w, None, None, r), hd.linear(x, w, None, None, r))
    finally:
        HipDense.TILE = 0


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-3), (torch.float16, 6e-4)])
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
            assert rel(hd.linear(x, w, None, None).float(), y32) < tol, (M, N, K, "none")
            assert rel(hd.linear(x, w, b, None).float(), y32 + b.float()) < tol, (M, N, K, "bias")
            assert rel(hd.linear(x, w, None, None, r).float(), y32 + r.float()) < tol, (M, N, K, "res")
            if N % 32 == 0 and (N // 2) % 16 == 0:
                assert rel(hd.linear_swiglu(x, interleave_gate_up(w)).float(), y32[:, N // 2:] * F.silu(y32[:, :N // 2])) < tol, (M, N, K, "swiglu")
        assert torch.equal(hd.linear(x, w, None, None, r), hd.linear(x, w, None, None, r))


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 3e-3), (torch.float16, 6e-4)])
def test_norms_rope_swiglu(hd, dt, tol):
    torch.manual_seed(2)
    for D in (768, 1024, 3072, 4096, 128):
        x = (torch.randn(1001, D, device="cuda") * 2 + 0.3).to(dt)
        w, b = torch.randn(D, device="cuda") * 0.2 + 1, torch.randn(D, device="cuda") * 0.1
        assert rel(hd.layer_norm(x, w, b, 1e-5).float(), F.layer_norm(x.float(), (D,), w, b, 1e-5)) < tol
        xf = x.float()
        assert rel(hd.rms_norm(x, w, 1e-5).float(), xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5) * w) < tol
    B, S, H, hdm = 3, 50, 6, 96
    qkv = torch.randn(B, S, 3 * H, hdm, device="cuda").to(dt)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hdm, 2, device="cuda").float() / hdm))
    ang = torch.arange(S, device="cuda").float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    ref = qkv.clone().float()
    x1, x2 = ref[:, :, :2 * H, :hdm // 2].clone(), ref[:, :, :2 * H, hdm // 2:].clone()
    c, s = cos[None, :, None], sin[None, :, None]
    ref[:, :, :2 * H] = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)
    got = qkv.clone()
    hd.rope_inplace(got.view(B * S, -1), cos, sin, S, 2 * H, hdm)
    assert rel(got.float(), ref) < tol and torch.equal(got[:, :, 2 * H:], qkv[:, :, 2 * H:])
    gu = torch.randn(333, 2 * 8192, device="cuda").to(dt)
    assert rel(hd.swiglu(gu).float(), gu[:, 8192:].float() * F.silu(gu[:, :8192].float())) < tol


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


@pytest.mark.parametrize("v_tr", [True, False])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 4e-3), (torch.float16, 6e-4)])
def test_flash_attention_vs_fp32_reference(hd, dt, tol, v_tr, monkeypatch):
    """d3d_flash_attention (head_dim 64/96, causal/full, ragged S, masked tail) vs float32 SDPA on the same 16-bit inputs.
    v_tr: V transposed by ds_read_b64_tr_b16 inside the kernel (the default) / the pre-transposed V^T workspace variant."""
    monkeypatch.setattr(type(hd), "V_TR", v_tr)
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


@pytest.mark.parametrize("v_tr", [True, False])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
def test_flash_attention_packed_ragged(hd, dt, tol, causal, v_tr, monkeypatch):
    """Packed variable-length batch (cu_seqlens): sequences of 1 token, below / at / just above a 128-row query block and a
    64-key tile, odd and even block counts (the causal kernel pairs the longest block of a sequence with its shortest), padding
    rows after the last sequence.  Reference: fp32 softmax attention per sequence.  Tolerance = 16-bit output rounding + 16-bit P."""
    monkeypatch.setattr(type(hd), "V_TR", v_tr)
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


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 6e-3), (torch.float16, 1e-3)])
def test_decode_attention_kv_cache(hd, dt, tol):
    """d3d_decode_attention: one query per sequence over [prompt keys read in place from a packed QKV buffer | side cache | the
    current token], ragged prompts, several decode steps; vs fp32 softmax attention over the same (16-bit) keys and values.  The
    fused-RoPE form (un-rotated q, k + cos/sin/pos) must give what rope_inplace followed by the plain form gives."""
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
        assert rel(out_b, out_a) < 1e-3 and torch.equal(kn_a[:, :t + 1], kn_b[:, :t + 1]) and torch.equal(vn_a[:, :t + 1], vn_b[:, :t + 1])
