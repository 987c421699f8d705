"""GPU, at the BENCHMARK's full sizes (BASELINE configs[2]: 8 prompts x ~860 rows -> M = 6912 packed rows, Phi-3-mini widths, ViT-L
M = 4616): properties that hold exactly whatever the size, so no float32 oracle has to run at that size --

  * GEMM rows are independent: permuting the rows of A permutes the rows of C bit for bit (every tile shape, the M remainder, the
    split-K tail and its fix-up, the transposed epilogue with residual);
  * scaling A by a power of two scales C by it bit for bit (fp32 accumulation + one 16-bit store: exact absent overflow);
  * launches are deterministic;
  * causal attention: the output rows of a prefix do not change when later rows change; packed == per-sequence evaluation;
  * the decode attention over a KV cache equals the last row of the prefill attention over the same keys (two different kernels).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hd():
    from dynam3d_amd.hip_dense import HipDense
    return HipDense()


SHAPES = [("phi3.qkv", 6912, 9216, 3072, torch.bfloat16), ("phi3.o", 6912, 3072, 3072, torch.bfloat16),
          ("phi3.gate_up", 6912, 16384, 3072, torch.bfloat16), ("phi3.down", 6912, 3072, 8192, torch.bfloat16),
          ("vit.qkv", 4616, 3072, 1024, torch.float16), ("vit.fc2", 4616, 1024, 4096, torch.float16)]


@pytest.mark.parametrize("name,M,N,K,dt", SHAPES)
def test_gemm_row_permutation_scaling_determinism(hd, name, M, N, K, dt):
    from dynam3d_amd.hip_dense import interleave_gate_up
    torch.manual_seed(hash(name) % 1000)
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    r = torch.randn(M, N, device="cuda").to(dt)
    b = (torch.randn(N, device="cuda") * 0.1).to(dt)
    perm = torch.randperm(M, device="cuda")
    if name == "phi3.gate_up":
        wi = interleave_gate_up(w)
        f = lambda a, res: hd.linear_swiglu(a, wi)
    elif name in ("phi3.o", "phi3.down"):
        f = lambda a, res: hd.linear(a, w, None, None, res)                       # residual epilogue (+ split-K tail and fix-up)
    elif name == "vit.fc2":
        f = lambda a, res: hd.linear(a, w, b, None, res)                          # bias + residual
    else:
        f = lambda a, res: hd.linear(a, w, None if dt == torch.bfloat16 else b, None)
    y = f(x, r)
    assert torch.equal(y, f(x, r))                                                # deterministic
    assert torch.isfinite(y.float()).all()
    yp = f(x[perm].contiguous(), r[perm].contiguous())
    if name in ("phi3.o", "phi3.down"):
        # These grids end in a split-K tail: a row that lands in a tail tile is the sum of 3-5 K-slice partials, a row in a full-round
        # tile one accumulation chain -- the same products in another fp32 summation order, so a row may move by one 16-bit unit in the
        # last place when the permutation moves it across that boundary.  Everything else about the row is position-independent.
        d = (yp.float() - y[perm].float()).abs()
        assert float((d > 0).float().mean()) < 0.15 and float(d.max() / y.float().abs().max()) < 2 ** -7
        same = perm[:256 * 8] < 256 * 8                                              # rows that stay inside the first 8 tile rows (full rounds)
        assert torch.equal(yp[:256 * 8][same], y[perm[:256 * 8]][same])
    else:
        assert torch.equal(yp, y[perm])                                           # rows are independent, wherever they land in the grid
    if name in ("phi3.qkv", "vit.qkv") and dt == torch.bfloat16:
        y2 = f((x.float() * 2).to(dt), r)                                         # exact scaling (no bias / residual / nonlinearity here)
        assert torch.equal(y2.float(), y.float() * 2)


def _packed_qkv(lens, H, hd_, dt, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    T = sum(lens)
    Tp = (T + 255) // 256 * 256
    qkv = torch.zeros(Tp, 3 * H, hd_, device="cuda", dtype=dt)
    qkv[:T] = (torch.randn(T, 3 * H, hd_, device="cuda", generator=g) * 0.7).to(dt)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    return qkv, cu, T, Tp


def test_packed_causal_attention_prefix_property_full_size(hd):
    """8 prompts of the benchmark's lengths, 32 heads x 96: (1) changing the LAST 100 rows of every prompt leaves the outputs of the
    rows before them bit-identical (causality through every tile / masking path); (2) packed evaluation == each prompt alone."""
    lens = [861, 836, 1109, 793, 970, 765, 780, 774]
    H, d = 32, 96
    qkv, cu, T, Tp = _packed_qkv(lens, H, d, torch.bfloat16, 1)
    out = hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T)
    assert torch.isfinite(out.float()).all()
    qkv2 = qkv.clone()
    keep = torch.ones(Tp, dtype=torch.bool, device="cuda")
    o = 0
    for n in lens:
        qkv2[o + n - 100:o + n] = (torch.randn(100, 3 * H, d, device="cuda") * 0.7).to(torch.bfloat16)
        # (rows that share a 128-query block with a changed row are left out: the kernel's deferred-max vote is taken per wave of 32
        #  queries, so a neighbour's scores may move a row's intermediate rescaling -- same mathematics, not the same bits)
        keep[o + (n - 100) // 128 * 128:o + n] = False
        o += n
    keep[T:] = False
    out2 = hd.attention_packed(qkv2, H, True, cu, len(lens), max(lens), n_valid=T)
    assert torch.equal(out2[keep], out[keep])
    o = 0
    for b, n in enumerate(lens[:3]):                                              # packed == alone (the same kernel on one sequence)
        one = torch.zeros((n + 255) // 256 * 256, 3 * H, d, device="cuda", dtype=torch.bfloat16)
        one[:n] = qkv[o:o + n]
        cu1 = torch.tensor([0, n], dtype=torch.int32, device="cuda")
        o1 = hd.attention_packed(one, H, True, cu1, 1, n, n_valid=n)
        assert torch.equal(o1[:n], out[o:o + n]), b
        o += n


def test_decode_attention_equals_last_prefill_row(hd):
    """Two different kernels, one definition: the flash-attention output of a prompt's LAST row == d3d_decode_attention with that row as
    the new token over the other rows as the KV cache (rotated inputs, no RoPE inside) -- at the benchmark's 8 x 32 x 96 and ~860 keys."""
    lens = [861, 836, 1109, 793, 970, 765, 780, 774]
    H, d = 32, 96
    qkv, cu, T, Tp = _packed_qkv(lens, H, d, torch.bfloat16, 2)
    out = hd.attention_packed(qkv, H, True, cu, len(lens), max(lens), n_valid=T).view(Tp, H * d)
    last = (cu[1:] - 1).long()
    new = qkv.view(Tp, -1)[last].contiguous()                                     # (B, 3*H*d): the last rows as "this step's projection"
    # the cache = every prompt WITHOUT its last row: same buffer, sequence b = rows [cu[b], cu[b+1] - 1)  -> shift by one row per prompt
    rows = torch.cat([torch.arange(int(cu[b]), int(cu[b + 1]) - 1, device="cuda") for b in range(len(lens))])
    cache = qkv.view(Tp, -1)[rows].contiguous()
    cu_c = torch.tensor([0] + list(np.cumsum([n - 1 for n in lens])), dtype=torch.int32, device="cuda")
    kn = torch.zeros(len(lens), 2, H, d, dtype=torch.bfloat16, device="cuda")
    vn = torch.zeros_like(kn)
    dec = hd.decode_attention(new, cache, cu_c, kn, vn, H, 0, max(lens) - 1)
    ref = out[last]
    err = float((dec.float() - ref.float()).norm() / ref.float().norm())
    assert err < 4e-3, err                                                        # one bf16 store each, different summation orders
