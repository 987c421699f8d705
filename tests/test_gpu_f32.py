"""GPU: float32 dense kernels of the 3D-token builder (csrc/f32_kernels.hip) against float64 PyTorch references: the fp32 MFMA
GEMM (v_mfma_f32_16x16x4_f32: an exact f32 multiply-add chain) with its fused epilogues, ragged M, N not a multiple of the tile,
K zero-padded (the 1539-wide merge input); the tiny-K / tiny-N linears; LayerNorm with fused residual add / GELU; and FFDense's
GPU set encoder against the padded PyTorch path it replaces."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture(scope="module")
def f32():
    from dynam3d_amd.f32_ops import F32Ops
    return F32Ops()


def test_gemm_f32_epilogues_and_shapes(f32):
    torch.manual_seed(0)
    for M, N, K in ((4736, 2304, 768), (300, 768, 3072), (77, 3072, 1539), (1, 768, 768), (129, 132, 16), (4752, 3072, 768)):
        x = torch.randn(M, K, device="cuda") * 0.7
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        w[3] *= 4.0                                                   # asymmetric operands
        x[:, 5] += 1.0
        b = torch.randn(N, device="cuda") * 0.3
        r = torch.randn(M, N, device="cuda")
        y64 = x.double() @ w.double().t() + b.double()
        assert rel(f32.linear(x, w, b), y64) < 2e-6, (M, N, K)
        assert rel(f32.linear(x, w, b, act="gelu"), F.gelu(y64)) < 2e-6, (M, N, K, "gelu")
        assert rel(f32.linear(x, w, b, residual=r), y64 + r.double()) < 2e-6, (M, N, K, "res")
        assert torch.equal(f32.linear(x, w, b), f32.linear(x, w, b))               # deterministic
    # strided input (a column slice of a wider buffer)
    big = torch.randn(500, 768 + 32, device="cuda")
    w = torch.randn(768, 768, device="cuda") * 768 ** -0.5
    b = torch.zeros(768, device="cuda")
    assert rel(f32.linear(big[:, :768], w, b), big[:, :768].double() @ w.double().t()) < 2e-6


def test_small_linears_and_layer_norm(f32):
    torch.manual_seed(1)
    for K, N in ((7, 768), (4, 768), (3, 768), (6, 3072), (8, 96)):
        x, w, b = torch.randn(4608, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
        assert rel(f32.linear(x, w, b), x.double() @ w.double().t() + b.double()) < 1e-6
    for K in (3072, 768, 96):
        x, w, b = torch.randn(301, K, device="cuda"), torch.randn(2, K, device="cuda") * K ** -0.5, torch.randn(2, device="cuda")
        assert rel(f32.linear(x, w, b), x.double() @ w.double().t() + b.double()) < 1e-6
    for D in (768, 3072, 96, 384):
        x, r = torch.randn(1001, D, device="cuda") * 2 + 0.3, torch.randn(1001, D, device="cuda")
        w, b = torch.randn(D, device="cuda") * 0.2 + 1, torch.randn(D, device="cuda") * 0.1
        for eps in (1e-5, 1e-12):
            ref = F.layer_norm(x.double(), (D,), w.double(), b.double(), eps)
            assert rel(f32.layer_norm(x, w, b, eps), ref) < 2e-6
            assert rel(f32.layer_norm(x, w, b, eps, gelu=True), F.gelu(ref)) < 2e-6
            assert rel(f32.layer_norm(x, w, b, eps, residual=r), F.layer_norm((x + r).double(), (D,), w.double(), b.double(), eps)) < 2e-6


def test_ffdense_gpu_encoder_matches_padded_torch_path():
    """FFDense on the GPU (packed sets, fp32 MFMA GEMMs, fused LN, varlen set attention) == its padded PyTorch expression on the CPU."""
    from dynam3d_amd.ff_dense import FFDense
    from dynam3d_amd.weights import ff_param_spec, synth_state_dict
    sd = synth_state_dict(ff_param_spec(), seed=0)
    g, c = FFDense(sd, "cuda"), FFDense(sd, "cpu")
    rng = np.random.default_rng(3)
    lens = [1, 37, 2, 64, 300, 5, 17]
    T = sum(lens)
    fts = torch.from_numpy(rng.standard_normal((T, 768)).astype(np.float32))
    geom = torch.from_numpy(rng.standard_normal((T, 7)).astype(np.float32))
    a = g.encode_patch_sets(fts.cuda(), geom.cuda(), lens).cpu()
    b = c.encode_patch_sets(fts, geom, lens)
    assert rel(a, b) < 2e-5, rel(a, b)
    zl = [3, 1, 9]
    zf = torch.from_numpy(rng.standard_normal((sum(zl), 768)).astype(np.float32))
    zg = torch.from_numpy(rng.standard_normal((sum(zl), 4)).astype(np.float32))
    assert rel(g.encode_zone_sets(zf.cuda(), zg.cuda(), zl).cpu(), c.encode_zone_sets(zf, zg, zl)) < 2e-5
    x = torch.from_numpy(rng.standard_normal((23, 1539)).astype(np.float32))
    assert rel(g.merge_logits(x.cuda()).cpu(), c.merge_logits(x)) < 2e-5
