"""The backward kernels of the pre-training step (d3d_layer_norm_bwd_f32, d3d_set_attention_bwd, d3d_composite_bwd)
against float64 PyTorch autograd of the same expressions: forward values and every gradient."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30))


@pytest.mark.parametrize("rows,D,gelu", [(1, 768, False), (200, 768, True), (577, 3072, True), (64, 256, False), (130, 8, True)])
def test_layer_norm_backward(rows, D, gelu):
    from dynam3d_amd.train_ops import layer_norm
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 2.0 + 0.3
    w, b = 1.0 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(rows, D, generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.layer_norm(xd, (D,), wd, bd, 1e-5)
    yr = F.gelu(yr) if gelu else yr
    yr.backward(dy.double())
    xc, wc, bc = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = layer_norm(xc, wc, bc, 1e-5, gelu=gelu)
    y.backward(dy.cuda())
    assert rel(y, yr) < 2e-6
    assert rel(xc.grad, xd.grad) < 5e-6 and rel(wc.grad, wd.grad) < 5e-6 and rel(bc.grad, bd.grad) < 5e-6
    xc2 = x.cuda().requires_grad_(True)                              # deterministic: partial sums are added in a fixed order
    wc2, bc2 = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    layer_norm(xc2, wc2, bc2, 1e-5, gelu=gelu).backward(dy.cuda())
    assert torch.equal(xc2.grad, xc.grad) and torch.equal(wc2.grad, wc.grad) and torch.equal(bc2.grad, bc.grad)


@pytest.mark.parametrize("q_rows", [0, 1])
def test_set_attention_backward(q_rows):
    from dynam3d_amd.train_ops import set_attention
    H = 12
    lens = [1, 37, 64, 65, 130, 5]
    T = sum(lens)
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(T, 3 * H * 64, generator=g) * 0.7
    dout = torch.randn(T, H * 64, generator=g)
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    qd = qkv.double().requires_grad_(True)
    ref = set_attention(qd, off, lens, H, q_rows)                    # the CPU expression (per-set SDPA), float64
    ref.backward(dout.double())
    qc = qkv.cuda().requires_grad_(True)
    out = set_attention(qc, off.cuda(), lens, H, q_rows)
    out.backward(dout.cuda())
    assert rel(out, ref) < 3e-6
    assert rel(qc.grad, qd.grad) < 1e-5
    qc2 = qkv.cuda().requires_grad_(True)
    set_attention(qc2, off.cuda(), lens, H, q_rows).backward(dout.cuda())
    assert torch.equal(qc2.grad, qc.grad)


def test_composite_backward():
    from dynam3d_amd.train_ops import composite, composite_reference
    n, S, N, Fd = 290, 8, 501, 768
    g = torch.Generator().manual_seed(5)
    feat = (torch.randn(n * S, Fd, generator=g)).half().float()
    dens = (torch.randn(n * S, generator=g) * 2.0).half().float()
    dens[::11] = 25.0                                                # softplus threshold branch
    topk = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(n)]).int()
    topk[3] = torch.tensor([500, 0, 7, 8, 9, 250, 499, 1])          # the open-ended last bin
    rel_dist = torch.linspace(0.0, 10.0, N).half().float()
    gout = torch.randn(n, Fd, generator=g)
    fd, dd = feat.double().requires_grad_(True), dens.double().requires_grad_(True)
    fr, dr = composite_reference(fd, dd, rel_dist.double(), topk, N)
    fr.backward(gout.double())
    fc, dc = feat.cuda().requires_grad_(True), dens.cuda().requires_grad_(True)
    fm, depth = composite(fc, dc, rel_dist.cuda(), topk.cuda(), N)
    fm.backward(gout.cuda())
    assert rel(fm, fr) < 2e-6 and rel(depth, dr) < 2e-5
    assert rel(fc.grad, fd.grad) < 2e-5, rel(fc.grad, fd.grad)
    assert rel(dc.grad, dd.grad) < 2e-4, rel(dc.grad, dd.grad)
