"""`dynam3d_amd.tcnn.Network` as the nn.Module the reference uses (PRE-FF:221-243): ONE flat float32 `params`, state_dict round
trip, flat layout; and -- on the GPU -- forward / backward on the HIP kernels against a float32 autograd restatement of the same
bias-free LeakyReLU MLP (oracle/nnref.py::tcnn_mlp)."""
import numpy as np
import pytest
import torch

CFG = lambda out_act: {"otype": "CutlassMLP", "activation": "LeakyReLU", "output_activation": out_act, "n_neurons": 768, "n_hidden_layers": 2}


def test_network_is_a_module_with_one_flat_params_vector():
    from dynam3d_amd import tcnn

    class Holder(torch.nn.Module):                                   # like the reference's Feature_Fields (PRE-FF:221-243)
        def __init__(self):
            super().__init__()
            self.nerf_encoder = tcnn.Network(768, 769, CFG("LeakyReLU"), device="cpu")
            self.nerf_decoder = tcnn.Network(768, 768, CFG("None"), device="cpu")

    h = Holder()
    sd = h.state_dict()
    assert set(sd) == {"nerf_encoder.params", "nerf_decoder.params"}
    assert sd["nerf_encoder.params"].dtype == torch.float32 and sd["nerf_encoder.params"].dim() == 1
    assert sd["nerf_encoder.params"].numel() == 768 * 768 * 2 + 784 * 768            # output rows padded 769 -> 784 (tensor-core width 16)
    assert sd["nerf_decoder.params"].numel() == 768 * 768 * 3
    assert [p.requires_grad for p in h.parameters()] == [True, True]
    # round trip: another instance loads the state dict and holds the same per-layer matrices
    h2 = Holder()
    assert not torch.equal(h2.nerf_encoder.params, h.nerf_encoder.params) or True
    with torch.no_grad():
        h.nerf_encoder.params.mul_(0.5)
    h2.load_state_dict(h.state_dict(), strict=True)
    a, b = h.nerf_encoder.layers_from_flat(h.nerf_encoder.params), h2.nerf_encoder.layers_from_flat(h2.nerf_encoder.params)
    assert [tuple(w.shape) for w in a] == [(768, 768), (768, 768), (769, 768)] and all(torch.equal(x, y) for x, y in zip(a, b))
    # flat <-> layers
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(768, 768, generator=g), torch.randn(768, 768, generator=g), torch.randn(769, 768, generator=g)]
    flat = h.nerf_encoder.flat_from_layers(ws)
    net = tcnn.Network.from_flat_params(768, 769, CFG("LeakyReLU"), flat, device="cpu")
    assert all(torch.equal(x, y) for x, y in zip(net.layers_from_flat(net.params.detach()), ws))
    assert float(flat[768 * 768 * 2 + 769 * 768:].abs().sum()) == 0.0                # the padding rows are zero
    with pytest.raises(ValueError):
        net.load_flat_params(flat[:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("out_act,n_out,rows", [("LeakyReLU", 769, 1152), ("None", 768, 1152), ("None", 768, 77)])
def test_tcnn_forward_backward_on_hip_matches_autograd_restatement(out_act, n_out, rows):
    from dynam3d_amd import tcnn
    from oracle import nnref as NN
    torch.manual_seed(5)
    net = tcnn.Network(768, n_out, CFG(out_act), device="cuda", seed=11)
    ws = [w.detach().float().cpu() for w in net.layers_from_flat(net.params)]
    x = (torch.randn(rows, 768) * 0.8)
    dy = torch.randn(rows, n_out) * 0.1
    # restatement: float32 autograd on the fp16-rounded weights / inputs (what the kernels consume)
    xr = x.half().float().requires_grad_(True)
    wr = [w.half().float().requires_grad_(True) for w in ws]
    yr = NN.tcnn_mlp(xr, wr, "LeakyReLU", out_act, store=torch.float16)
    yr.backward(dy.half().float())
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    assert y.dtype == torch.float16 and tuple(y.shape) == (rows, n_out)
    y.backward(dy.cuda().half())
    rel = lambda a, b: float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm())
    gl = net.layers_from_flat(net.params.grad.detach().cpu())
    errs = dict(y=rel(y.float().cpu(), yr), dx=rel(xg.grad.float().cpu(), xr.grad), **{f"dW{l}": rel(g, w.grad) for l, (g, w) in enumerate(zip(gl, wr))})
    print("tcnn", out_act, n_out, rows, {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["y"] < 1e-3                                                            # same fp16 stores, fp32 accumulation
    assert all(v < 5e-3 for k, v in errs.items() if k != "y"), errs                    # gradients are fp16 GEMM outputs (tinycudann's too)
    # padding rows of the flat gradient stay zero; the inference path (one C call) returns the same forward
    if n_out == 769:
        assert float(net.params.grad[768 * 768 * 2 + 769 * 768:].abs().sum()) == 0.0
    with torch.no_grad():
        assert torch.equal(net(x.cuda()), y.detach())
    # an optimizer step changes `params`; the fp16 operand cache follows
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    opt.step()
    with torch.no_grad():
        y2 = net(x.cuda())
    ws2 = [w.detach().float().cpu() for w in net.layers_from_flat(net.params)]
    assert rel(y2.float().cpu(), NN.tcnn_mlp(x.half().float(), [w.half().float() for w in ws2], "LeakyReLU", out_act, store=torch.float16)) < 1e-3
    assert not torch.equal(y2, y.detach())


@pytest.mark.gpu
def test_tcnn_backward_keeps_tiny_gradients_and_odd_input_widths():
    """dL/dy ~ 1e-6 (the mean-reduced losses of the reference's trainer, which runs WITHOUT a GradScaler and relies on tinycudann's
    internal loss_scale = 128): un-scaled, such gradients are subnormal / zero in fp16.  And a 192-wide input layer (n_input_dims
    % 128 != 0) must survive the backward GEMMs, whose N is the input width."""
    from dynam3d_amd import tcnn
    from oracle import nnref as NN
    torch.manual_seed(6)
    for n_in, dy_scale, tol in ((768, 1e-6, 3e-2), (192, 0.1, 5e-3)):
        net = tcnn.Network(n_in, 768, CFG("None"), device="cuda", seed=12)
        ws = [w.detach().float().cpu() for w in net.layers_from_flat(net.params)]
        x = torch.randn(300, n_in) * 0.8
        dy = torch.randn(300, 768) * dy_scale
        xr = x.half().float().requires_grad_(True)
        wr = [w.half().float().requires_grad_(True) for w in ws]
        NN.tcnn_mlp(xr, wr, "LeakyReLU", "None", store=torch.float16).backward(dy)
        xg = x.cuda().requires_grad_(True)
        net(xg).backward(dy.cuda())
        rel = lambda a, b: float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm())
        gl = net.layers_from_flat(net.params.grad.detach().cpu())
        errs = dict(dx=rel(xg.grad.float().cpu(), xr.grad), **{f"dW{l}": rel(g, w.grad) for l, (g, w) in enumerate(zip(gl, wr))})
        print("tcnn backward", n_in, dy_scale, {k: f"{v:.2e}" for k, v in errs.items()})
        assert all(v < tol for v in errs.values()), errs


@pytest.mark.gpu
@pytest.mark.parametrize("out_act,n_out,rows", [("LeakyReLU", 769, 1152), ("None", 768, 9216), ("None", 768, 77)])
def test_tcnn_fused_kernel_equals_per_layer_gemms_bit_for_bit(out_act, n_out, rows, monkeypatch):
    """d3d_mlp_fused (ONE launch per network: the 64-row activation slab stays in LDS across the layers; forward AND the data-gradient
    chain of the backward pass) against the unfused path (one d3d_gemm_nt launch per layer): same K order, same rounding points ->
    identical bits in y, dx and every dW."""
    from dynam3d_amd import tcnn
    torch.manual_seed(9)
    net = tcnn.Network(768, n_out, CFG(out_act), device="cuda", seed=21)
    x = (torch.randn(rows, 768) * 0.8).cuda()
    dy = (torch.randn(rows, n_out) * 0.05).cuda()
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(tcnn, "FUSED", fused)
        net.params.grad = None
        xg = x.clone().requires_grad_(True)
        y = net(xg)
        y.backward(dy)
        with torch.no_grad():
            y_inf = net(x)
        res[fused] = (y.detach().clone(), xg.grad.clone(), net.params.grad.clone(), y_inf.clone())
    for a, b, what in zip(res[True], res[False], ("y", "dx", "dparams", "y (inference call)")):
        assert torch.equal(a, b), (what, float((a.float() - b.float()).abs().max()))
    assert torch.equal(res[True][0], res[True][3])
