"""GPU: the split-precision float32 GEMM (csrc/f32x3_kernels.hip: fp16 hi + lo operands, three MFMAs, float32 accumulation) against
float64, next to the float32-MFMA kernel it replaces in the inference token builder: its error must be of float32 size -- within a
small factor of the float32 kernel's own distance from float64 -- on the shapes of the set encoders / merge discriminator /
prefix MLPs, with every epilogue, ragged M and N, operands spanning 1e-4 .. 1e3, and exact zeros / a padded K."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops(split):
    from dynam3d_amd.f32_ops import F32Ops
    o = F32Ops()
    o.SPLIT = split                                  # instance attribute shadows the class default
    return o


SHAPES = [(4858, 2304, 768), (4858, 768, 768), (4858, 2048, 768), (4858, 768, 2048), (300, 768, 1539), (37, 2304, 768), (1, 768, 768),
          (129, 12, 768), (8 * 576, 3072, 768)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_split_gemm_has_float32_accuracy(M, N, K):
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda")
    ref = x.double() @ w.double().t() + b.double()
    f32, x3 = _ops(False), _ops(True)
    for act, res in ((None, None), ("gelu", None), (None, r)):
        want = ref
        if act == "gelu":
            want = torch.nn.functional.gelu(ref)
        if res is not None:
            want = ref + r.double()
        e32 = float((f32.linear(x, w, b, act, res).double() - want).norm() / want.norm())
        e3 = float((x3.linear(x, w, b, act, res).double() - want).norm() / want.norm())
        print(f"M {M} N {N} K {K} act {act} res {res is not None}: rel-L2 vs float64  f32 kernel {e32:.2e}   split kernel {e3:.2e}")
        assert e3 < 1.5e-6 and e3 < 2 * e32 + 1e-7, (e32, e3)      # (the float32 kernel is a k-ordered fmaf chain: ~sqrt(K) * 2^-24 itself)


def test_split_gemm_wide_dynamic_range_and_zeros():
    """Per-element magnitudes from 1e-4 to 1e3 inside one row (lo parts of the small ones are fp16 subnormals -- kept by the MFMA), whole
    zero rows / columns, and a K that is zero-padded (1539 -> 1568)."""
    torch.manual_seed(3)
    M, N, K = 700, 768, 1539
    x = torch.randn(M, K, device="cuda") * torch.pow(10.0, torch.empty(M, K, device="cuda").uniform_(-4, 3))
    w = torch.randn(N, K, device="cuda") * torch.pow(10.0, torch.empty(N, K, device="cuda").uniform_(-4, 0))
    x[5] = 0
    w[7] = 0
    b = torch.zeros(N, device="cuda")
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()                     # the sum of |terms|: what a rounding error is relative to
    f32, x3 = _ops(False), _ops(True)
    y32, y3 = f32.linear(x, w, b).double(), x3.linear(x, w, b).double()
    assert float(y3[5].abs().max()) == 0.0 and float(y3[:, 7].abs().max()) == 0.0
    ok = scale > 0
    e32 = float(((y32 - ref).abs()[ok] / scale[ok]).max())
    e3 = float(((y3 - ref).abs()[ok] / scale[ok]).max())
    print(f"max |err| / sum|terms|:  f32 kernel {e32:.2e}   split kernel {e3:.2e}")
    assert e3 < 1e-6, (e32, e3)


def test_split_gemm_refuses_bad_shapes():
    from dynam3d_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 48, device="cuda")
    w = torch.zeros(64, 48, device="cuda")
    y = torch.zeros(64, 64, device="cuda")
    rc = lib.d3d_gemm_nt_f32x3(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()), None, None, 64, 64, 48, 48, 48, 64, 0,
                               None, None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0                                                      # K % 32 != 0


def test_row_scaling_removes_the_fp16_range_contract():
    """Operands spanning 1e-6 .. 1e3 ACROSS rows, far beyond fp16 on both sides: rows of ~1e5 (the unscaled kernel: inf), rows of ~1e-6
    (the unscaled kernel: fp16 subnormals, a few bits) and rows mixing 1e-6 .. 1e3 -- against float64, error relative to sum|terms|."""
    torch.manual_seed(11)
    M, N, K = 640, 768, 768
    x = torch.randn(M, K, device="cuda")
    x[:128] *= 3e5                                                    # beyond 65504
    x[128:256] *= 1e-6                                                # uniformly tiny rows
    x[256:384] *= torch.pow(10.0, torch.empty(128, K, device="cuda").uniform_(-6, 3))
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    w[:64] *= 1e4
    w[64:128] *= 1e-5
    b = torch.zeros(N, device="cuda")
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()
    x3 = _ops(True)
    y = x3.linear(x, w, b).double()
    x3.check_status()                                                 # nothing non-finite
    assert bool(torch.isfinite(y).all())
    e = float(((y - ref).abs() / scale).max())
    y32 = _ops(False).linear(x, w, b).double()
    e32 = float(((y32 - ref).abs() / scale).max())
    print(f"scaled split kernel, rows from 1e-6 to 3e5: max |err| / sum|terms| {e:.2e} (float32-MFMA kernel {e32:.2e})")
    assert e < 1e-6, e
    un = _ops(True)
    un.SCALE = False
    yu = un.linear(x, w, b)
    assert not bool(torch.isfinite(yu[:128]).all())                   # the round-3 kernel: rows beyond fp16 overflow ...
    import pytest as _pt
    with _pt.raises(FloatingPointError):
        un.check_status()                                             # ... and the device status word says so
    eu = float(((yu[128:256].double() - ref[128:256]).abs() / scale[128:256]).max())
    es = float(((y[128:256] - ref[128:256]).abs() / scale[128:256]).max())
    print(f"uniformly tiny rows: unscaled {eu:.2e}, scaled {es:.2e}")
    assert es < 1e-6 < eu


def test_one_ulp_apart_inputs_keep_their_order_and_ties_stay_ties():
    """Merge-discriminator input rows one float32 ulp apart in one element.  Identical rows give identical bits (a tie in float32 stays
    a tie: the kernel is a fixed-order sum).  A +1 ulp input moves an output in the direction of its weight or not at all -- the hi + lo
    representation is monotone in x; the one exception is a step that carries from lo into hi (probability ~2^-13 per element), where
    the dropped a_lo w_lo term jumps by <= 2^-22 |a w|: float32 summation-order size, bounded below."""
    torch.manual_seed(5)
    K, N = 1568, 3072
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.zeros(N, device="cuda")
    base = torch.randn(1, K, device="cuda")
    rows = base.repeat(257, 1)
    idx = torch.arange(256, device="cuda") * 6
    up = torch.nextafter(base[0, idx], torch.full_like(base[0, idx], float("inf")))
    rows[torch.arange(256, device="cuda") + 1, idx] = up              # row r + 1 = base with element idx[r] one ulp up
    x3 = _ops(True)
    y = x3.linear(rows, w, b)
    twin = x3.linear(torch.cat([base, base], 0), w, b)
    assert torch.equal(twin[0], twin[1]) and torch.equal(twin[0], y[0])
    d = (y[1:] - y[:1]).double()                                      # (256, N): effect of +1 ulp in element idx[r]
    sign_w = torch.sign(w[:, idx].t().double())                       # (256, N)
    wrong = d * sign_w < 0
    assert float(wrong.double().mean()) < 1e-3, float(wrong.double().mean())
    assert float((d != 0).double().mean()) > 1e-3                     # (the ulp is not simply lost: some outputs do move -- 1.5 % measured)


def test_status_word_is_polled_one_update_late_without_a_sync():
    x3 = _ops(True)
    x = torch.randn(64, 768, device="cuda")
    x[3, 5] = float("nan")
    w = torch.randn(768, 768, device="cuda")
    x3.linear(x, w, torch.zeros(768, device="cuda"))
    x3.snapshot_status()
    torch.cuda.synchronize()
    import pytest as _pt
    with _pt.raises(FloatingPointError):
        x3.poll_status()
    x3.poll_status()                                                  # cleared
