"""Training BatchNorm split over its neighbours (SURVEY.md 7 hard part (b); reference: networks/mobilenet_v2.py:42-56, every
InvertedResidual is pw -> BN -> ReLU6 -> dw -> BN -> ReLU6 -> pw -> BN): the producer's epilogue delivers the statistics, one small
finalize launch turns them into scale / shift, the CONSUMER applies act(x * scale + shift) where it loads its input.  The kernels
that take (raw tensor, scale, shift, act) must compute exactly what they compute from the materialised tensor."""
import numpy as np
import pytest
import torch

from pixelpick_amd import _lib
from pixelpick_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def L():
    return _lib.lib()


def st():
    return _lib.current_stream_ptr()


def _affine(C, gen):
    scale = (torch.rand(C, device=DEV, generator=gen) + 0.5) * torch.where(torch.rand(C, device=DEV, generator=gen) < 0.2, -1.0, 1.0)
    shift = torch.randn(C, device=DEV, generator=gen)
    return scale.contiguous(), shift.contiguous()


def _materialise(x, scale, shift, act):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    _lib.check(L().pp_scale_shift_act(x.data_ptr(), C, B * H * W, C, scale.data_ptr(), shift.data_ptr(), None, 0, act, y.data_ptr(), C, st()),
               "scale_shift_act")
    return y


DW_CASES = [(4, 18, 34, 960, 1, 0, 1), (4, 18, 34, 384, 1, 0, 1), (2, 66, 130, 144, 1, 0, 1), (2, 34, 66, 192, 2, 0, 1),
            (3, 20, 36, 960, 1, 0, 2), (2, 17, 23, 96, 2, 1, 1), (2, 13, 19, 32, 1, 1, 1), (1, 130, 258, 96, 2, 0, 1)]


@pytest.mark.parametrize("case", DW_CASES, ids=[str(c) for c in DW_CASES])
@pytest.mark.parametrize("act", [2, 1, 0])
def test_depthwise_with_input_affine_and_statistics(case, act):
    B, H, W, C, stride, pad, dil = case
    gen = torch.Generator(device=DEV).manual_seed(C + H)
    x = torch.randn(B, H, W, C, device=DEV, generator=gen) * 2
    w = torch.randn(3, 3, C, device=DEV, generator=gen)
    scale, shift = _affine(C, gen)
    Ho, Wo = E.out_size(H, 3, stride, pad, dil), E.out_size(W, 3, stride, pad, dil)
    ref = torch.empty(B, Ho, Wo, C, device=DEV)
    xm = _materialise(x, scale, shift, act)
    _lib.check(L().pp_dwconv3x3_fwd(xm.data_ptr(), C, B, H, W, C, w.data_ptr(), stride, pad, dil, ref.data_ptr(), C, st()), "dw")
    rows = int(L().pp_dwconv3x3_fwd_stats_rows(B, H, W, C, stride, pad, dil))
    assert rows > 0
    stats = torch.full((rows, 2, C), float("nan"), device=DEV)
    y = torch.empty_like(ref)
    _lib.check(L().pp_dwconv3x3_fwd_fused(x.data_ptr(), C, B, H, W, C, w.data_ptr(), stride, pad, dil, scale.data_ptr(), shift.data_ptr(), act,
                                          y.data_ptr(), C, stats.data_ptr(), stats.numel(), st()), "dw fused")
    assert torch.equal(y, ref)                                    # same taps, same order, same activated inputs: bit-equal
    s = stats.double().sum(0)
    r = ref.double().reshape(-1, C)
    assert torch.allclose(s[0], r.sum(0), rtol=1e-5, atol=1e-4 * r.abs().sum(0).max().item() / r.shape[0] + 1e-6)
    assert torch.allclose(s[1], (r * r).sum(0), rtol=1e-5)
    # without the affine / without statistics
    y2 = torch.empty_like(ref)
    _lib.check(L().pp_dwconv3x3_fwd_fused(xm.data_ptr(), C, B, H, W, C, w.data_ptr(), stride, pad, dil, None, None, 0, y2.data_ptr(), C,
                                          None, 0, st()), "dw fused plain")
    assert torch.equal(y2, ref)
    # weight gradient from (raw, scale, shift, act) == from the materialised input
    dy = torch.randn(B, Ho, Wo, C, device=DEV, generator=gen)
    ws = torch.empty(int(L().pp_colreduce_workspace_bytes(B * Ho * Wo, C)) * 9 + 1024, dtype=torch.uint8, device=DEV)
    dw0, dw1 = torch.empty(3, 3, C, device=DEV), torch.empty(3, 3, C, device=DEV)
    _lib.check(L().pp_dwconv3x3_bwd_weight(xm.data_ptr(), C, B, H, W, C, dy.data_ptr(), C, stride, pad, dil, dw0.data_ptr(), ws.data_ptr(),
                                           ws.numel(), st()), "dw wgrad")
    _lib.check(L().pp_dwconv3x3_bwd_weight_affine_in(x.data_ptr(), C, B, H, W, C, scale.data_ptr(), shift.data_ptr(), act, dy.data_ptr(), C,
                                                     stride, pad, dil, dw1.data_ptr(), ws.data_ptr(), ws.numel(), st()), "dw wgrad affine")
    assert torch.equal(dw0, dw1)


@pytest.mark.parametrize("M,C,rows", [(2448, 960, 77), (131072, 32, 4096), (2048, 64, 3), (7, 24, 1)])
def test_finalize_partials_matches_batchnorm_statistics(M, C, rows):
    gen = torch.Generator(device=DEV).manual_seed(M + C)
    x = torch.randn(M, C, device=DEV, generator=gen) * 1.7 + 0.4
    # partial rows: any partition of the M rows into `rows` groups
    bounds = np.linspace(0, M, rows + 1).astype(np.int64)
    stats = torch.stack([torch.stack([x[a:b].double().sum(0), (x[a:b].double() ** 2).sum(0)]) for a, b in zip(bounds[:-1], bounds[1:])]).float().contiguous()
    gamma, beta = torch.rand(C, device=DEV, generator=gen) + 0.5, torch.randn(C, device=DEV, generator=gen)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    _lib.check(L().pp_bn_finalize_partials(stats.data_ptr(), rows, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(),
                                           mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), st()), "finalize")
    bn = torch.nn.BatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
        yr = bn(x.t().reshape(1, C, M, 1))
    mu, var = x.double().mean(0), x.double().var(0, unbiased=False)
    assert torch.allclose(mean.double(), mu, rtol=1e-5, atol=1e-6) and torch.allclose(invstd.double(), 1 / torch.sqrt(var + 1e-5), rtol=1e-5)
    assert torch.allclose(rm, bn.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(rv, bn.running_var, rtol=1e-5, atol=1e-6)
    y = x * scale + shift
    assert torch.allclose(y, yr.reshape(C, M).t(), rtol=1e-4, atol=1e-4)


PW_CASES = [(4, 16, 32, 960, 160), (4, 16, 32, 960, 320), (4, 16, 32, 576, 96), (4, 16, 32, 384, 64), (3, 23, 30, 576, 160),
            (4, 32, 64, 192, 32), (4, 64, 128, 144, 24), (4, 128, 256, 32, 16), (2, 64, 128, 96, 24), (2, 9, 13, 776, 68)]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(c) for c in PW_CASES])
@pytest.mark.parametrize("act", [2, 0])
def test_pointwise_conv_with_input_affine(case, act):
    B, H, W, Cin, Cout = case
    assert L().pp_conv2d_fwd_accepts_affine_in(B, H, W, Cin, Cout, 1, 1, 1, 0, 1) == 1
    gen = torch.Generator(device=DEV).manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=gen) * 2
    w = torch.randn(1, 1, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin)
    scale, shift = _affine(Cin, gen)
    xm = _materialise(x, scale, shift, act)
    wsn = int(L().pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, 1, 1, 1, 0, 1))
    ws = torch.empty(max(wsn, 256), dtype=torch.uint8, device=DEV)
    ref, y = torch.empty(B, H, W, Cout, device=DEV), torch.full((B, H, W, Cout), float("nan"), device=DEV)
    L().pp_debug_set_conv_rows(1)          # the reference through the same MFMA kernel (the whole-row VALU kernels add in another order)
    try:
        _lib.check(L().pp_conv2d_fwd(xm.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, 1, 1, 1, 0, 1, ref.data_ptr(), Cout, Cout,
                                     ws.data_ptr() if wsn else None, wsn, st()), "conv")
    finally:
        L().pp_debug_set_conv_rows(0)
    _lib.check(L().pp_conv2d_fwd_affine_in(x.data_ptr(), Cin, B, H, W, Cin, scale.data_ptr(), shift.data_ptr(), act, w.data_ptr(), None, 1, 1, 1,
                                           0, 1, y.data_ptr(), Cout, Cout, ws.data_ptr() if wsn else None, wsn, st()), "conv affine")
    assert torch.equal(y, ref)                                    # same products in the same order
    tr = torch.nn.functional.conv2d(xm.permute(0, 3, 1, 2).cpu(), w.permute(3, 2, 0, 1).cpu()).permute(0, 2, 3, 1)
    assert (y.cpu() - tr).abs().max().item() <= 2e-4 * tr.abs().max().item()


def test_shapes_without_an_input_affine_kernel_are_refused():
    assert L().pp_conv2d_fwd_accepts_affine_in(4, 16, 32, 192, 64, 1, 1, 1, 0, 1) == 0        # K < 256, 64 outputs: tiled LDS-DMA kernel
    assert L().pp_conv2d_fwd_accepts_affine_in(4, 16, 32, 960, 160, 1, 1, 1, 1, 1) == 0       # padding
    assert L().pp_conv2d_fwd_accepts_affine_in(4, 16, 32, 320, 256, 3, 3, 1, 6, 6) == 0
    x = torch.randn(4, 16, 32, 192, device=DEV)
    w = torch.randn(1, 1, 192, 64, device=DEV)
    s = torch.ones(192, device=DEV)
    y = torch.empty(4, 16, 32, 64, device=DEV)
    rc = L().pp_conv2d_fwd_affine_in(x.data_ptr(), 192, 4, 16, 32, 192, s.data_ptr(), s.data_ptr(), 2, w.data_ptr(), None, 1, 1, 1, 0, 1,
                                     y.data_ptr(), 64, 64, None, 0, st())
    assert rc != 0
