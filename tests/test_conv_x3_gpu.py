"""conv_x3_kernel: the large-tile convolutions (SegmentHead 3x3, decoders.py:107-114; ResNet50 bottlenecks, resnet_models.py:58-94)
on the bf16 matrix pipe with every fp32 operand split into three bf16 planes (six MFMAs per product).  The claim that has to hold
is that this is an fp32 convolution: its error against a float64 evaluation must be that of the fp32-MFMA kernels (both are
measured here), forward and backward-data, including padding taps, ragged rows / channels, bias and channel-slice inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pixelpick_amd import _lib
from pixelpick_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    (4, 64, 128, 304, 256, 3, 1, 1, False),     # SegmentHead conv1 at the BASELINE shape
    (4, 64, 128, 256, 256, 3, 1, 1, False),     # SegmentHead conv2
    (3, 90, 100, 256, 304, 3, 1, 1, True),      # ragged rows (27000), 304 outputs (ragged tile columns), bias
    (2, 128, 128, 1280, 256, 1, 0, 1, False),   # pointwise, 32768 rows
    (4, 64, 128, 300, 256, 3, 2, 2, False),     # Cin not a multiple of 16 (pads to 304), dilation 2
    (4, 96, 160, 128, 128, 3, 1, 1, True),      # FPN UpsampleBlock conv (61440 rows)
]


def _run(case, x3):
    B, H, W, Cin, Cout, k, pad, dil, has_bias = case
    L = _lib.lib()
    L.pp_debug_set_x3(1 if x3 else 0)
    try:
        gen = torch.Generator(device=DEV).manual_seed(Cin * 7 + Cout)
        x = torch.randn(B, H, W, Cin, device=DEV, generator=gen) * torch.exp(torch.randn(B, H, W, Cin, device=DEV, generator=gen))   # wide dynamic range
        w = torch.randn(k, k, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin * k * k)
        bias = torch.randn(Cout, device=DEV, generator=gen) if has_bias else None
        Ho, Wo = E.out_size(H, k, 1, pad, dil), E.out_size(W, k, 1, pad, dil)
        dy = torch.randn(B, Ho, Wo, Cout, device=DEV, generator=gen)
        tape = E.Tape()
        xv = E.Var(x)
        yv = E.conv2d(tape, xv, w.requires_grad_(True), bias, 1, pad, dil)
        y = yv.t.clone()
        tape.backward(yv, dy)
        torch.cuda.synchronize()
        return x, w.detach(), bias, dy, y, xv.grad.clone(), tape.param_grads[id(w)].clone()
    finally:
        L.pp_debug_set_x3(1)


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_bf16x3_convolution_is_an_fp32_convolution(case):
    B, H, W, Cin, Cout, k, pad, dil, has_bias = case
    L = _lib.lib()
    L.pp_debug_set_x3(1)
    assert L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, k, k, 1, pad, dil) >= 3 * 2 * (B * H * W + 1) * Cin, "case is not a large-tile layer"
    x, w, bias, dy, y3, dx3, dw3 = _run(case, True)
    _, _, _, _, y1, dx1, dw1 = _run(case, False)
    xd, wd = x.double().permute(0, 3, 1, 2).cpu(), w.double().permute(3, 2, 0, 1).cpu()
    xd.requires_grad_(True)
    wd.requires_grad_(True)
    ref = F.conv2d(xd, wd, bias.double().cpu() if has_bias else None, 1, pad, dil)
    ref.backward(dy.double().permute(0, 3, 1, 2).cpu())
    ref_y, ref_dx = ref.detach().permute(0, 2, 3, 1), xd.grad.permute(0, 2, 3, 1)

    def err(a, r):
        return ((a.double().cpu() - r).abs().max() / r.abs().max()).item(), ((a.double().cpu() - r).norm() / r.norm()).item()

    e3, e1 = err(y3, ref_y), err(y1, ref_y)
    d3, d1 = err(dx3, ref_dx), err(dx1, ref_dx)
    ref_dw = wd.grad.permute(2, 3, 1, 0)                                 # OIHW -> HWIO
    g3, g1 = err(dw3, ref_dw), err(dw1, ref_dw)
    print(f"[x3] weight gradient max/l2 rel err vs fp64: bf16x3 {g3[0]:.2e}/{g3[1]:.2e}  fp32-MFMA {g1[0]:.2e}/{g1[1]:.2e}")
    assert g3[1] <= max(2.0 * g1[1], 3e-7) and g3[0] <= 5e-6
    print(f"\n[x3] fwd max/l2 rel err vs fp64: bf16x3 {e3[0]:.2e}/{e3[1]:.2e}  fp32-MFMA {e1[0]:.2e}/{e1[1]:.2e} | "
          f"bwd-data: bf16x3 {d3[0]:.2e}/{d3[1]:.2e}  fp32-MFMA {d1[0]:.2e}/{d1[1]:.2e}")
    # the same class of error as the fp32 kernels (both are a few 1e-7 in l2), far inside the op-level bar of 1e-4
    assert e3[1] <= max(2.0 * e1[1], 3e-7) and d3[1] <= max(2.0 * d1[1], 3e-7)
    assert e3[0] <= 5e-6 and d3[0] <= 5e-6
    # bit-reproducible
    _, _, _, _, y3b, dx3b, dw3b = _run(case, True)
    assert torch.equal(y3, y3b) and torch.equal(dx3, dx3b) and torch.equal(dw3, dw3b)


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[3]], ids=[str(CASES[0]), str(CASES[2]), str(CASES[3])])
def test_shared_operand_planes_are_bit_identical(case, monkeypatch):
    """PIXELPICK_X3_SHARE: the forward's bf16 planes of x are kept for the weight gradient and dy is split once for backward-data
    and the weight gradient (pp_x3_split + the *_pre entry points) - the same planes the calls would have made themselves, so
    forward, both gradients and the bias gradient are bit-equal to the every-call-splits path; and pp_x3_split + pp_conv2d_fwd_pre
    called by hand equal pp_conv2d_fwd."""
    monkeypatch.setattr(E, "_X3_SHARE", True)
    E._WS_BYTES.clear()
    _, _, _, _, y1, dx1, dw1 = _run(case, True)
    monkeypatch.setattr(E, "_X3_SHARE", False)
    _, _, _, _, y0, dx0, dw0 = _run(case, True)
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0) and torch.equal(dw1, dw0)
    B, H, W, Cin, Cout, k, pad, dil, has_bias = case
    L = _lib.lib()
    nb = L.pp_conv2d_x3_planes_bytes(0, B, H, W, Cin, Cout, k, k, 1, pad, dil)
    if nb:
        assert nb == L.pp_x3_planes_bytes(B * H * W, Cin)
        gen = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(B, H, W, Cin, device=DEV, generator=gen)
        w = torch.randn(k, k, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin * k * k)
        st = torch.cuda.current_stream().cuda_stream
        planes = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _lib.check(L.pp_x3_split(x.data_ptr(), Cin, B * H * W, Cin, planes.data_ptr(), nb, st), "split")
        ws = torch.empty(L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, k, k, 1, pad, dil), dtype=torch.uint8, device=DEV)
        Ho, Wo = E.out_size(H, k, 1, pad, dil), E.out_size(W, k, 1, pad, dil)
        ya, yb = torch.empty(B, Ho, Wo, Cout, device=DEV), torch.empty(B, Ho, Wo, Cout, device=DEV)
        _lib.check(L.pp_conv2d_fwd(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, k, k, 1, pad, dil, ya.data_ptr(), Cout, Cout,
                                   ws.data_ptr(), ws.numel(), st), "fwd")
        _lib.check(L.pp_conv2d_fwd_pre(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), None, k, k, 1, pad, dil, yb.data_ptr(), Cout, Cout,
                                       ws.data_ptr(), ws.numel(), planes.data_ptr(), st), "fwd_pre")
        assert torch.equal(ya, yb)


def test_bf16_split_is_exact():
    """hi + mid + lo == a exactly for normal fp32 values (the property the path rests on) - checked through a 1x1 convolution with an
    identity-like weight: y = x * 1.0 must reproduce x bit for bit in the split path."""
    L = _lib.lib()
    B, H, W, C = 2, 128, 128, 512
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(B, H, W, C, device=DEV, generator=gen) * torch.exp(4 * torch.randn(B, H, W, C, device=DEV, generator=gen))
    w = torch.eye(C, device=DEV).reshape(1, 1, C, C).contiguous()
    assert L.pp_conv2d_fwd_workspace_bytes(B, H, W, C, C, 1, 1, 1, 0, 1) > 0
    y = E.conv2d(E.Tape(False), E.Var(x), w, None, 1, 0, 1).t
    assert torch.equal(y, x)

