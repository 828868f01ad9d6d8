"""gemm_pw_kernel (csrc/gemm_pw.hip): the pointwise convolutions of the ResNet50 models (resnet_models.py:58-94 Bottleneck conv1 / conv3,
decoders.py:25-77, aspp.py:49,73-75) as a plain row-major fp32-MFMA GEMM with one large tile per CU.  Held here: every tile form gives the
SAME bits (one k order per accumulator), the result is an fp32 convolution (vs float64 and vs the implicit-GEMM kernels it replaces),
ragged rows / columns / reductions, K tails of the 32-deep step, bias, channel-slice inputs and outputs (pixel strides)."""
import numpy as np
import pytest
import torch

from pixelpick_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _reset():
    yield
    _lib.lib().pp_debug_set_gemm_pw(1)


def _conv1x1(x, w, y, B, H, W, Cin, Cout, ldx, ldy, bias=None):
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nb = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, 1, 1, 1, 0, 1))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    rc = L.pp_conv2d_fwd(x.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), bias.data_ptr() if bias is not None else None, 1, 1, 1, 0, 1,
                         y.data_ptr(), ldy, Cout, ws.data_ptr() if nb else None, nb, st)
    _lib.check(rc, "pp_conv2d_fwd")


# (B, H, W, Cin, Cout): the graded Bottleneck shapes and the edges of the kernel
SHAPES = [(4, 32, 64, 256, 1024), (4, 32, 64, 1024, 256), (4, 32, 64, 2048, 512), (4, 64, 128, 64, 256), (4, 64, 128, 256, 64),
          (4, 32, 64, 512, 2048), (3, 37, 41, 200, 136), (2, 50, 50, 72, 68), (1, 70, 61, 1000, 332)]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_every_tile_form_gives_the_same_fp32_convolution(shape):
    B, H, W, Cin, Cout = shape
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(Cin + Cout)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=gen) * torch.exp(torch.randn(B, H, W, Cin, device=DEV, generator=gen))
    w = torch.randn(1, 1, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin)
    bias = torch.randn(Cout, device=DEV, generator=gen)
    ref = (x.double().reshape(-1, Cin) @ w.double().reshape(Cin, Cout) + bias.double()).reshape(B, H, W, Cout)
    outs = []
    for form in range(6):
        L.pp_debug_set_gemm_pw((2 + form) | (1 << 4))            # force the form, from one row on
        y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        _conv1x1(x, w, y, B, H, W, Cin, Cout, Cin, Cout, bias)
        outs.append(y)
    for form in range(1, 6):
        assert torch.equal(outs[0], outs[form]), f"form {form} differs from form 0"
    L.pp_debug_set_gemm_pw(0)                                     # the kernels it replaces
    y_old = torch.empty(B, H, W, Cout, device=DEV)
    _conv1x1(x, w, y_old, B, H, W, Cin, Cout, Cin, Cout, bias)
    e_new = ((outs[0].double() - ref).norm() / ref.norm()).item()
    e_old = ((y_old.double() - ref).norm() / ref.norm()).item()
    print(f"\n[gemm_pw] {shape}: rel-l2 vs fp64 {e_new:.2e} (implicit-GEMM kernels {e_old:.2e})")
    assert e_new <= max(2.0 * e_old, 3e-7)
    assert ((outs[0].double() - ref).abs().max() / ref.abs().max()).item() <= 5e-6


def test_channel_slices_and_the_planner_rule():
    """Input read from a channel slice of a wider tensor, output written into a slice of a wider tensor (the reference's torch.cat
    operands, aspp.py:73, decoders.py:77): pixel strides other than the channel counts; nothing outside the slice is touched.  With the
    rule on (default) the same bits as every forced form."""
    L = _lib.lib()
    B, H, W, Cin, Cout, ldx, ldy = 2, 64, 64, 192, 320, 256, 512
    gen = torch.Generator(device=DEV).manual_seed(7)
    xw = torch.randn(B, H, W, ldx, device=DEV, generator=gen)
    w = torch.randn(1, 1, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin)
    yw = torch.full((B, H, W, ldy), 7.0, device=DEV)
    x = xw[..., 32:32 + Cin]
    y = yw[..., 64:64 + Cout]
    _conv1x1(x, w, y, B, H, W, Cin, Cout, ldx, ldy)
    ref = (x.double().reshape(-1, Cin) @ w.double().reshape(Cin, Cout)).reshape(B, H, W, Cout)
    assert ((y.double() - ref).norm() / ref.norm()).item() <= 3e-7
    assert (yw[..., :64] == 7.0).all() and (yw[..., 64 + Cout:] == 7.0).all()
    L.pp_debug_set_gemm_pw(2 | (1 << 4))
    y2w = torch.full((B, H, W, ldy), 7.0, device=DEV)
    _conv1x1(x, w, y2w[..., 64:64 + Cout], B, H, W, Cin, Cout, ldx, ldy)
    assert torch.equal(y2w, yw)


@pytest.mark.parametrize("K", [64, 68, 96, 100, 128, 132, 160, 224])
def test_reduction_tails(K):
    """2 .. 7 K steps of 32 with and without a ragged last step: prologue with fewer steps than the ring holds, steady loop of 0 .. 4 steps, tail."""
    L = _lib.lib()
    L.pp_debug_set_gemm_pw(2 | (1 << 4))
    B, H, W, Cout = 1, 64, 80, 256
    gen = torch.Generator(device=DEV).manual_seed(K)
    x = torch.randn(B, H, W, K, device=DEV, generator=gen)
    w = torch.randn(1, 1, K, Cout, device=DEV, generator=gen)
    y = torch.empty(B, H, W, Cout, device=DEV)
    _conv1x1(x, w, y, B, H, W, K, Cout, K, Cout)
    ref = (x.double().reshape(-1, K) @ w.double().reshape(K, Cout)).reshape(B, H, W, Cout)
    assert ((y.double() - ref).norm() / ref.norm()).item() <= 3e-7


def _bwd1x1(dy, w, dx, B, H, W, Cin, Cout, accumulate=False):
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nb = int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, 1, 1, 1, 0, 1))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    rc = L.pp_conv2d_bwd_data(dy.data_ptr(), Cout, B, H, W, Cout, w.data_ptr(), 1, 1, 1, 0, 1, dx.data_ptr(), Cin, H, W, Cin, 1 if accumulate else 0,
                              ws.data_ptr() if nb else None, nb, st)
    _lib.check(rc, "pp_conv2d_bwd_data")


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_backward_data_reads_the_weight_as_the_transposed_operand(shape):
    """dX = dY x W^T (model.py:121 through a 1x1 convolution): the same kernel with the B tile taken from W's rows; every tile form the
    same bits, fp32 accuracy, and the accumulate epilogue (dx += ..., the residual branch's gradient already in place)."""
    B, H, W, Cin, Cout = shape
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(3 * Cin + Cout)
    dy = torch.randn(B, H, W, Cout, device=DEV, generator=gen) * torch.exp(torch.randn(B, H, W, Cout, device=DEV, generator=gen))
    w = torch.randn(1, 1, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cout)
    ref = (dy.double().reshape(-1, Cout) @ w.double().reshape(Cin, Cout).t()).reshape(B, H, W, Cin)
    outs = []
    for form in range(6):
        L.pp_debug_set_gemm_pw((2 + form) | (1 << 4))
        dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
        _bwd1x1(dy, w, dx, B, H, W, Cin, Cout)
        outs.append(dx)
    for form in range(1, 6):
        assert torch.equal(outs[0], outs[form]), f"form {form} differs from form 0"
    base = torch.randn(B, H, W, Cin, device=DEV, generator=gen)
    acc = base.clone()
    _bwd1x1(dy, w, acc, B, H, W, Cin, Cout, accumulate=True)
    assert torch.equal(acc, outs[5] + base)
    L.pp_debug_set_gemm_pw(0)
    dx_old = torch.empty(B, H, W, Cin, device=DEV)
    _bwd1x1(dy, w, dx_old, B, H, W, Cin, Cout)
    e_new = ((outs[0].double() - ref).norm() / ref.norm()).item()
    e_old = ((dx_old.double() - ref).norm() / ref.norm()).item()
    assert e_new <= max(2.0 * e_old, 3e-7) and ((outs[0].double() - ref).abs().max() / ref.abs().max()).item() <= 5e-6
