"""An EXPERIMENT kept in the test build (measured slower, profiles/r06_wgrad_fold.txt; the product reduces in a launch of its own):
the split-K reduction of the convolution weight gradients folded into the producing launch (conv_igemm.hip: wgrad_fold_tail; the autograd
backward of every dense nn.Conv2d, model.py:121): the last block of a tile to arrive sums the tile's slices in slice order.  Held here: the
gradient is bit-identical to partial sums + wgrad_reduce4_kernel (the switch brings those back), whichever block arrives last - repeated
launches, with a second stream busy beside them, give the same bits every time - for the fp32-MFMA tiles, the LDS-DMA tiles, the bf16x3
kernel, dead taps (dilation larger than the map) and ragged channel counts; and the arrival counters are left re-armed."""
import numpy as np
import pytest
import torch

from pixelpick_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FOLD_ON = 1024 | (1 << 25)           # pp_debug_set_wgrad_target: default target, the folded reduction (an experiment of the test build) ON


@pytest.fixture(autouse=True)
def _reset():
    yield
    _lib.lib().pp_debug_set_wgrad_target(0)


def _wgrad(x, dy, k, stride, pad, dil, Cin, Cout, stream=None):
    L = _lib.lib()
    B, H, W, _ = x.shape
    st = (stream or torch.cuda.current_stream()).cuda_stream
    nb = int(L.pp_conv2d_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    dw = torch.full((k, k, Cin, Cout), float("nan"), device=DEV)
    rc = L.pp_conv2d_bwd_weight(x.data_ptr(), Cin, B, H, W, Cin, dy.data_ptr(), Cout, Cout, k, k, stride, pad, dil, dw.data_ptr(), None,
                                ws.data_ptr(), nb, st)
    _lib.check(rc, "pp_conv2d_bwd_weight")
    return dw


# (B, H, W, Cin, Cout, k, stride, pad, dil)
CASES = [(4, 64, 128, 304, 256, 3, 1, 1, 1),      # SegmentHead conv1: bf16x3 weight gradient, ragged Cin
         (4, 64, 128, 256, 256, 3, 1, 1, 1),      # SegmentHead conv2
         (4, 16, 32, 320, 256, 3, 1, 18, 18),     # ASPP d = 18 on a 16 x 32 map: dead taps stay zero
         (4, 16, 32, 960, 160, 1, 1, 0, 1),       # MobileNetV2 project
         (4, 18, 34, 160, 960, 1, 1, 0, 1),       # MobileNetV2 expand
         (4, 32, 64, 256, 1024, 1, 1, 0, 1),      # ResNet50 Bottleneck
         (4, 32, 64, 512, 512, 3, 1, 2, 2),       # ResNet50 layer-4 3x3
         (2, 33, 31, 68, 100, 3, 2, 1, 1),        # ragged everything, stride 2
         (4, 64, 128, 64, 64, 1, 1, 0, 1)]        # 64 x 64 tiles


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_folded_reduction_is_bit_identical_and_reproducible(case):
    B, H, W, Cin, Cout, k, stride, pad, dil = case
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(Cin + 7 * Cout)
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    x = torch.randn(B, H, W, Cin, device=DEV, generator=gen)
    dy = torch.randn(B, Ho, Wo, Cout, device=DEV, generator=gen)
    L.pp_debug_set_wgrad_target(0)
    ref = _wgrad(x, dy, k, stride, pad, dil, Cin, Cout)
    L.pp_debug_set_wgrad_target(FOLD_ON)
    assert not torch.isnan(ref).any()
    side = torch.cuda.Stream()
    noise = torch.randn(64 << 20, device=DEV)
    for it in range(12):
        if it % 2:                                   # a bandwidth hog on a second queue: block arrival order changes
            with torch.cuda.stream(side):
                noise.mul_(1.0001)
        got = _wgrad(x, dy, k, stride, pad, dil, Cin, Cout)
        assert torch.equal(got, ref), f"launch {it} differs from partial sums + reduce launch"
    torch.cuda.synchronize()


def test_two_folded_launches_on_two_streams_do_not_share_counters():
    """Weight gradients of two layers in flight at once (the engine's main and weight-gradient queues): each launch takes its own range of
    the counter ring."""
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(5)
    a = (torch.randn(4, 64, 128, 256, device=DEV, generator=gen), torch.randn(4, 64, 128, 256, device=DEV, generator=gen), 3, 1, 1, 1, 256, 256)
    b = (torch.randn(4, 32, 64, 256, device=DEV, generator=gen), torch.randn(4, 32, 64, 1024, device=DEV, generator=gen), 1, 1, 0, 1, 256, 1024)
    L.pp_debug_set_wgrad_target(0)
    ra, rb = _wgrad(*a), _wgrad(*b)
    L.pp_debug_set_wgrad_target(FOLD_ON)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(8):
        with torch.cuda.stream(s1):
            ga = _wgrad(*a, stream=s1)
        with torch.cuda.stream(s2):
            gb = _wgrad(*b, stream=s2)
        torch.cuda.synchronize()
        assert torch.equal(ga, ra) and torch.equal(gb, rb)
