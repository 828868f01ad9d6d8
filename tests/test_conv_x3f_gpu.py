"""conv_x3f_kernel (csrc/conv_x3f.hip; an experiment kept in the TEST BUILD only - it measured slower than the kernels it was meant to
replace, profiles/r06_x3f_in_kernel_split.txt): the bf16x3 convolution whose ACTIVATION operand is read as fp32 and split into its three
bf16 planes inside the kernel - the ResNet50 Bottleneck 1x1 / 3x3 layers (resnet_models.py:58-94), the decoder / head convolutions
(decoders.py:25-77,107-114) and their backward-data (model.py:121) without an x3_split launch.

What has to hold: (1) it is the SAME arithmetic as conv_x3_kernel on pre-split planes - v_cvt_pk_bf16_f32 rounds to nearest even as
x3_split_kernel does and the MFMA order is the same, so the two paths agree bit for bit wherever both can run; (2) on the layers only
this path serves it is an fp32 convolution (error vs float64 = the fp32-MFMA kernels'); (3) every tile form, K tail, ragged edge,
short reduction, accumulate epilogue and the backward-data direction."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pixelpick_amd import _lib
from pixelpick_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

X3_EVERYTHING_CLASSIC = 1 | (7 << 9) | (3 << 12)        # pp_debug_set_x3: mid-size layers from 1 GFLOP / 32 tiles take conv_x3_kernel
X3F_ON = 1                                              # pp_debug_set_x3f bit 0: the experiment kernel on, with the default plan
X3F_LOOSE = 1 | (6 << 1) | (1 << 4) | (3 << 6)          # ... any work, from 64 tiles, K >= 16
X3F_OFF = 0


@pytest.fixture(autouse=True)
def _reset_knobs():
    yield
    L = _lib.lib()
    L.pp_debug_set_x3(1)
    L.pp_debug_set_x3f(0)


def _data(B, H, W, Cin, Cout, k, seed):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=gen) * torch.exp(torch.randn(B, H, W, Cin, device=DEV, generator=gen))
    w = torch.randn(k, k, Cin, Cout, device=DEV, generator=gen) / np.sqrt(Cin * k * k)
    return x, w


def _fwd(x, w, stride, pad, dil, planes=None, bias=None):
    L = _lib.lib()
    B, H, W, Cin = x.shape
    k, _, _, Cout = w.shape
    Ho, Wo = E.out_size(H, k, stride, pad, dil), E.out_size(W, k, stride, pad, dil)
    y = torch.empty(B, Ho, Wo, Cout, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = int(L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    bp = bias.data_ptr() if bias is not None else None
    if planes is None:
        rc = L.pp_conv2d_fwd(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), bp, k, k, stride, pad, dil, y.data_ptr(), Cout, Cout,
                             ws.data_ptr() if nb else None, nb, st)
    else:
        rc = L.pp_conv2d_fwd_pre(x.data_ptr(), Cin, B, H, W, Cin, w.data_ptr(), bp, k, k, stride, pad, dil, y.data_ptr(), Cout, Cout,
                                 ws.data_ptr() if nb else None, nb, planes.data_ptr(), st)
    _lib.check(rc, "conv fwd")
    return y


def _bwd(dy, w, H, W, stride, pad, dil, planes=None, into=None):
    L = _lib.lib()
    B, Ho, Wo, Cout = dy.shape
    k, _, Cin, _ = w.shape
    dx = into if into is not None else torch.empty(B, H, W, Cin, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=DEV)
    args = (dy.data_ptr(), Cout, B, Ho, Wo, Cout, w.data_ptr(), k, k, stride, pad, dil, dx.data_ptr(), Cin, H, W, Cin, 1 if into is not None else 0,
            ws.data_ptr() if nb else None, nb)
    rc = L.pp_conv2d_bwd_data(*args, st) if planes is None else L.pp_conv2d_bwd_data_pre(*args, planes.data_ptr(), st)
    _lib.check(rc, "conv bwd_data")
    return dx


def _split(t):
    L = _lib.lib()
    rows, C = t.numel() // t.shape[-1], t.shape[-1]
    nb = int(L.pp_x3_planes_bytes(rows, C))
    planes = torch.empty(nb, dtype=torch.uint8, device=DEV)
    _lib.check(L.pp_x3_split(t.data_ptr(), C, rows, C, planes.data_ptr(), nb, torch.cuda.current_stream().cuda_stream), "split")
    return planes


# (B, H, W, Cin, Cout, k, stride, pad, dil) - the ResNet50 / head layers at the BASELINE shapes and the corner cases of the kernel
SHAPES = [
    (4, 32, 64, 256, 1024, 1, 1, 0, 1),      # Bottleneck expand at 8192 rows: 256 x 128 tiles, K = 16 steps
    (4, 32, 64, 1024, 256, 1, 1, 0, 1),      # Bottleneck reduce: 128 x 64 tiles (a block per CU)
    (4, 32, 64, 2048, 512, 1, 1, 0, 1),      # 128 x 128 tiles, 128 K steps
    (4, 32, 64, 256, 256, 3, 1, 1, 1),       # Bottleneck 3x3: nine taps, border masks
    (4, 32, 64, 512, 512, 3, 1, 2, 2),       # dilated layer-3 3x3 (a classic layer)
    (8, 64, 128, 64, 64, 3, 1, 1, 1),        # layer-1 3x3 at 65536 rows: 256 x 64 tiles
    (4, 64, 128, 304, 256, 3, 1, 1, 1),      # SegmentHead conv1: K = 304 (19 chunks), classic
    (3, 30, 33, 300, 192, 1, 1, 0, 1),       # ragged rows (2970), Cin = 300 (half-chunk and quad masks), 192 = 128 + 64 columns
    (4, 32, 64, 72, 128, 3, 1, 1, 1),        # Cin = 72: chunks of 16 with a half chunk at the end
    (4, 64, 64, 256, 128, 1, 2, 0, 1),       # the downsample 1x1 with stride 2 (resnet_models.py:79-83)
]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_in_kernel_split_equals_the_pre_split_kernel_bit_for_bit(shape):
    B, H, W, Cin, Cout, k, stride, pad, dil = shape
    L = _lib.lib()
    L.pp_debug_set_x3(X3_EVERYTHING_CLASSIC)
    L.pp_debug_set_x3f(X3F_LOOSE)
    x, w = _data(B, H, W, Cin, Cout, k, Cin + Cout)
    bias = torch.randn(Cout, device=DEV)
    if not L.pp_conv2d_x3_planes_bytes(0, B, H, W, Cin, Cout, k, k, stride, pad, dil):
        pytest.skip("conv_x3_kernel does not take this shape even with the loosest plan (split-K plan): covered by the fp64 test")
    y_f = _fwd(x, w, stride, pad, dil, None, bias)                  # no planes: conv_x3f_kernel
    y_c = _fwd(x, w, stride, pad, dil, _split(x), bias)             # the caller's planes: conv_x3_kernel
    assert torch.equal(y_f, y_c)
    L.pp_debug_set_x3f(X3F_OFF)                                      # in-kernel split off: x3_split_kernel + conv_x3_kernel inside the call
    assert torch.equal(_fwd(x, w, stride, pad, dil, None, bias), y_c)
    L.pp_debug_set_x3f(X3F_LOOSE)
    if stride == 1 and L.pp_conv2d_x3_planes_bytes(1, B, H, W, Cin, Cout, k, k, stride, pad, dil):
        dy = torch.randn_like(y_f)
        dx_f = _bwd(dy, w, H, W, stride, pad, dil)
        dx_c = _bwd(dy, w, H, W, stride, pad, dil, _split(dy))
        assert torch.equal(dx_f, dx_c)
        base = torch.randn(B, H, W, Cin, device=DEV)                # accumulate epilogue: dx += result
        a_f = _bwd(dy, w, H, W, stride, pad, dil, None, base.clone())
        a_c = _bwd(dy, w, H, W, stride, pad, dil, _split(dy), base.clone())
        assert torch.equal(a_f, a_c)


@pytest.mark.parametrize("shape", SHAPES, ids=[str(s) for s in SHAPES])
def test_in_kernel_split_is_an_fp32_convolution(shape):
    """With the product's own plan: whatever kernel the library picks with the in-kernel split offered must be as close to float64 as
    the fp32-MFMA kernels are (switches off), forward and backward-data."""
    B, H, W, Cin, Cout, k, stride, pad, dil = shape
    L = _lib.lib()
    L.pp_debug_set_x3f(X3F_ON)
    x, w = _data(B, H, W, Cin, Cout, k, 3 * Cin + Cout)
    y = _fwd(x, w, stride, pad, dil)
    dy = torch.randn_like(y)
    dx = _bwd(dy, w, H, W, stride, pad, dil)
    L.pp_debug_set_x3f(X3F_OFF)
    L.pp_debug_set_x3(0)
    y1 = _fwd(x, w, stride, pad, dil)
    dx1 = _bwd(dy, w, H, W, stride, pad, dil)
    xd, wd = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(True), w.double().permute(3, 2, 0, 1).cpu()
    ref = F.conv2d(xd, wd, None, stride, pad, dil)
    ref.backward(dy.double().permute(0, 3, 1, 2).cpu())
    ry, rdx = ref.detach().permute(0, 2, 3, 1), xd.grad.permute(0, 2, 3, 1)

    def err(a, r):
        return ((a.double().cpu() - r).abs().max() / r.abs().max()).item(), ((a.double().cpu() - r).norm() / r.norm()).item()
    e, e1, d, d1 = err(y, ry), err(y1, ry), err(dx, rdx), err(dx1, rdx)
    print(f"\n[x3f] {shape}: fwd max/l2 {e[0]:.2e}/{e[1]:.2e} (fp32-MFMA {e1[0]:.2e}/{e1[1]:.2e}) bwd {d[0]:.2e}/{d[1]:.2e} ({d1[0]:.2e}/{d1[1]:.2e})")
    assert e[1] <= max(2.0 * e1[1], 3e-7) and d[1] <= max(2.0 * d1[1], 3e-7) and e[0] <= 5e-6 and d[0] <= 5e-6


@pytest.mark.parametrize("Cin", [16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 208])
def test_short_reductions_cover_prologue_and_tail(Cin):
    """1 .. 13 K steps: fewer steps than the ring holds, the first checked pair, a steady loop of 0 / 1 / 2 pairs, odd and even tails -
    on 128 x 128 tiles (ring of three) and, at 32768 rows, 256 x 128 tiles (ring of four)."""
    L = _lib.lib()
    L.pp_debug_set_x3f(X3F_LOOSE)
    for B, H, W, Cout in ((2, 32, 64, 512), (4, 64, 128, 256)):
        x, w = _data(B, H, W, Cin, Cout, 1, Cin)
        assert L.pp_conv2d_x3_planes_bytes(3, B, H, W, Cin, Cout, 1, 1, 1, 0, 1) > 0, "the loose plan should hand this layer to conv_x3f_kernel"
        y = _fwd(x, w, 1, 0, 1)
        ref = (x.double().reshape(-1, Cin) @ w.double().reshape(Cin, Cout)).reshape(y.shape)
        rel = ((y.double() - ref).norm() / ref.norm()).item()
        assert rel <= 3e-7, (Cin, B, rel)


def test_engine_keeps_weight_planes_for_layers_that_split_in_the_kernel():
    """A Bottleneck 1x1 in a tape-enabled step: no activation planes (pp_conv2d_x3_planes_bytes(0) == 0: nothing is split ahead), the
    step's pre-split WEIGHT planes are used (which = 3 / 4), results equal the plain entry points bit for bit."""
    L = _lib.lib()
    L.pp_debug_set_x3f(X3F_ON)
    B, H, W, Cin, Cout = 4, 32, 64, 1024, 256
    assert L.pp_conv2d_x3_planes_bytes(0, B, H, W, Cin, Cout, 1, 1, 1, 0, 1) == 0
    assert L.pp_conv2d_x3_planes_bytes(3, B, H, W, Cin, Cout, 1, 1, 1, 0, 1) == L.pp_x3_weight_planes_bytes(1, Cin, Cout, 1)
    assert L.pp_conv2d_x3_planes_bytes(4, B, H, W, Cin, Cout, 1, 1, 1, 0, 1) == L.pp_x3_weight_planes_bytes(1, Cin, Cout, 0)
    x, w = _data(B, H, W, Cin, Cout, 1, 99)
    w.requires_grad_(True)
    dy = torch.randn(B, H, W, Cout, device=DEV)
    bias = torch.zeros(Cout, device=DEV)     # (a bias keeps the layer off the deferred conv + BatchNorm path: the plain forward is the one under test)
    outs = []
    for _ in range(2):                       # second round: the planes registered in the first are split at begin_step and used
        E.begin_step()
        tape = E.Tape()
        xv = E.Var(x)
        yv = E.conv2d(tape, xv, w, bias, 1, 0, 1)
        y = yv.t.clone()
        tape.backward(yv, dy)
        torch.cuda.synchronize()
        outs.append((y, xv.grad.clone()))
        E.end_step()
    assert id(w) in E._X3_WPL and set(E._X3_WPL[id(w)]["planes"]) == {0, 1}
    y_plain = _fwd(x, w.detach(), 1, 0, 1)
    dx_plain = _bwd(dy, w.detach(), H, W, 1, 0, 1)
    for y, dx in outs:
        assert torch.equal(y, y_plain) and torch.equal(dx, dx_plain)


def test_idle_weights_are_not_split_again_by_other_steps():
    """engine._X3_WPL is process-wide: begin_step() re-splits only the weights the previous step used.  A layer that ran in two steps
    and then sits idle (another trainer's model, a stand-alone conv2d call) keeps stale planes - its entry's epoch stops advancing -
    while a layer that keeps running is split at every begin_step(); when the idle layer runs again it splits inline (same bits) and is
    back in the prefetch from the step after."""
    L = _lib.lib()
    L.pp_debug_set_x3f(X3F_ON)
    B, H, W, Cin, Cout = 4, 32, 64, 1024, 256
    xa, wa = _data(B, H, W, Cin, Cout, 1, 5)
    xb, wb = _data(B, H, W, Cin, Cout, 1, 6)
    wa.requires_grad_(True)
    wb.requires_grad_(True)
    bias = torch.zeros(Cout, device=DEV)

    def step(pairs):
        E.begin_step()
        tape = E.Tape()
        ys = [E.conv2d(tape, E.Var(x), w, bias, 1, 0, 1).t.clone() for x, w in pairs]
        torch.cuda.synchronize()
        E.end_step()
        return ys

    for _ in range(2):
        step([(xa, wa), (xb, wb)])
    ea, eb = E._X3_WPL[id(wa)], E._X3_WPL[id(wb)]
    assert ea["epoch"] == eb["epoch"] == E._STEP_EPOCH[0] - 1            # both split at the last begin_step()
    for _ in range(3):
        ya = step([(xa, wa)])[0]
    assert ea["epoch"] == E._STEP_EPOCH[0] - 1                            # the running layer: split at every begin_step()
    assert eb["epoch"] < E._STEP_EPOCH[0] - 4                             # the idle one: left alone
    ya2, yb = step([(xa, wa), (xb, wb)])                                   # idle layer runs again: inline split, same bits
    assert torch.equal(ya, ya2) and torch.equal(yb, _fwd(xb, wb.detach(), 1, 0, 1))
    step([(xa, wa), (xb, wb)])
    assert eb["epoch"] == E._STEP_EPOCH[0] - 1                            # and is prefetched again

