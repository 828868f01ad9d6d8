"""GPU parity tests of the network-layer HIP kernels (through the C ABI / engine tape) against plain
PyTorch fp32 CPU references of the same ops (the reference's torch.nn calls).  Floating-point kernels:
tolerance 1e-4 relative to the tensor scale (north_star: logits/grads within 1e-3)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from pixelpick_amd import engine as E

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):   # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):   # NHWC gpu -> NCHW cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def hwio(w):   # OIHW -> HWIO gpu
    return w.permute(2, 3, 1, 0).contiguous().to(DEV)


def oihw(w):
    return w.permute(3, 2, 0, 1).contiguous().cpu()


def close(a, b, tol=1e-4, what=""):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def gparam(t):
    t = t.detach().clone().to(DEV)
    t.requires_grad_(True)
    return t


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dil, bias
    (2, 16, 32, 320, 256, 1, 1, 0, 1, False),     # ASPP 1x1
    (2, 16, 32, 320, 256, 3, 1, 6, 6, False),     # atrous d=6
    (2, 16, 32, 64, 96, 3, 1, 18, 18, False),     # atrous d=18: only the centre row of taps is live
    (1, 24, 20, 304, 256, 3, 1, 1, 1, False),     # SegmentHead 3x3, ragged M
    (2, 17, 19, 256, 19, 1, 1, 0, 1, True),       # classifier (bias, Cout=19)
    (2, 18, 34, 16, 96, 1, 1, 0, 1, False),       # MNv2 expand on the padded map
    (2, 9, 11, 24, 144, 1, 1, 0, 1, False),       # Cin=24 (not a multiple of 16)
    (2, 10, 12, 960, 160, 1, 1, 0, 1, False),     # MNv2 project
    (2, 33, 47, 3, 32, 3, 2, 1, 1, False),        # stem 3x3 s2, Cin=3
    (1, 40, 36, 3, 64, 7, 2, 3, 1, False),        # ResNet stem 7x7 s2
    (2, 12, 12, 128, 128, 3, 1, 1, 1, True),      # FPN UpsampleBlock conv (bias)
    # >= 32768 output pixels with 3..32 channels on one side: the narrow-layer weight-gradient kernels
    (2, 256, 264, 3, 32, 3, 2, 1, 1, False),      # MNv2 stem
    (2, 128, 160, 32, 16, 1, 1, 0, 1, False),     # MNv2 block 1 project
    (2, 128, 130, 16, 96, 1, 1, 1, 1, False),     # expand with the folded fixed_padding (1x1, padding 1)
    (2, 128, 136, 96, 24, 1, 1, 0, 1, False),     # narrow output
    (2, 128, 132, 144, 32, 1, 1, 0, 1, False),
    (1, 128, 160, 3, 64, 7, 2, 3, 1, False),      # ResNet stem below 32 K pixels (stays on the MFMA path)
    (2, 256, 264, 3, 64, 7, 2, 3, 1, False),      # ResNet stem, 33792 pixels: seven tap rows of the narrow-input kernel over grid.y
    (3, 96, 128, 256, 19, 1, 1, 0, 1, True),      # classifier (bias), 36864 pixels
    (1, 64, 128, 304, 256, 3, 1, 1, 1, False),    # SegmentHead at Cityscapes-quarter size (128x128 tiles)
    (2, 23, 30, 1280, 256, 1, 1, 0, 1, False),    # ASPP fuse at CamVid size (M = 1380)
    # few-row pointwise layers: K >= 256 -> conv1x1_ksplit_dma_kernel (forward, and the backward-data of the mirrored shape), the
    # others -> the tiled kernels; ragged rows / channels, the folded padding, bias
    (4, 16, 32, 160, 960, 1, 1, 1, 1, False),     # MNv2 expand 160->960 on the padded 18x34 map
    (4, 16, 32, 960, 160, 1, 1, 0, 1, False),     # MNv2 project 960->160
    (4, 16, 32, 960, 320, 1, 1, 0, 1, False),
    (4, 16, 32, 1280, 256, 1, 1, 0, 1, False),    # ASPP fuse
    (3, 23, 30, 576, 96, 1, 1, 0, 1, True),       # ragged rows (2070), bias
    (4, 16, 32, 64, 384, 1, 1, 1, 1, False),
    (2, 15, 17, 384, 68, 1, 1, 0, 1, False),      # Cout not a multiple of 32
    (2, 23, 30, 320, 256, 3, 1, 12, 12, False),   # atrous d=12 at CamVid size
    (2, 17, 22, 160, 960, 1, 1, 0, 1, False),
    # deep-K few-row pointwise layers -> conv1x1_ksplit_dma_kernel (K >= 768): ragged K (not a multiple of 16), ragged rows, a
    # partial 32-column tile, bias; the mirrored shape sends the backward-data through it
    (2, 9, 13, 776, 68, 1, 1, 0, 1, True),
    (2, 9, 13, 68, 776, 1, 1, 0, 1, False),
    (3, 11, 7, 1284, 36, 1, 1, 1, 1, True),       # folded padding: border rows read zeros
    (4, 64, 128, 304, 256, 3, 1, 1, 1, False),    # SegmentHead conv1 at bench scale: the 128x128 large-tile kernels
    (3, 50, 100, 256, 304, 3, 1, 1, 1, True),     # large tile, ragged M (15000 rows), Cout=304 (partial N tile)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_fwd_bwd(case):
    B, H, W, Cin, Cout, k, stride, pad, dil, has_bias = case
    torch.manual_seed(hash(case) % 2**31)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout) if has_bias else None
    xr = x.clone().requires_grad_(stride == 1)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn_like(yr)
    yr.backward(dy)

    tape = E.Tape()
    xv = E.Var(nhwc(x), needs_grad=stride == 1)
    wg = gparam(hwio(w))
    bg = gparam(b) if has_bias else None
    yv = E.conv2d(tape, xv, wg, bg, stride, pad, dil)
    close(nchw(yv.t), yr.detach(), what="conv fwd")
    tape.backward(yv, nhwc(dy))
    close(oihw(tape.param_grads[id(wg)]), wr.grad, what="conv dW")
    if has_bias:
        close(tape.param_grads[id(bg)].cpu(), br.grad, what="conv dbias")
    if stride == 1:
        close(nchw(xv.grad), xr.grad, what="conv dX")


@pytest.mark.parametrize("case", [(4, 256, 512), (2, 256, 256), (3, 200, 260), (1, 512, 258)], ids=str)
def test_mobilenet_stem_forward(case):
    """mobilenet_v2.py:7-12 Conv2d(3, 32, 3, stride 2, padding 1, bias=False) on an even-sized image: conv_stem3x3s2_fwd_kernel (nine
    contiguous floats per tap row, weights as scalar operands) against torch and the MFMA path (pp_debug_set_conv_rows(1));
    bit-reproducible."""
    B, H, W = case
    gen = torch.Generator().manual_seed(B + H + W)
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(32, 3, 3, 3, generator=gen) / np.sqrt(27)
    yr = F.conv2d(x, w, stride=2, padding=1)
    L = _lib_mod().lib()
    def run(off):
        L.pp_debug_set_conv_rows(off)
        try:
            yv = E.conv2d(E.Tape(), E.Var(nhwc(x), needs_grad=False), gparam(hwio(w)), None, 2, 1, 1)
            return nchw(yv.t)
        finally:
            L.pp_debug_set_conv_rows(0)
    a, a2, b = run(0), run(0), run(1)
    assert torch.equal(a, a2)
    close(a, yr, what="stem forward")
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


@pytest.mark.parametrize("case", [(4, 130, 258, 96, 0), (4, 66, 130, 144, 0), (2, 34, 66, 192, 0), (2, 33, 47, 32, 1), (1, 18, 34, 576, 0), (3, 21, 20, 8, 2)],
                         ids=str)
def test_depthwise_stride2_backward_data_is_bit_identical_to_the_generic_kernel(case):
    """mobilenet_v2.py:50 stride-2 depthwise layers: dwconv_s2_bwd_data_kernel (one thread = a 2 x 2 block of dx pixels, four dy pixels,
    no branches around the loads) adds every pixel's taps in the generic kernel's order - bit-identical to it
    (pp_debug_set_dw_variant bit 20) - and equals torch autograd."""
    B, H, W, C, pad = case
    gen = torch.Generator().manual_seed(H + W + C)
    x = torch.randn(B, C, H, W, generator=gen)
    wd = torch.randn(C, 1, 3, 3, generator=gen) / 3
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(xr, wd, stride=2, padding=pad, groups=C)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    L = _lib_mod().lib()
    def run(variant):
        L.pp_debug_set_dw_variant(variant)
        try:
            tape = E.Tape()
            xv = E.Var(nhwc(x))
            yv = E.dwconv3x3(tape, xv, gparam(wd[:, 0].permute(1, 2, 0).contiguous()), 2, pad, 1)
            tape.backward(yv, nhwc(dy))
            torch.cuda.synchronize()
            return xv.grad.clone()
        finally:
            L.pp_debug_set_dw_variant(0)
    a, b = run(0), run(1 << 20)
    assert torch.equal(a, b)
    close(nchw(a), xr.grad, what="depthwise stride-2 dx")


ROWS_BWD = [(4, 128, 256, 16, 96, 1), (4, 64, 128, 24, 144, 1), (2, 100, 83, 16, 96, 0), (1, 129, 131, 24, 144, 1), (3, 80, 90, 32, 192, 0),
            (2, 96, 96, 32, 64, 1)]


@pytest.mark.parametrize("case", ROWS_BWD, ids=[str(c) for c in ROWS_BWD])
def test_pointwise_backward_data_of_the_narrow_layers(case):
    """mobilenet_v2.py:48 expand convolutions at 1/2 and 1/4 resolution (16 -> 96, 24 -> 144; fixed padding folded in): their
    backward-data is conv1x1_rows_kernel<.., true> (whole rows through LDS, VALU).  Against torch autograd, against the MFMA path
    (pp_debug_set_conv_rows(1)) to fp32 rounding, with a gradient already present (the residual branch: accumulate), ragged row
    counts; bit-reproducible."""
    B, H, W, Cin, Cout, pad = case
    gen = torch.Generator().manual_seed(H + W + Cin)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    dy = torch.randn(B, Cout, H + 2 * pad, W + 2 * pad, generator=gen)
    dres = torch.randn(B, Cin, H, W, generator=gen)
    xr = x.clone().requires_grad_(True)
    (F.conv2d(xr, w, padding=pad) * dy).sum().backward()
    L = _lib_mod().lib()
    def run(variant, with_res):
        L.pp_debug_set_conv_rows(variant)
        try:
            tape = E.Tape()
            xv = E.Var(nhwc(x))
            yv = E.conv2d(tape, xv, gparam(hwio(w)), None, 1, pad, 1)
            if with_res:
                g0 = nhwc(dres).clone()
                g0._pp_owned = True
                xv.grad = g0                    # what the residual branch left: the convolution adds into it
            tape.backward(yv, nhwc(dy))
            torch.cuda.synchronize()
            return nchw(xv.grad)
        finally:
            L.pp_debug_set_conv_rows(0)
    a, a2, b = run(0, False), run(0, False), run(1, False)
    assert torch.equal(a, a2)
    close(a, xr.grad, what="dx of the narrow pointwise layer")
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    c = run(0, True)
    close(c, xr.grad + dres, what="dx added to the residual branch's gradient")


@pytest.mark.parametrize("M,Cin,Cout,nflag,bias", [(32768, 256, 19, 80, False), (524288, 128, 19, 80, True), (5000, 48, 7, 5000, True),
                                                  (1537, 128, 32, 1, False), (4096, 256, 21, 0, True)])
def test_sparse_pointwise_weight_gradient(M, Cin, Cout, nflag, bias):
    """pp_conv1x1_bwd_weight_sparse: dw = sum over the flagged rows of x[r]^T dy[r] (+ db) - the classifier behind model.py:113-119's
    sparse labels (decoders.py:64 at full resolution for FPNSeg, :120 for DeepLab).  Against a float64 sum of the same rows; bit-identical
    run to run (fixed partition and order); every row flagged and no row flagged are ordinary cases."""
    L = _lib_mod().lib()
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV).manual_seed(M + Cin)
    x = torch.randn((M, Cin), device=DEV, generator=gen)
    dy = torch.zeros((M, Cout), device=DEV)
    rows = torch.randperm(M, device=DEV, generator=gen)[:nflag]
    dy[rows] = torch.randn((nflag, Cout), device=DEV, generator=gen) + 0.1
    flags = torch.empty(M, dtype=torch.uint8, device=DEV)
    _lib_mod().check(L.pp_row_flags(dy.data_ptr(), Cout, M, Cout, flags.data_ptr(), st), "flags")
    assert int(flags.sum().item()) == nflag
    nws = int(L.pp_conv1x1_bwd_weight_sparse_workspace_bytes(M, Cin, Cout))
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        dw = torch.full((Cin, Cout), 7.0, device=DEV)
        db = torch.full((Cout,), 7.0, device=DEV) if bias else None
        _lib_mod().check(L.pp_conv1x1_bwd_weight_sparse(x.data_ptr(), Cin, M, Cin, dy.data_ptr(), Cout, Cout, flags.data_ptr(), dw.data_ptr(),
                                                       db.data_ptr() if bias else None, ws.data_ptr(), nws, st), "sparse wgrad")
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and (not bias or torch.equal(outs[0][1], outs[1][1]))
    ref = x[rows].double().t() @ dy[rows].double()
    scale = ref.abs().max().item() + 1e-6
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-6 * scale * max(1.0, nflag ** 0.5)
    if bias:
        rb = dy[rows].double().sum(0)
        assert (outs[0][1].double() - rb).abs().max().item() <= 2e-6 * (rb.abs().max().item() + 1e-6) * max(1.0, nflag ** 0.5)
    assert L.pp_conv1x1_bwd_weight_sparse_workspace_bytes(100, 512, 32) == 0         # 16416 accumulators: outside the kernel's range


@pytest.mark.parametrize("shape,act", [((4, 64, 128, 256), 1), ((4, 64, 128, 256), 0), ((2, 100, 90, 48), 2), ((2, 33, 47, 64), 1)])
def test_sparse_loss_gradient_rows(shape, act, monkeypatch):
    """model.py:113-119: 20 labelled pixels per image -> the loss gradient is zero in all but a few hundred low-resolution rows.
    pp_row_flags marks them; the classifier's backward-data (pp_conv1x1_bwd_data_sparse) writes zeros elsewhere and equals the dense
    backward-data to fp32 rounding; the BatchNorm backward told about the flags (pp_bn_bwd_fused_sparse) returns dx / dgamma / dbeta
    BIT-EQUAL to the dense kernel on the same gradient (the skipped terms are exact zeros) and equals torch autograd."""
    B, H, W, C = shape
    ncls = 19
    gen = torch.Generator().manual_seed(C + act)
    x = torch.randn(B, C, H, W, generator=gen) * 2 + 0.3
    wc = torch.randn(ncls, C, 1, 1, generator=gen) / np.sqrt(C)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.3
    rows = torch.randperm(B * H * W, generator=gen)[:300]
    dl_rows = torch.zeros(B * H * W, ncls)
    dl_rows[rows] = torch.randn(300, ncls, generator=gen)
    dlog = dl_rows.reshape(B, H, W, ncls).permute(0, 3, 1, 2).contiguous()
    def run(sparse):
        monkeypatch.setattr(E, "_SPARSE_ROWS", sparse)
        tape = E.Tape()
        xv = E.Var(nhwc(x))
        gg, bg, wg = gparam(gamma), gparam(beta), gparam(hwio(wc))
        y = E.batch_norm_act(tape, xv, gg, bg, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, act)
        logits = E.conv2d(tape, y, wg, None, 1, 0, 1)
        dl = nhwc(dlog)
        if sparse:
            flags = torch.empty(B * H * W, dtype=torch.uint8, device=DEV)
            _lib_mod().check(_lib_mod().lib().pp_row_flags(dl.data_ptr(), ncls, B * H * W, ncls, flags.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "flags")
            assert int(flags.sum().item()) == 300
            dl._pp_rowflags = flags
        tape.backward(logits, dl)
        torch.cuda.synchronize()
        return [xv.grad.clone(), tape.param_grads[id(gg)].clone(), tape.param_grads[id(bg)].clone(), tape.param_grads[id(wg)].clone()]
    a, a2, b = run(True), run(True), run(False)
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
    # the classifier's dx feeds the BatchNorm: with the SAME dy the two BatchNorm kernels are bit-equal; here dy itself comes from two
    # different backward-data kernels, so compare to rounding ...
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 2e-5 * (v.abs().max().item() + 1e-6)
    # ... and bit-equal when both BatchNorm kernels see the same gradient tensor
    L = _lib_mod().lib()
    st = torch.cuda.current_stream().cuda_stream
    M = B * H * W
    xn = nhwc(x)
    dy = torch.zeros(M, C, device=DEV)
    dy[rows.to(DEV)] = torch.randn(300, C, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV)
    flags = torch.empty(M, dtype=torch.uint8, device=DEV)
    _lib_mod().check(L.pp_row_flags(dy.data_ptr(), C, M, C, flags.data_ptr(), st), "flags")
    mean = xn.reshape(M, C).mean(0).contiguous()
    invstd = (1.0 / torch.sqrt(xn.reshape(M, C).var(0, unbiased=False) + 1e-5)).contiguous()
    g_, b_ = gamma.to(DEV).contiguous(), beta.to(DEV).contiguous()
    sync, ws = E._bn_exchange(torch.device(DEV))
    outs = []
    for fl in (None, flags):
        dx, dg, db = torch.empty(M, C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        args = [xn.data_ptr(), C, dy.data_ptr(), C, None, 0, act, M, C, mean.data_ptr(), invstd.data_ptr(), g_.data_ptr(), dg.data_ptr(),
                db.data_ptr(), dx.data_ptr(), C, None, 0, 1.0, b_.data_ptr(), ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel()]
        if fl is None:
            _lib_mod().check(L.pp_bn_bwd_fused(*args, st), "dense")
        else:
            _lib_mod().check(L.pp_bn_bwd_fused_sparse(*args, fl.data_ptr(), st), "rows")
        torch.cuda.synchronize()
        outs.append((dx, dg, db))
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    # torch autograd
    xr, gr, br, wr = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), wc.clone().requires_grad_(True)
    z = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
    z = F.relu(z) if act == 1 else (F.relu6(z) if act == 2 else z)
    F.conv2d(z, wr).backward(dlog)
    close(nchw(a[0]), xr.grad, what="dx through the sparse-row kernels")
    close(a[1].cpu(), gr.grad, what="dgamma")
    close(a[2].cpu(), br.grad, what="dbeta")
    close(oihw(a[3]), wr.grad, what="classifier dW")


ROWS_FWD = [(4, 128, 256, 32, 16, 0), (4, 64, 128, 96, 24, 0), (4, 64, 128, 144, 32, 0), (2, 100, 83, 144, 24, 0),      # narrow output: project
            (4, 128, 256, 16, 96, 1), (4, 64, 128, 24, 144, 1), (2, 96, 96, 32, 192, 0), (1, 129, 131, 16, 96, 1)]        # narrow input: expand


@pytest.mark.parametrize("case", ROWS_FWD, ids=[str(c) for c in ROWS_FWD])
def test_pointwise_forward_of_the_narrow_layers(case):
    """mobilenet_v2.py:48-56 on the 1/2- and 1/4-resolution maps: the project convolutions (32 -> 16, 96 -> 24, 144 -> 24 / 32) run
    conv1x1_rows_kernel<.., false>, the expand convolutions (16 -> 96, 24 -> 144, 32 -> 192, fixed padding folded in)
    conv1x1_fwd_widen_kernel - rows through LDS, VALU, weights as scalar operands.  Against torch, against the MFMA path
    (pp_debug_set_conv_rows(1)) to fp32 rounding, ragged row counts; bit-reproducible; the gradients of the same layers too."""
    B, H, W, Cin, Cout, pad = case
    gen = torch.Generator().manual_seed(H + W + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    dy = torch.randn(B, Cout, H + 2 * pad, W + 2 * pad, generator=gen)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=pad)
    yr.backward(dy)
    L = _lib_mod().lib()
    def run(variant):
        L.pp_debug_set_conv_rows(variant)
        try:
            tape = E.Tape()
            xv, wg = E.Var(nhwc(x)), gparam(hwio(w))
            yv = E.conv2d(tape, xv, wg, None, 1, pad, 1)
            y = nchw(yv.t)
            tape.backward(yv, nhwc(dy))
            torch.cuda.synchronize()
            return y, nchw(xv.grad), oihw(tape.param_grads[id(wg)])
        finally:
            L.pp_debug_set_conv_rows(0)
    a, a2, b = run(0), run(0), run(1)
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
    close(a[0], yr.detach(), what="forward of the narrow pointwise layer")
    close(a[1], xr.grad, what="dx")
    close(a[2], wr.grad, what="dW")
    assert (a[0] - b[0]).abs().max().item() <= 2e-5 * b[0].abs().max().item()


STEM_WGRAD = [(4, 256, 512, 32), (2, 256, 256, 24), (3, 200, 260, 32), (1, 512, 258, 16)]


@pytest.mark.parametrize("case", STEM_WGRAD, ids=[str(c) for c in STEM_WGRAD])
def test_mobilenet_stem_weight_gradient(case):
    """mobilenet_v2.py:7-12 Conv2d(3, C, 3, stride 2, padding 1) on an even-sized image: the specialised weight-gradient kernel
    (wgrad_stem3x3s2_kernel: nine contiguous floats per tap row, the row lanes of a block reduced in LDS) against torch in fp64 and
    against the generic narrow-input kernel (pp_debug_set_conv_variant bit 23) - same sums in another order; bit-reproducible."""
    B, H, W, C = case
    gen = torch.Generator().manual_seed(B * H + C)
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(C, 3, 3, 3, generator=gen) / np.sqrt(27)
    dy = torch.randn(B, C, H // 2, W // 2, generator=gen)
    wr = w.double().requires_grad_(True)
    F.conv2d(x.double(), wr, stride=2, padding=1).backward(dy.double())
    L = _lib_mod().lib()
    def run(variant):
        L.pp_debug_set_conv_variant(variant)
        try:
            tape = E.Tape()
            wg = gparam(hwio(w))
            yv = E.conv2d(tape, E.Var(nhwc(x), needs_grad=False), wg, None, 2, 1, 1)
            tape.backward(yv, nhwc(dy))
            torch.cuda.synchronize()
            return oihw(tape.param_grads[id(wg)])
        finally:
            L.pp_debug_set_conv_variant(0)
    a, a2, b = run(0), run(0), run(1 << 23)
    assert torch.equal(a, a2)
    scale = wr.grad.abs().max().item()
    assert (a.double() - wr.grad).abs().max().item() <= 2e-6 * scale * np.sqrt(B * H * W / 4) / 50, "stem dW vs fp64"
    assert (b.double() - wr.grad).abs().max().item() <= 2e-6 * scale * np.sqrt(B * H * W / 4) / 50, "generic kernel vs fp64"
    assert (a - b).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("case", [(4, 256, 512, 64), (2, 130, 256, 64), (1, 512, 128, 48)], ids=["cityscapes-batch", "ragged-rows", "narrow-48-generic"])
def test_resnet_stem_weight_gradient(case):
    """resnet_models.py:115-117 Conv2d(3, 64, 7, stride 2, padding 3) on an even-sized image: the specialised weight-gradient kernel
    (wgrad_stem7x7s2_kernel: a lane per (tap, channel) of a tap row, dy read with wave-uniform loads; 64 output channels only) against torch in
    fp64 and against the generic narrow-input kernel (pp_debug_set_conv_variant bit 23) - same sums in another order; bit-reproducible."""
    B, H, W, C = case
    gen = torch.Generator().manual_seed(B * H + C + 7)
    x = torch.randn(B, 3, H, W, generator=gen)
    w = torch.randn(C, 3, 7, 7, generator=gen) / np.sqrt(147)
    dy = torch.randn(B, C, H // 2, W // 2, generator=gen)
    wr = w.double().requires_grad_(True)
    F.conv2d(x.double(), wr, stride=2, padding=3).backward(dy.double())
    L = _lib_mod().lib()
    def run(variant):
        L.pp_debug_set_conv_variant(variant)
        try:
            tape = E.Tape()
            wg = gparam(hwio(w))
            yv = E.conv2d(tape, E.Var(nhwc(x), needs_grad=False), wg, None, 2, 3, 1)
            tape.backward(yv, nhwc(dy))
            torch.cuda.synchronize()
            return oihw(tape.param_grads[id(wg)])
        finally:
            L.pp_debug_set_conv_variant(0)
    a, a2, b = run(0), run(0), run(1 << 23)
    assert torch.equal(a, a2)
    scale = wr.grad.abs().max().item()
    assert (a.double() - wr.grad).abs().max().item() <= 2e-6 * scale * np.sqrt(B * H * W / 4) / 50, "stem dW vs fp64"
    assert (b.double() - wr.grad).abs().max().item() <= 2e-6 * scale * np.sqrt(B * H * W / 4) / 50, "generic kernel vs fp64"
    assert (a - b).abs().max().item() <= 1e-5 * scale


STRIDED_BWD = [(2, 17, 23, 128, 128, 3, 2, 1, 1), (2, 16, 24, 256, 512, 1, 2, 0, 1), (1, 15, 15, 64, 96, 3, 2, 1, 1),
               (2, 19, 22, 32, 48, 3, 2, 2, 2), (1, 21, 20, 8, 32, 7, 2, 3, 1), (2, 14, 17, 16, 24, 3, 3, 0, 1), (2, 13, 11, 64, 64, 1, 2, 0, 1)]


@pytest.mark.parametrize("case", STRIDED_BWD, ids=[str(c) for c in STRIDED_BWD])
def test_conv2d_strided_bwd_data(case):
    """ResNet layer2.0: 3x3 stride-2 conv2 and 1x1 stride-2 downsample need dL/dx (resnet_models.py:63-64,142-144)."""
    B, H, W, Cin, Cout, k, stride, pad, dil = case
    torch.manual_seed(11)
    x = torch.randn(B, Cin, H, W, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).requires_grad_(True)
    yr = F.conv2d(x, w, None, stride, pad, dil)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    wg = gparam(hwio(w.detach()))
    yv = E.conv2d(tape, xv, wg, None, stride, pad, dil)
    close(nchw(yv.t), yr.detach(), what="strided fwd")
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, what="strided dX")
    close(oihw(tape.param_grads[id(wg)]), w.grad, what="strided dW")


@pytest.mark.parametrize("case", STRIDED_BWD, ids=[str(c) for c in STRIDED_BWD])
def test_conv2d_strided_bwd_data_by_phases_equals_masked_taps(case):
    """pp_conv2d_bwd_data of a strided convolution runs as stride^2 stride-1 problems (one per pixel class ih % s, iw % s) plus an
    interleave pass; pp_debug_set_conv_variant bit 24 selects the single launch with divisibility-masked taps.  Same dX, also
    when accumulating into an existing gradient."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    B, H, W, Cin, Cout, k, stride, pad, dil = case
    Ho, Wo = E.out_size(H, k, stride, pad, dil), E.out_size(W, k, stride, pad, dil)
    torch.manual_seed(21)
    dy = torch.randn(B, Ho, Wo, Cout, device=DEV)
    w = torch.randn(k, k, Cin, Cout, device=DEV) / np.sqrt(Cin * k * k)
    base = torch.randn(B, H, W, Cin, device=DEV)
    ws = torch.empty(max(1, L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil)), dtype=torch.uint8, device=DEV)
    assert ws.numel() >= B * H * W * Cin * 4
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for variant in (0, 16777216):
        L.pp_debug_set_conv_variant(variant)
        try:
            for accumulate in (0, 1):
                dx = base.clone()
                _lib.check(L.pp_conv2d_bwd_data(dy.data_ptr(), Cout, B, Ho, Wo, Cout, w.data_ptr(), k, k, stride, pad, dil, dx.data_ptr(), Cin,
                                                H, W, Cin, accumulate, ws.data_ptr(), ws.numel(), st), "bwd_data")
                outs.append(dx)
        finally:
            L.pp_debug_set_conv_variant(0)
    close(outs[0].cpu(), outs[2].cpu(), tol=2e-5, what="phases vs masked taps")
    close(outs[1].cpu(), outs[3].cpu(), tol=2e-5, what="phases vs masked taps, accumulating")
    close((outs[1] - base).cpu(), outs[0].cpu(), tol=2e-5, what="accumulate adds")


@pytest.mark.parametrize("shape", [(2, 16, 24, 128), (3, 9, 7, 128), (2, 32, 48, 128), (1, 5, 6, 64)])
def test_groupnorm_relu(shape):
    B, H, W, C = shape
    torch.manual_seed(12)
    x = (torch.randn(B, C, H, W) * 1.5 + 0.3).requires_grad_(True)
    gn = torch.nn.GroupNorm(32, C)
    with torch.no_grad():
        gn.weight.copy_(torch.rand(C) + 0.5)
        gn.bias.copy_(torch.randn(C) * 0.3)
    yr = F.relu(gn(x))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    g, b = gparam(gn.weight), gparam(gn.bias)
    yv = E.group_norm_relu(tape, xv, g, b, 32, True)
    close(nchw(yv.t), yr.detach(), what="gn fwd")
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, tol=2e-4, what="gn dX")
    close(tape.param_grads[id(g)].cpu(), gn.weight.grad, tol=2e-4, what="gn dgamma")
    close(tape.param_grads[id(b)].cpu(), gn.bias.grad, tol=2e-4, what="gn dbeta")


@pytest.mark.parametrize("shape", [(2, 32, 48, 64), (1, 17, 23, 64), (2, 8, 8, 16)])
def test_maxpool_fwd_bwd_incl_relu_ties(shape):
    B, H, W, C = shape
    torch.manual_seed(13)
    x = F.relu(torch.randn(B, C, H, W)).requires_grad_(True)      # many exact zeros: tie rule must match torch
    yr = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    yv = E.max_pool2d(tape, xv, 3, 2, 1)
    assert torch.equal(nchw(yv.t), yr.detach())
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, tol=1e-6, what="maxpool bwd")


def test_add_and_nchw_roundtrip():
    torch.manual_seed(14)
    a, b = torch.randn(2, 12, 5, 7), torch.randn(2, 12, 5, 7)
    tape = E.Tape()
    av, bv = E.Var(nhwc(a)), E.Var(nhwc(b))
    s = E.add(tape, av, bv)
    o = E.nhwc_to_nchw(tape, s)
    assert torch.equal(o.t.cpu(), a + b)
    dy = torch.randn(2, 12, 5, 7)
    tape.backward(o, dy.to(DEV))
    assert torch.equal(nchw(av.grad), dy) and torch.equal(nchw(bv.grad), dy)


@pytest.mark.parametrize("case", [(4, 16, 32, 320, 256, 3, 1, 6, 6), (4, 16, 32, 960, 320, 1, 1, 0, 1), (2, 16, 32, 1280, 256, 1, 1, 0, 1),
                                  (2, 9, 13, 576, 96, 1, 1, 0, 1), (2, 8, 8, 144, 19, 3, 1, 1, 1), (1, 20, 18, 3, 64, 7, 2, 3, 1)])
def test_conv2d_split_k_matches_single_pass(case):
    """Few-tile layers slice their K loop over grid.y (partials + fixed-order reduce).  Same numbers (to fp32
    summation-order noise) as the single-pass kernel, bit-identical run to run, correct into a channel slice."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    B, H, W, Cin, Cout, k, stride, pad, dil = case
    KSPLIT_OFF = 1 << 25
    # deep-K pointwise layers split K INSIDE the block (conv1x1_ksplit_dma_kernel: no workspace); switched off, the tiled kernel
    # slices K over grid.y - both must agree with the single pass
    in_block = L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil) == 0
    if in_block:
        assert k == 1 and Cin >= 256, "case does not split"
        L.pp_debug_set_conv_variant(KSPLIT_OFF)
        try:
            assert L.pp_conv2d_fwd_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad, dil) > 0, "case does not split"
        finally:
            L.pp_debug_set_conv_variant(0)
    torch.manual_seed(3)
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.randn(k, k, Cin, Cout, device=DEV) / np.sqrt(Cin * k * k)
    bias = torch.randn(Cout, device=DEV)
    Ho, Wo = E.out_size(H, k, stride, pad, dil), E.out_size(W, k, stride, pad, dil)
    dy = torch.randn(B, Ho, Wo, Cout, device=DEV)

    def run():
        tape = E.Tape()
        xv = E.Var(x.clone()); xv.needs_grad = True
        wide = torch.full((B, Ho, Wo, Cout + 8), 5.0, device=DEV)
        yv = E.conv2d(tape, xv, w, bias, stride, pad, dil, dst=wide[..., 4:4 + Cout])
        assert (wide[..., :4] == 5.0).all() and (wide[..., 4 + Cout:] == 5.0).all()
        y = yv.t.clone()
        dx = None
        if stride == 1:
            tape.backward(yv, dy)
            dx = xv.grad.clone()
        return y, dx

    y1, dx1 = run()
    y2, dx2 = run()
    assert torch.equal(y1, y2) and (dx1 is None or torch.equal(dx1, dx2))
    L.pp_debug_set_conv_variant(64 | KSPLIT_OFF)          # split-K off (both forms)
    try:
        y0, dx0 = run()
    finally:
        L.pp_debug_set_conv_variant(0)
    close(y1, y0, tol=2e-5, what="split-K fwd vs single pass")
    if dx1 is not None:
        close(dx1, dx0, tol=2e-5, what="split-K bwd-data vs single pass")
    if in_block:
        L.pp_debug_set_conv_variant(KSPLIT_OFF)           # the grid.y split-K + reduce form of the same layer
        try:
            y3, dx3 = run()
            y4, dx4 = run()
        finally:
            L.pp_debug_set_conv_variant(0)
        assert torch.equal(y3, y4) and torch.equal(dx3, dx4)
        close(y3, y0, tol=2e-5, what="grid split-K fwd vs single pass")
        close(dx3, dx0, tol=2e-5, what="grid split-K bwd-data vs single pass")
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu(), w.permute(3, 2, 0, 1).cpu(), bias.cpu(), stride, pad, dil)
    close(nchw(y1), ref, what="split-K fwd vs torch")


def test_conv2d_channel_slices():
    """Inputs/outputs that are channel slices of wider buffers (the zero-copy concat of aspp.py:73)."""
    torch.manual_seed(0)
    B, H, W = 2, 8, 12
    x = torch.randn(B, 48, H, W)
    w = torch.randn(32, 48, 1, 1) / 7
    big_in = torch.zeros(B, H, W, 80, device=DEV)
    big_in[..., 16:64] = nhwc(x)
    big_out = torch.full((B, H, W, 96), 7.0, device=DEV)
    tape = E.Tape(enabled=False)
    E.conv2d(tape, E.Var(big_in[..., 16:64]), hwio(w), None, dst=big_out[..., 32:64])
    close(nchw(big_out[..., 32:64]), F.conv2d(x, w), what="sliced conv")
    assert (big_out[..., :32] == 7.0).all() and (big_out[..., 64:] == 7.0).all()


DW_CASES = [(2, 18, 34, 32, 1, 0, 1), (2, 19, 35, 96, 2, 0, 1), (2, 20, 36, 960, 1, 0, 2), (1, 13, 9, 144, 2, 0, 1),
            (2, 16, 16, 24, 1, 1, 1), (1, 7, 11, 16, 1, 1, 1), (2, 9, 6, 8, 1, 0, 1), (1, 5, 5, 4, 1, 2, 1), (3, 6, 13, 192, 1, 1, 1),
            (2, 12, 14, 32, 1, 2, 2), (1, 9, 7, 8, 1, 1, 2), (1, 5, 6, 4, 1, 3, 2)]      # (dilation 2 with padding: the four-outputs-per-thread form)


@pytest.mark.parametrize("case", DW_CASES, ids=[str(c) for c in DW_CASES])
def test_dwconv_fwd_bwd(case):
    B, H, W, C, stride, pad, dil = case
    torch.manual_seed(1)
    x = torch.randn(B, C, H, W, requires_grad=True)
    w = (torch.randn(C, 1, 3, 3) / 3).requires_grad_(True)
    yr = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil, groups=C)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    wg = gparam(w.detach()[:, 0].permute(1, 2, 0).contiguous())
    yv = E.dwconv3x3(tape, xv, wg, stride, pad, dil)
    close(nchw(yv.t), yr.detach(), what="dw fwd")
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, what="dw dX")
    close(tape.param_grads[id(wg)].permute(2, 0, 1).cpu()[:, None], w.grad, what="dw dW")


def test_batched_weight_gradient_reduces_are_bit_identical(monkeypatch):
    """PIXELPICK_BATCH_REDUCE: the layers on the weight-gradient stream run pp_*_bwd_weight_partials and ONE
    pp_wgrad_reduce_batch launch reduces them all (table in the kernel arguments) - the arithmetic of the per-layer reduce
    kernels, so every gradient is bit-equal to the undeferred path.  One tape with every job kind: 3x3 with dead taps (kind 1, tap
    map), pointwise, narrow-input / narrow-output layers (kind 2), depthwise stride 1 / 2 (kind 3), a bias layer (not
    deferrable: completed in place), a parameter used twice (forces an early flush) and > 64 layers (several launches)."""
    convs = [(2, 16, 32, 320, 256, 3, 1, 6, 6, False), (2, 16, 32, 64, 96, 3, 1, 18, 18, False), (2, 10, 12, 960, 160, 1, 1, 0, 1, False),
             (2, 128, 160, 32, 16, 1, 1, 0, 1, False), (2, 128, 136, 96, 24, 1, 1, 0, 1, False), (2, 256, 264, 3, 32, 3, 2, 1, 1, False),
             (2, 12, 12, 128, 128, 3, 1, 1, 1, True), (2, 9, 11, 24, 144, 1, 1, 0, 1, False), (1, 24, 20, 304, 256, 3, 1, 1, 1, False)]
    convs = convs + [(2, 9, 11, 24 + 8 * i, 40, 1, 1, 0, 1, False) for i in range(60)]
    dws = [(2, 18, 34, 32, 1, 0, 1), (2, 19, 35, 96, 2, 0, 1), (2, 20, 36, 960, 1, 0, 2), (4, 16, 32, 384, 1, 1, 1)]
    gen = torch.Generator().manual_seed(9)
    data = []
    for (B, H, W, Cin, Cout, k, st, pad, dil, hb) in convs:
        data.append(("c", nhwc(torch.randn(B, Cin, H, W, generator=gen)), (torch.randn(k, k, Cin, Cout, generator=gen) / np.sqrt(Cin * k * k)).to(DEV),
                     torch.randn(Cout, generator=gen).to(DEV) if hb else None, st, pad, dil))
    for (B, H, W, C, st, pad, dil) in dws:
        data.append(("d", nhwc(torch.randn(B, C, H, W, generator=gen)), (torch.randn(3, 3, C, generator=gen) / 3).to(DEV), None, st, pad, dil))

    def run(batch):
        monkeypatch.setattr(E, "_BATCH_REDUCE", batch)
        tape = E.Tape()
        params, outs = [], []
        shared = None
        for i, (kind, x, w, b, st, pad, dil) in enumerate(data):
            w = w.clone().requires_grad_(True)
            b = b.clone().requires_grad_(True) if b is not None else None
            params += [w] + ([b] if b is not None else [])
            y = E.conv2d(tape, E.Var(x), w, b, st, pad, dil) if kind == "c" else E.dwconv3x3(tape, E.Var(x), w, st, pad, dil)
            outs.append(y)
            if i == 2:                    # the same weight on a second input: its two gradients are accumulated on the tape
                shared = (w, E.conv2d(tape, E.Var(x * 0.5), w, None, st, pad, dil))
                outs.append(shared[1])
        g = torch.Generator(device=DEV).manual_seed(4)
        # one backward per output on the same tape is not how the tape works: chain them through a sum of means instead
        for y in outs:
            y.grad = torch.randn(y.t.shape, device=DEV, generator=g)
        last = outs[-1]
        tape.backward(last, last.grad)
        torch.cuda.synchronize()
        return [tape.param_grads[id(p)].clone() for p in params]

    a = run(True)
    b = run(False)
    assert len(a) == len(b) and len(a) >= 70
    bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
    assert not bad, f"gradients {bad} differ"


FUSED_CONV_BN = [
    # B, H, W, Cin, Cout, act, residual   (pointwise, stride 1)
    (4, 16, 32, 64, 384, 2, False),      # 64x64-tiled LDS-DMA kernel, 234 blocks
    (4, 16, 32, 96, 576, 2, False),
    (4, 32, 64, 32, 192, 2, False),      # 1/8 resolution, 8192 rows
    (4, 18, 34, 160, 960, 2, False),     # 585 blocks of 64x64 are too many: 128x64 tiles
    (4, 34, 66, 144, 32, 0, False),      # 128x32 register-staged kernel
    (4, 16, 32, 960, 160, 0, True),      # in-block split-K kernel, BatchNorm + residual (InvertedResidual project)
    (4, 16, 32, 384, 64, 0, False),
    (4, 16, 32, 320, 256, 1, False),     # ASPP 1x1 branch
    (3, 23, 30, 576, 96, 0, True),       # ragged rows (2070)
    (2, 9, 11, 64, 96, 2, False),        # 198 rows: a handful of blocks
]


@pytest.mark.parametrize("case", FUSED_CONV_BN, ids=[str(c) for c in FUSED_CONV_BN])
def test_conv_batchnorm_in_one_launch(case, monkeypatch):
    """PIXELPICK_CONV_BN_FUSE: a training BatchNorm (+ residual, activation) right behind a dense convolution whose grid is co-resident is
    finished in the convolution's epilogue (pp_conv2d_fwd_bn_train: the BatchNorm kernels' cross-block exchange inside the convolution).
    Output, batch / running statistics and every gradient equal the two-launch path to fp32 rounding (the column sums are taken over
    other partitions of the rows), the result equals torch, repeated launches are bit-identical, and the convolution really was deferred."""
    B, H, W, Cin, Cout, act, with_res = case
    gen = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(Cout, Cin, 1, 1, generator=gen) / np.sqrt(Cin)
    gamma, beta = torch.rand(Cout, generator=gen) + 0.5, torch.randn(Cout, generator=gen)
    res = torch.randn(B, Cout, H, W, generator=gen) if with_res else None
    dy = torch.randn(B, Cout, H, W, generator=gen)

    def run(fuse):
        monkeypatch.setattr(E, "_CONV_BN_FUSE", fuse)
        tape = E.Tape()
        xv = E.Var(nhwc(x))
        wg, gg, bg = gparam(hwio(w)), gparam(gamma), gparam(beta)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        c = E.conv2d(tape, xv, wg, None, 1, 0, 1)
        deferred = c._pending is not None
        rvv = E.Var(nhwc(res)) if with_res else None
        y = E.batch_norm_act(tape, c, gg, bg, rm, rv, True, act, rvv)
        yt = y.t.clone()
        tape.backward(y, nhwc(dy))
        torch.cuda.synchronize()
        return deferred, [yt, rm, rv, xv.grad.clone(), tape.param_grads[id(wg)].clone(), tape.param_grads[id(gg)].clone(),
                          tape.param_grads[id(bg)].clone()] + ([rvv.grad.clone()] if with_res else [])

    d1, a = run(True)
    d1b, a2 = run(True)
    d0, b = run(False)
    assert d1 and d1b and not d0, "the fused path was not taken"
    for u, v in zip(a, a2):
        assert torch.equal(u, v)                                      # bit-reproducible
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 2e-5 * (v.abs().max().item() + 1e-6)
    # torch reference of the forward
    xr = x.clone()
    yr = F.batch_norm(F.conv2d(xr, w), None, None, gamma, beta, True, 0.1, 1e-5)
    if with_res:
        yr = yr + res
    yr = F.relu(yr) if act == 1 else (F.relu6(yr) if act == 2 else yr)
    close(nchw(a[0]), yr, what="conv + BatchNorm fused forward")


FUSED_BN_BWD = [(4, 16, 32, 384, 64, 2), (4, 16, 32, 576, 96, 2), (4, 32, 64, 192, 32, 2), (3, 23, 30, 384, 64, 2), (4, 16, 32, 384, 64, 0),
                (4, 16, 32, 384, 64, 1), (2, 9, 11, 96, 32, 2)]


@pytest.mark.parametrize("case", FUSED_BN_BWD, ids=[str(c) for c in FUSED_BN_BWD])
def test_batchnorm_backward_inside_the_consumer_convolutions_backward_data(case, monkeypatch):
    """PIXELPICK_CONV_BN_FUSE_BWD: depthwise -> BatchNorm -> activation -> pointwise convolution, the activated tensor read by that
    convolution only (single_consumer): the convolution's backward-data launch also runs the BatchNorm backward
    (pp_conv2d_bwd_data_bn_bwd) - the gradient of the BatchNorm's output is never written and the BatchNorm node is skipped.
    Every gradient equals the two-launch path to fp32 rounding and torch autograd; repeated runs are bit-identical."""
    B, H, W, C, Cout, act = case
    gen = torch.Generator().manual_seed(C + Cout + act)
    x = torch.randn(B, C, H, W, generator=gen)
    wd = torch.randn(C, 1, 3, 3, generator=gen) / 3
    w = torch.randn(Cout, C, 1, 1, generator=gen) / np.sqrt(C)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    dy = torch.randn(B, Cout, H, W, generator=gen)
    def run(fuse):
        monkeypatch.setattr(E, "_CONV_BN_FUSE_BWD", fuse)
        tape = E.Tape()
        xv = E.Var(nhwc(x))
        wdg, wg, gg, bg = gparam(wd[:, 0].permute(1, 2, 0).contiguous()), gparam(hwio(w)), gparam(gamma), gparam(beta)
        d = E.dwconv3x3(tape, xv, wdg, 1, 1, 1)
        y = E.batch_norm_act(tape, d, gg, bg, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, act, single_consumer=True)
        fused_ctx = y._bn_bwd_ctx is not None
        c = E.conv2d(tape, y, wg, None, 1, 0, 1)
        tape.backward(c, nhwc(dy))
        torch.cuda.synchronize()
        return fused_ctx, [xv.grad.clone()] + [tape.param_grads[id(t)].clone() for t in (wdg, wg, gg, bg)]

    c1, a = run(True)
    _, a2 = run(True)
    _, b = run(False)
    assert c1
    assert E._conv_bn_bwd_fusable(B, H, W, C, Cout, 1, 1, 1, 0, 1), "the fused path was not available for this case"
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 2e-5 * (v.abs().max().item() + 1e-6)
    xr, wdr, wr = x.clone().requires_grad_(True), wd.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(F.conv2d(xr, wdr, padding=1, groups=C), None, None, gr, br, True, 0.1, 1e-5)
    z = F.relu(z) if act == 1 else (F.relu6(z) if act == 2 else z)
    F.conv2d(z, wr).backward(dy)
    close(nchw(a[0]), xr.grad, what="dx through the fused BatchNorm backward")
    close(a[3].cpu(), gr.grad, what="dgamma")
    close(a[4].cpu(), br.grad, what="dbeta")


LAST_CONSUMER_BN_BWD = [
    # B, H, W, Cexp (project conv input), C (BatchNorm channels), pad of the consuming 1x1 convolution, BatchNorm residual, consumers
    (4, 16, 32, 384, 64, 0, True, 2),      # 1/16 resolution, in-block split-K backward-data, one 64-wide tile
    (4, 16, 32, 384, 96, 1, True, 1),      # ragged second column tile; the fixed padding folded into the convolution
    (4, 16, 32, 576, 160, 0, False, 2),    # three column tiles, no residual into the BatchNorm, residual add behind
    (4, 16, 32, 192, 64, 1, False, 1),
    (4, 32, 64, 144, 32, 0, True, 2),      # 1/8 resolution, 32 channels: the 128x32 register-staged kernel (its grid split-K plan as one pass)
    (4, 32, 64, 144, 32, 1, False, 1),
    (4, 32, 64, 192, 64, 0, True, 2),      # 64x64 LDS-DMA tiles with a gradient already there and a residual gradient to write
    (2, 9, 11, 96, 32, 0, True, 2),        # ragged rows
    (2, 9, 11, 96, 32, 1, True, 1),
]


@pytest.mark.parametrize("case", LAST_CONSUMER_BN_BWD, ids=[str(c) for c in LAST_CONSUMER_BN_BWD])
def test_batchnorm_backward_inside_the_last_consumers_backward_data(case, monkeypatch):
    """mobilenet_v2.py:63-66 across two InvertedResiduals: project convolution -> BatchNorm (+ the block's residual) = the block
    output, read FIRST by the next block's expand convolution (fixed padding folded in) and then by that block's residual add
    (`consumers=2`).  The expand convolution's backward runs last: its backward-data launch adds the gradient that is already
    there, runs the BatchNorm backward and writes the gradient of the BatchNorm's residual input (pp_conv2d_bwd_data_bn_bwd with
    grad_in / dres).  Every gradient equals the separate-launch path to fp32 rounding and torch autograd; repeated runs are
    bit-identical; the BatchNorm output is closed afterwards."""
    B, H, W, Cexp, C, pad, with_res, ncons = case
    gen = torch.Generator().manual_seed(Cexp + C + pad)
    x = torch.randn(B, Cexp, H, W, generator=gen)
    wp = torch.randn(C, Cexp, 1, 1, generator=gen) / np.sqrt(Cexp)
    we = torch.randn(6 * C, C, 1, 1, generator=gen) / np.sqrt(C)
    wf = torch.randn(C, 6 * C, 1, 1, generator=gen) / np.sqrt(6 * C)
    res = torch.randn(B, C, H, W, generator=gen)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    Ho, Wo = H + 2 * pad, W + 2 * pad
    dy = torch.randn(B, C, Ho, Wo, generator=gen)
    if ncons == 2:
        assert pad == 0
    def run(fuse):
        monkeypatch.setattr(E, "_CONV_BN_FUSE_BWD", fuse)
        tape = E.Tape()
        xv, rv = E.Var(nhwc(x)), (E.Var(nhwc(res)) if with_res else None)
        wpg, weg, wfg, gg, bg = gparam(hwio(wp)), gparam(hwio(we)), gparam(hwio(wf)), gparam(gamma), gparam(beta)
        pr = E.conv2d(tape, xv, wpg, None, 1, 0, 1)
        y = E.batch_norm_act(tape, pr, gg, bg, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, E.ACT_NONE, rv, consumers=ncons)
        assert y._bn_bwd_ctx is not None
        e = E.conv2d(tape, y, weg, None, 1, pad, 1)
        f = E.conv2d(tape, e, wfg, None, 1, 0, 1)
        out = E.add(tape, f, y) if ncons == 2 else f
        tape.backward(out, nhwc(dy))
        torch.cuda.synchronize()
        return y._closed, [xv.grad.clone()] + ([rv.grad.clone()] if with_res else []) + [tape.param_grads[id(t)].clone() for t in (wpg, weg, wfg, gg, bg)]

    c1, a = run(True)
    _, a2 = run(True)
    c0, b = run(False)
    assert E._conv_bn_bwd_fusable(B, H, W, C, 6 * C, 1, 1, 1, pad, 1), "the fused path is not available for this case"
    assert c1 and not c0, "the fused path was not taken"
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 2e-5 * (v.abs().max().item() + 1e-6)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    wpr, wer, wfr = wp.clone().requires_grad_(True), we.clone().requires_grad_(True), wf.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.batch_norm(F.conv2d(xr, wpr), None, None, gr, br, True, 0.1, 1e-5)
    if with_res:
        yr = yr + rr
    fr = F.conv2d(F.conv2d(yr, wer, padding=pad), wfr)
    (fr + yr if ncons == 2 else fr).backward(dy)
    i = 0
    close(nchw(a[i]), xr.grad, what="dx through the fused BatchNorm backward"); i += 1
    if with_res:
        close(nchw(a[i]), rr.grad, what="gradient of the BatchNorm's residual input"); i += 1
    close(a[i + 3].cpu(), gr.grad, what="dgamma")
    close(a[i + 4].cpu(), br.grad, what="dbeta")


def test_a_late_gradient_at_a_closed_batchnorm_output_raises(monkeypatch):
    """A `consumers` hint that is too small must not lose a gradient silently: the BatchNorm output is closed once its backward ran
    inside the convolution's launch, and a gradient arriving afterwards raises."""
    monkeypatch.setattr(E, "_CONV_BN_FUSE_BWD", True)
    B, H, W, Cexp, C = 4, 16, 32, 384, 64
    gen = torch.Generator().manual_seed(5)
    tape = E.Tape()
    xv = E.Var(nhwc(torch.randn(B, Cexp, H, W, generator=gen)))
    wpg = gparam(hwio(torch.randn(C, Cexp, 1, 1, generator=gen)))
    gg, bg = gparam(torch.ones(C)), gparam(torch.zeros(C))
    y = E.batch_norm_act(tape, E.conv2d(tape, xv, wpg, None, 1, 0, 1), gg, bg, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True,
                         E.ACT_NONE, None, consumers=1)                  # wrong: two ops read y
    first = E.add(tape, y, y)                                            # recorded first, so its backward runs LAST
    e = E.conv2d(tape, y, gparam(hwio(torch.randn(384, C, 1, 1, generator=gen))), None, 1, 0, 1)
    e2 = E.conv2d(tape, e, gparam(hwio(torch.randn(C, 384, 1, 1, generator=gen))), None, 1, 0, 1)
    out = E.add(tape, e2, first)
    with pytest.raises(RuntimeError, match="consumers"):
        tape.backward(out, torch.randn(B, H, W, C, device=DEV))
    torch.cuda.synchronize()


def _lib_mod():
    from pixelpick_amd import _lib
    return _lib


@pytest.fixture(params=[True, False], ids=["bn-1launch", "bn-3launch"])
def bn_fused(request):
    old = E._BN_FUSED
    E._BN_FUSED = request.param
    yield request.param
    E._BN_FUSED = old


@pytest.mark.parametrize("act,with_res", [(0, False), (1, False), (2, False), (0, True), (1, True)])
@pytest.mark.parametrize("shape", [(4, 18, 34, 96), (2, 7, 5, 16), (4, 16, 32, 960), (3, 1, 1, 256), (2, 6, 10, 2048),
                                   (2, 23, 30, 256), (2, 13, 18, 960), (2, 17, 22, 960),
                                   # strip widths 6/4/5/7/8-with-a-ragged-last-strip/1/2/3, and a many-chunk map
                                   (2, 9, 11, 24), (2, 9, 11, 144), (2, 9, 11, 304), (2, 9, 11, 100), (2, 9, 11, 28),
                                   (2, 9, 11, 44), (2, 33, 9, 4), (2, 9, 11, 8), (2, 9, 11, 12), (2, 128, 160, 32)])
def test_batchnorm_train_fwd_bwd(shape, act, with_res, bn_fused):
    B, H, W, C = shape
    torch.manual_seed(2)
    x = (torch.randn(B, C, H, W) * 2 + 0.5).requires_grad_(True)
    res = torch.randn(B, C, H, W).requires_grad_(True) if with_res else None
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5)
        bn.bias.copy_(torch.randn(C) * 0.3)
        bn.running_mean.copy_(torch.randn(C) * 0.1)
        bn.running_var.copy_(torch.rand(C) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    z = bn(x)
    if with_res:
        z = z + res
    yr = {0: lambda t: t, 1: F.relu, 2: F.relu6}[act](z)
    dy = torch.randn_like(yr)
    yr.backward(dy)

    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    rv = E.Var(nhwc(res.detach())) if with_res else None
    g, bta = gparam(bn.weight), gparam(bn.bias)
    rm, rvv = rm0.to(DEV), rv0.to(DEV)
    yv = E.batch_norm_act(tape, xv, g, bta, rm, rvv, True, act, rv)
    close(nchw(yv.t), yr.detach(), what="bn fwd")
    close(rm.cpu(), bn.running_mean, what="running_mean")
    close(rvv.cpu(), bn.running_var, what="running_var")
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, tol=2e-4, what="bn dX")
    close(tape.param_grads[id(g)].cpu(), bn.weight.grad, tol=2e-4, what="dgamma")
    close(tape.param_grads[id(bta)].cpu(), bn.bias.grad, tol=2e-4, what="dbeta")
    if with_res:
        close(nchw(rv.grad), res.grad, what="dres")


@pytest.mark.parametrize("act,with_res", [(0, False), (2, False), (1, True)])
@pytest.mark.parametrize("shape", [(4, 18, 34, 96), (4, 16, 32, 960), (4, 32, 64, 192), (2, 23, 30, 256), (2, 9, 11, 100), (3, 1, 1, 256)])
def test_batchnorm_register_cached_variant_is_bit_identical(shape, act, with_res):
    """Maps small enough that a thread's rows fit in registers take the row-cached single-launch kernels (no second read of
    x / dy): same sums in the same order as the two-pass kernels, selected off with pp_debug_set_bn_bytes_per_block(-1)."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    B, H, W, C = shape
    torch.manual_seed(12)
    x = torch.randn(B, H, W, C, device=DEV) * 2 + 0.5
    res = torch.randn(B, H, W, C, device=DEV) if with_res else None
    dy = torch.randn(B, H, W, C, device=DEV)
    g0, b0 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3

    def run(cached):
        L.pp_debug_set_bn_bytes_per_block(0 if cached else -1)
        try:
            tape = E.Tape()
            xv = E.Var(x.clone())
            rv = E.Var(res.clone()) if with_res else None
            g, bta = gparam(g0), gparam(b0)
            rm, rvar = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            yv = E.batch_norm_act(tape, xv, g, bta, rm, rvar, True, act, rv)
            y = yv.t.clone()
            tape.backward(yv, dy.clone())
            return y, xv.grad.clone(), tape.param_grads[id(g)].clone(), tape.param_grads[id(bta)].clone(), rm, rvar, \
                (rv.grad.clone() if with_res else None)
        finally:
            L.pp_debug_set_bn_bytes_per_block(0)

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        if u is not None:
            assert torch.equal(u, v)


def _read_sync(sync):
    import ctypes
    probe = torch.empty(2, dtype=torch.int32, device=DEV)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    torch.cuda.synchronize()
    assert hip.hipMemcpy(probe.data_ptr(), sync.data_ptr(), 8, 3) == 0          # D2D
    e, d = probe.cpu().tolist()
    return e, d


def test_batchnorm_single_launch_exchange_is_coherent_deterministic_and_rearms():
    """pp_bn_train_fwd_fused / pp_bn_bwd_fused exchange partial sums between blocks on different XCDs through the
    fine-grained area of engine._bn_exchange.  With DIFFERENT data in every launch (a stale partial from the previous
    launch would show), 60 back-to-back launches agree with the three-launch path, a repeat of the whole sequence is
    bit-identical, and every launch advances the launch epoch once and leaves the done-counter at zero."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    dev = torch.device(DEV)
    sync, ws = E._bn_exchange(dev)
    st = torch.cuda.current_stream().cuda_stream
    for (M, C) in [(2048, 960), (8192, 192), (32768, 24), (100, 304), (4 * 64 * 128, 256)]:
        assert L.pp_bn_fused_workspace_bytes(M, C) <= ws.numel() and L.pp_bn_fused_sync_ints(C) <= sync.numel()
        gen = torch.Generator(device=DEV).manual_seed(9)
        xs = [torch.randn(M, C, device=DEV, generator=gen) * (1 + i % 3) + i for i in range(6)]
        dy = torch.randn(M, C, device=DEV, generator=gen)
        gamma, beta = torch.rand(C, device=DEV, generator=gen) + 0.5, torch.randn(C, device=DEV, generator=gen)
        ws3 = torch.empty(L.pp_colreduce_workspace_bytes(M, C), dtype=torch.uint8, device=DEV)

        epoch0, done0 = _read_sync(sync)
        assert done0 == 0

        def sequence():
            outs = []
            for it in range(60):
                x = xs[it % 6]
                mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
                y, dx = torch.empty_like(x), torch.empty_like(x)
                dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
                _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                                                   mean.data_ptr(), invstd.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(),
                                                   ws.numel(), sync.data_ptr(), sync.numel(), st), "fwd")
                _lib.check(L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, y.data_ptr(), C, 2, M, C, mean.data_ptr(),
                                             invstd.data_ptr(), gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), C,
                                             None, 0, 1.0, None, ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st), "bwd")
                outs.append((mean, invstd, dg, db, y if it % 20 == 0 else None, dx if it % 20 == 0 else None))
            return outs
        a, b = sequence(), sequence()
        torch.cuda.synchronize()
        for (ta, tb) in zip(a, b):
            for u, v in zip(ta, tb):
                assert (u is None and v is None) or torch.equal(u, v)
        # against the three-launch kernels (other summation grouping: not bitwise)
        for it in (0, 1, 2, 3, 4, 5, 59):
            x = xs[it % 6]
            mean3, inv3, sc, sf = (torch.empty(C, device=DEV) for _ in range(4))
            _lib.check(L.pp_bn_train_fwd(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                                         mean3.data_ptr(), inv3.data_ptr(), sc.data_ptr(), sf.data_ptr(), ws3.data_ptr(), ws3.numel(), st), "fwd3")
            close(a[it][0], mean3, tol=1e-5, what=f"mean it {it}")
            close(a[it][1], inv3, tol=1e-5, what=f"invstd it {it}")
        epoch1, done1 = _read_sync(sync)
        assert done1 == 0 and epoch1 - epoch0 == 2 * 60 * 2      # every launch advanced the epoch once and re-armed
        # too-small sync array / workspace are refused, not overrun
        x = xs[0]
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, invstd, dg, db = (torch.empty(C, device=DEV) for _ in range(4))
        assert L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                                       mean.data_ptr(), invstd.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(),
                                       ws.numel(), sync.data_ptr(), 1, st) != 0
        assert L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, y.data_ptr(), C, 2, M, C, mean.data_ptr(),
                                 invstd.data_ptr(), gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), C,
                                 None, 0, 1.0, None, ws.data_ptr(), 16, sync.data_ptr(), sync.numel(), st) != 0


@pytest.mark.parametrize("shape", [(4, 18, 34, 160, 960, 1, 0, 1), (4, 16, 32, 960, 160, 1, 0, 1), (4, 16, 32, 1280, 256, 1, 0, 1),
                                   (4, 16, 32, 320, 256, 3, 6, 6), (2, 64, 128, 304, 256, 3, 1, 1), (4, 64, 128, 24, 48, 1, 0, 1),
                                   (2, 128, 256, 3, 32, 3, 1, 1), (3, 23, 30, 64, 384, 1, 0, 1), (4, 32, 64, 256, 1024, 1, 0, 1)])
@pytest.mark.parametrize("act,with_res,drop", [(E.ACT_RELU6, False, 0.0), (E.ACT_NONE, True, 0.0), (E.ACT_RELU, False, 0.5)])
def test_conv_epilogue_statistics_feed_the_batchnorm(shape, act, with_res, drop, monkeypatch):
    """Dense convolution -> training BatchNorm with the statistics taken from the convolution's epilogue (pp_conv2d_fwd_stats:
    per-wave column sums, or the split-K reduce's per-block sums) and applied by pp_bn_train_fwd_partials, against the same
    pair through the self-contained single-launch BatchNorm: same convolution output bit for bit, statistics / outputs /
    running statistics / all gradients within 2e-5, identical dropout mask; every tile configuration and the split-K path."""
    B, H, W, Cin, Cout, k, pad, dil = shape
    if with_res and act != E.ACT_NONE:
        pytest.skip("combination not used")
    gen = torch.Generator(device=DEV).manual_seed(B * 100 + Cout)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=gen)
    w = (torch.randn(k, k, Cin, Cout, device=DEV, generator=gen) * (2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    gamma = (torch.rand(Cout, device=DEV, generator=gen) + 0.5).requires_grad_(True)
    beta = torch.randn(Cout, device=DEV, generator=gen).requires_grad_(True)
    Ho, Wo = E.out_size(H, k, 1, pad, dil), E.out_size(W, k, 1, pad, dil)
    res = torch.randn(B, Ho, Wo, Cout, device=DEV, generator=gen) if with_res else None
    dy = torch.randn(B, Ho, Wo, Cout, device=DEV, generator=gen)
    outs = {}
    monkeypatch.setattr(E, "_CONV_BN_FUSE", False)            # (the one-launch conv + BatchNorm path has its own test)
    for mode in (True, False):
        monkeypatch.setattr(E, "_CONV_BN_STATS", mode)
        monkeypatch.setattr(E, "_CONV_BN_STATS_MAX_ROWS", 1 << 30)
        E.set_dropout_seed(77)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        tape = E.Tape(True)
        xv = E.Var(x.clone())
        rvv = E.Var(res.clone()) if with_res else None
        cv = E.conv2d(tape, xv, w, None, 1, pad, dil)
        assert (cv._pending is not None) == mode
        yv = E.batch_norm_act(tape, cv, gamma, beta, rm, rv, True, act, rvv, dropout_p=drop)
        tape.backward(yv, dy.clone())
        torch.cuda.synchronize()
        outs[mode] = dict(conv=cv.t.clone(), y=yv.t.clone(), rm=rm, rv=rv, dx=xv.grad.clone(), dw=tape.param_grads[id(w)].clone(),
                          dg=tape.param_grads[id(gamma)].clone(), db=tape.param_grads[id(beta)].clone())
    a, b = outs[True], outs[False]
    if k == 1 and Cin >= 256 and B * Ho * Wo <= 4096 and Cout <= 512:
        # without statistics these layers run the in-block split-K kernel, with them the tiled kernel in one pass: same sums in
        # another order
        close(a["conv"], b["conv"], tol=2e-5, what="conv")
    else:
        assert torch.equal(a["conv"], b["conv"])                           # the epilogue does not change what is stored
    assert torch.equal(a["y"] == 0, b["y"] == 0) or drop == 0.0            # same dropout mask (same seed stream)
    for key, tol in (("y", 2e-5), ("rm", 2e-6), ("rv", 2e-6), ("dx", 5e-5), ("dw", 5e-5), ("dg", 5e-5), ("db", 5e-5)):
        close(a[key], b[key], tol=tol, what=key)
    rows = int(L().pp_conv2d_fwd_stats_rows(B, H, W, Cin, Cout, k, k, 1, pad, dil))
    assert rows > 0


def L():
    from pixelpick_amd import _lib
    return _lib.lib()


def test_two_single_launch_batchnorms_on_two_streams_do_not_interfere():
    """Two spin-waiting BatchNorm launches in flight at once (main + side stream, each with its own exchange area from
    engine._bn_exchange): every launch asks for at most half of the co-resident capacity (pp_bn_fused_capacity), so both
    become resident whatever the interleaving - no hang - and the results equal the one-stream results bit for bit.
    Shapes include the 33 / 51 MB maps whose grid is sized from bytes."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    dev = torch.device(DEV)
    cap = L.pp_bn_fused_capacity()
    assert cap >= 512, cap
    main = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    E.refresh_stream()
    ex_main = E._bn_exchange(dev)
    E._role[0] = 1
    try:
        ex_side = E._bn_exchange(dev)
    finally:
        E._role[0] = 0
    assert ex_main[0].data_ptr() != ex_side[0].data_ptr()
    gen = torch.Generator(device=DEV).manual_seed(3)
    shapes = [(4 * 64 * 128, 256), (4 * 130 * 258, 96), (2448, 960), (4 * 128 * 256, 32)]
    data = []
    for (M, C) in shapes:
        x = torch.randn(M, C, device=DEV, generator=gen)
        dy = torch.randn(M, C, device=DEV, generator=gen)
        gamma, beta = torch.rand(C, device=DEV, generator=gen) + 0.5, torch.randn(C, device=DEV, generator=gen)
        data.append((M, C, x, dy, gamma, beta))

    def run(item, ex, st):
        M, C, x, dy, gamma, beta = item
        sync, ws = ex
        assert L.pp_bn_fused_workspace_bytes(M, C) <= ws.numel()
        mean, invstd, dg, db = (torch.empty(C, device=DEV) for _ in range(4))
        y, dx = torch.empty_like(x), torch.empty_like(x)
        _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                                           mean.data_ptr(), invstd.data_ptr(), None, 0, 2, 0.0, 0, None, y.data_ptr(), C, ws.data_ptr(),
                                           ws.numel(), sync.data_ptr(), sync.numel(), st.cuda_stream), "fwd")
        _lib.check(L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, y.data_ptr(), C, 2, M, C, mean.data_ptr(),
                                     invstd.data_ptr(), gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), C,
                                     None, 0, 1.0, None, ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st.cuda_stream), "bwd")
        return mean, invstd, dg, db, y, dx

    ref = [run(it, ex_main, main) for it in data]
    torch.cuda.synchronize()
    side.wait_stream(main)
    outs_a, outs_b = [], []
    for rep in range(20):
        for i in range(len(data)):
            outs_a.append((i, run(data[i], ex_main, main)))
            outs_b.append(((i + 1) % len(data), run(data[(i + 1) % len(data)], ex_side, side)))
    main.wait_stream(side)
    torch.cuda.synchronize()                      # a deadlock between the two launches would hang here (pytest-timeout)
    for i, got in outs_a + outs_b:
        for u, v in zip(got, ref[i]):
            assert torch.equal(u, v)


@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("M,C", [(2048, 960), (33540, 96), (700, 24)])
def test_batchnorm_backward_mask_recomputed_from_x_is_bit_identical(M, C, act):
    """pp_bn_bwd_fused with y_act == NULL recomputes the ReLU/ReLU6 mask as act'(fma(x, scale, shift)) - the forward's
    own expression - instead of reading the saved output: every result bit-equal to the y_act path."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    dev = torch.device(DEV)
    sync, ws = E._bn_exchange(dev)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(M, C, device=DEV, generator=gen) * 2 + 0.7
    dy = torch.randn(M, C, device=DEV, generator=gen)
    gamma, beta = torch.rand(C, device=DEV, generator=gen) + 0.5, torch.randn(C, device=DEV, generator=gen) + 1.0
    mean, invstd, y = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty_like(x)
    _lib.check(L.pp_bn_train_fwd_fused(x.data_ptr(), C, M, C, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1, None, None,
                                       mean.data_ptr(), invstd.data_ptr(), None, 0, act, 0.0, 0, None, y.data_ptr(), C,
                                       ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st), "fwd")
    outs = []
    for ya, be in ((y, None), (None, beta)):
        dx, dg, db = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        _lib.check(L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, ya.data_ptr() if ya is not None else None, C, act, M, C,
                                     mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                     dx.data_ptr(), C, None, 0, 1.0, be.data_ptr() if be is not None else None,
                                     ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st), "bwd")
        outs.append((dx, dg, db))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    frac = (y == 0).float().mean().item()
    assert 0.02 < frac < 0.98            # the mask is not trivial
    # neither source -> refused
    assert L.pp_bn_bwd_fused(x.data_ptr(), C, dy.data_ptr(), C, None, C, act, M, C, mean.data_ptr(), invstd.data_ptr(),
                             gamma.data_ptr(), dg.data_ptr(), db.data_ptr(), dx.data_ptr(), C, None, 0, 1.0, None,
                             ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), st) != 0


@pytest.mark.parametrize("shape,p", [((4, 16, 32, 256), 0.5), ((2, 9, 11, 48), 0.1), ((2, 64, 128, 256), 0.5)])
def test_dropout_fused_into_batchnorm_equals_separate_dropout(shape, p):
    """BN -> ReLU -> nn.Dropout (aspp.py:59-61, decoders.py:108-114): the dropout rides in the single-launch BatchNorm
    (same mask stream as pp_dropout), its backward is the 1/(1-p) factor on the masked gradient."""
    B, H, W, C = shape
    torch.manual_seed(4)
    x = torch.randn(B, H, W, C, device=DEV) * 2 + 0.3
    dy = torch.randn(B, H, W, C, device=DEV)
    gamma, beta = gparam(torch.rand(C) + 0.5), gparam(torch.randn(C) * 0.2)

    def run(fused):
        E.set_dropout_seed(77)
        tape = E.Tape()
        xv = E.Var(x.clone())
        if fused:
            yv = E.batch_norm_act(tape, xv, gamma, beta, None, None, True, E.ACT_RELU, dropout_p=p)
        else:
            yv = E.dropout(tape, E.batch_norm_act(tape, xv, gamma, beta, None, None, True, E.ACT_RELU), p, True)
        n_launch_nodes = len(tape.nodes)
        tape.backward(yv, dy.clone())
        return yv.t, xv.grad, tape.param_grads[id(gamma)], tape.param_grads[id(beta)], n_launch_nodes

    yf, dxf, dgf, dbf, nf = run(True)
    ys, dxs, dgs, dbs, ns = run(False)
    assert nf == 1 and ns == 2
    assert torch.equal(yf, ys)
    frac = (yf == 0).float().mean().item()
    assert frac > p * 0.5                                   # relu zeros + dropped
    close(dxf, dxs, tol=1e-5, what="dx"); close(dgf, dgs, tol=1e-5, what="dgamma"); close(dbf, dbs, tol=1e-5, what="dbeta")


def test_batchnorm_eval():
    torch.manual_seed(3)
    B, C, H, W = 2, 48, 5, 7
    x = torch.randn(B, C, H, W)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C) + 0.5); bn.bias.copy_(torch.randn(C))
        bn.running_mean.copy_(torch.randn(C)); bn.running_var.copy_(torch.rand(C) + 0.5)
    yv = E.batch_norm_act(E.Tape(False), E.Var(nhwc(x)), bn.weight.detach().to(DEV), bn.bias.detach().to(DEV),
                          bn.running_mean.to(DEV), bn.running_var.to(DEV), False, E.ACT_RELU6)
    close(nchw(yv.t), F.relu6(bn(x)).detach(), what="bn eval")


def test_pad_and_crop_accumulate():
    torch.manual_seed(4)
    x = torch.randn(2, 8, 5, 6, requires_grad=True)
    yr = F.pad(x, (2, 2, 2, 2))
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    yv = E.pad2d(tape, xv, 2, 2)
    close(nchw(yv.t), yr.detach(), what="pad")
    extra = torch.randn(2, 8, 5, 6)
    xv.grad = nhwc(extra)                         # a gradient already present (residual branch)
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad + extra, what="crop+add")


BIL_CASES = [((2, 16, 32, 256), (64, 128), True, 0.0), ((2, 23, 30, 16), (90, 120), True, 0.0),
             ((2, 8, 12, 128), (16, 24), False, 2.0), ((1, 9, 7, 256), (18, 14), False, 0.0),
             ((2, 5, 6, 8), (11, 17), False, 0.0), ((2, 1, 1, 16), (4, 6), True, 0.0),
             ((2, 2, 2, 8), (4, 4), False, 2.0), ((1, 3, 2, 4), (6, 4), False, 0.0), ((1, 1, 5, 8), (2, 10), False, 0.0),
             ((2, 6, 5, 12), (24, 20), False, 0.0), ((1, 7, 9, 8), (25, 31), True, 0.0), ((2, 4, 4, 4), (12, 16), False, 0.0)]   # separable backward


@pytest.mark.parametrize("case", BIL_CASES, ids=[str(c) for c in BIL_CASES])
def test_bilinear_nhwc(case):
    (B, H, W, C), size, align, sf = case
    torch.manual_seed(5)
    x = torch.randn(B, C, H, W, requires_grad=True)
    if sf > 0:
        yr = F.interpolate(x, scale_factor=sf, mode="bilinear", align_corners=align)
    else:
        yr = F.interpolate(x, size=size, mode="bilinear", align_corners=align)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    yv = E.bilinear(tape, xv, size, align, sf)
    close(nchw(yv.t), yr.detach(), tol=2e-5, what="bilinear fwd")
    tape.backward(yv, nhwc(dy))
    close(nchw(xv.grad), x.grad, tol=2e-5, what="bilinear bwd")


@pytest.mark.parametrize("C,size_in,size_out", [(19, (16, 32), (64, 128)), (11, (23, 30), (90, 120)), (21, (9, 13), (33, 50))])
def test_bilinear_to_nchw(C, size_in, size_out):
    torch.manual_seed(6)
    x = torch.randn(2, C, *size_in, requires_grad=True)
    yr = F.interpolate(x, size=size_out, mode="bilinear", align_corners=True)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    yv = E.bilinear(tape, xv, size_out, True, 0.0, out_nchw=True)
    assert yv.t.shape == yr.shape
    close(yv.t.cpu(), yr.detach(), tol=2e-5, what="bilinear->nchw fwd")
    tape.backward(yv, dy.to(DEV))
    close(nchw(xv.grad), x.grad, tol=2e-5, what="bilinear->nchw bwd")


def test_gap_and_broadcast():
    torch.manual_seed(7)
    x = torch.randn(3, 320, 16, 32, requires_grad=True)
    yr = F.adaptive_avg_pool2d(x, 1)
    zr = F.interpolate(yr, size=(16, 32), mode="bilinear", align_corners=True)
    dz = torch.randn_like(zr)
    zr.backward(dz)
    tape = E.Tape()
    xv = E.Var(nhwc(x.detach()))
    yv = E.global_avg_pool(tape, xv)
    zv = E.broadcast_hw(tape, yv, 16, 32)
    close(nchw(yv.t), yr.detach(), what="gap")
    close(nchw(zv.t), zr.detach(), what="broadcast")
    tape.backward(zv, nhwc(dz))
    close(nchw(xv.grad), x.grad, what="gap/broadcast bwd")


def test_dropout_statistics_and_backward():
    x = torch.ones(2, 32, 64, 256, device=DEV)
    tape = E.Tape()
    xv = E.Var(x)
    yv = E.dropout(tape, xv, 0.5, True)
    y = yv.t
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01
    assert torch.all((y == 0) | (y == 2.0))
    dy = torch.full_like(y, 3.0)
    tape.backward(yv, dy)
    assert torch.equal(xv.grad != 0, y != 0) and torch.all((xv.grad == 0) | (xv.grad == 6.0))
    y2 = E.dropout(E.Tape(False), E.Var(x), 0.5, True).t
    assert not torch.equal(y2, y)                      # new mask every call
    assert E.dropout(E.Tape(False), xv, 0.5, False) is xv   # eval: identity


def test_dropout2d_drops_whole_channels_and_backward_uses_the_same_mask():
    """nn.Dropout2d (mobilenet_v2.py:114-115,133-134): one draw per (sample, channel); kept channels are scaled by 1/(1-p);
    the backward pass applies the identical mask to the gradient; eval mode / p = 0 is the identity."""
    torch.manual_seed(0)
    B, H, W, C, p = 6, 9, 11, 320, 0.2
    x = torch.randn(B, H, W, C, device=DEV) + 3.0            # no zeros in the input
    E.set_dropout_seed(123)
    tape = E.Tape(True)
    xv = E.Var(x)
    yv = E.dropout2d(tape, xv, p, True)
    y = yv.t
    per_map = (y != 0).reshape(B, H * W, C)
    assert (per_map.all(dim=1) | (~per_map).all(dim=1)).all(), "a channel map must be kept or dropped as a whole"
    kept = per_map[:, 0, :]
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - p) < 0.04, frac                        # 1920 draws: sigma = 0.009
    assert not torch.equal(kept[0], kept[1])                 # samples draw independently
    torch.testing.assert_close(y[kept[:, None, None, :].expand_as(y)], (x / (1 - p))[kept[:, None, None, :].expand_as(y)])
    dy = torch.randn_like(x)
    tape.backward(yv, dy)
    dx = xv.grad
    assert torch.equal(dx != 0, y != 0)
    torch.testing.assert_close(dx[y != 0], (dy / (1 - p))[y != 0])
    assert E.dropout2d(E.Tape(False), xv, p, False) is xv and E.dropout2d(E.Tape(False), xv, 0.0, True) is xv


def test_mc_accumulate_matches_torch_softmax_and_scores():
    """pp_acq_softmax_sum (query.py:181-187): mean probability and mean per-pass score over T stochastic passes."""
    from pixelpick_amd import acquisition as acq
    torch.manual_seed(1)
    T, C, H, W = 5, 21, 13, 17
    logits = torch.randn(T + 3, C, H, W + 2, device=DEV)[:T, :, :, :W] * 3       # strided view
    prob = torch.softmax(logits, dim=1)
    for st, ref_uc in (("entropy", (-prob * prob.log()).sum(1)), ("least_confidence", 1 - prob.max(1)[0]),
                       ("margin_sampling", (prob.topk(2, dim=1).values[:, 0] - prob.topk(2, dim=1).values[:, 1]).abs())):
        p_out = torch.empty(C, H, W, device=DEV)
        u_out = torch.empty(H, W, device=DEV)
        acq.mc_accumulate_(logits[:2], p_out, u_out, st, 1.0 / T, accumulate=False)
        acq.mc_accumulate_(logits[2:], p_out, u_out, st, 1.0 / T, accumulate=True)
        torch.testing.assert_close(p_out, prob.mean(0), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(u_out, ref_uc.mean(0), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,C,H,W,n_lab,ign", [(4, 19, 64, 128, 20, 19), (2, 21, 33, 47, 10, 255), (1, 11, 20, 24, 480, 11)])
def test_cross_entropy(B, C, H, W, n_lab, ign):
    torch.manual_seed(8)
    logits = (torch.randn(B, C, H, W) * 3).requires_grad_(True)
    y = torch.full((B, H, W), ign, dtype=torch.int64)
    for b in range(B):
        idx = torch.randperm(H * W)[:n_lab]
        y[b].view(-1)[idx] = torch.randint(0, C, (n_lab,))
    lr = F.cross_entropy(logits, y, ignore_index=ign)
    lr.backward()
    loss, dl = E.cross_entropy_nchw(logits.detach().to(DEV), y.to(DEV), ign)
    assert abs(loss.item() - lr.item()) < 1e-5 * max(1.0, abs(lr.item()))
    close(dl.cpu(), logits.grad, tol=1e-5, what="dlogits")
    assert (dl.cpu()[(y == ign)[:, None].expand(-1, C, -1, -1)] == 0).all()


LOWRES_CE_CASES = [
    # B, C, (h,w), (H,W), labelled px/img, ignore_index, align_corners
    (4, 19, (64, 128), (256, 512), 20, 19, True),      # the BASELINE train step
    (2, 21, (20, 20), (80, 80), 10, 255, True),        # VOC ignore 255
    (2, 11, (23, 30), (90, 120), 100, 11, True),       # CamVid, ragged ratio
    (1, 19, (9, 13), (33, 47), 33 * 47, 19, True),     # EVERY pixel labelled (borders, corners, clamped taps)
    (2, 26, (8, 8), (32, 32), 64, 26, True),           # generic C <= 32
    (1, 40, (8, 12), (16, 24), 50, 40, True),          # generic C <= 64
    (2, 19, (16, 32), (64, 128), 40, 19, False),       # align_corners=False arithmetic
    (1, 19, (16, 16), (16, 16), 30, 19, True),         # identity size
]


@pytest.mark.parametrize("B,C,lo,size,n_lab,ign,align", LOWRES_CE_CASES)
def test_cross_entropy_from_lowres_logits(B, C, lo, size, n_lab, ign, align):
    """deeplab.py:55-56 + model.py:116 in the sparse kernels vs torch autograd through F.interpolate + F.cross_entropy,
    and vs the dense product path (pp_bilinear_fwd -> pp_sparse_ce_fwd_bwd -> pp_bilinear_bwd)."""
    torch.manual_seed(8)
    H, W = size
    low = (torch.randn(B, C, *lo) * 3).requires_grad_(True)
    y = torch.full((B, H, W), ign, dtype=torch.int64)
    for b in range(B):
        idx = torch.randperm(H * W)[:n_lab]
        y[b].view(-1)[idx] = torch.randint(0, C, (len(idx),))
    lr = F.cross_entropy(F.interpolate(low, size=size, mode="bilinear", align_corners=align), y, ignore_index=ign)
    lr.backward()
    wide = torch.full((B, *lo, C + 3), 5.0, device=DEV)                   # channel slice of a wider buffer (ldx > C)
    wide[..., :C] = low.detach().permute(0, 2, 3, 1).to(DEV)
    low_d = wide[..., :C]
    loss, dlow = E.cross_entropy_lowres(low_d, size, y.to(DEV), ign, align_corners=align)
    assert abs(loss.item() - lr.item()) < 1e-5 * max(1.0, abs(lr.item()))
    close(dlow.permute(0, 3, 1, 2).cpu(), low.grad, tol=2e-5, what="dlow")
    # dense product path
    tape = E.Tape(True)
    lv = E.Var(low_d.contiguous())
    pred = E.bilinear(tape, lv, size, align, 0.0, out_nchw=True)
    loss2, dl = E.cross_entropy_nchw(pred.t, y.to(DEV), ign)
    tape.backward(pred, dl)
    assert abs(loss.item() - loss2.item()) < 2e-6 * max(1.0, abs(loss2.item()))
    close(dlow, lv.grad, tol=1e-5, what="dlow vs dense path")
    # fixed-order gather: bitwise reproducible
    loss3, dlow3 = E.cross_entropy_lowres(low_d, size, y.to(DEV), ign, align_corners=align)
    assert torch.equal(loss3, loss) and torch.equal(dlow3, dlow)


def test_cross_entropy_from_lowres_logits_without_labels_is_nan():
    low = torch.randn(1, 8, 8, 19, device=DEV)
    y = torch.full((1, 32, 32), 19, dtype=torch.int64, device=DEV)
    loss, _ = E.cross_entropy_lowres(low, (32, 32), y, 19)
    assert torch.isnan(loss).all()                     # 0/0 like F.cross_entropy


def test_adam_matches_torch():
    torch.manual_seed(9)
    n, n_split = 10007, 4001
    p0 = torch.randn(n)
    pa = p0[:n_split].clone().requires_grad_(True)
    pb = p0[n_split:].clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [pa], "lr": 5e-5, "weight_decay": 2e-4},
                            {"params": [pb], "lr": 5e-4, "weight_decay": 2e-4}], betas=(0.9, 0.999), eps=1e-7)
    from pixelpick_amd import _lib
    L = _lib.lib()
    p = p0.clone().to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        g = torch.randn(n) * 0.1
        pa.grad, pb.grad = g[:n_split].clone(), g[n_split:].clone()
        opt.step()
        gg = g.to(DEV)
        rc = L.pp_adam_step_flat(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), n, n_split, 5e-5, 5e-4, 0.9, 0.999,
                                 1e-7, 2e-4, step, 1.0, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = torch.cat([pa.detach(), pb.detach()])
        assert (p.cpu() - ref).abs().max().item() < 2e-6


def test_sgd_momentum_matches_torch_and_trainer_uses_it():
    """pp_sgd_step_flat == torch.optim.SGD(momentum 0.9, weight decay) with the reference's voc groups (utils/utils.py:
    208-240: lr 1e-3 backbone, 1e-2 rest, weight decay 5e-4); FlatTrainer(optimizer="sgd") drives it."""
    torch.manual_seed(10)
    n, n_split = 10007, 4001
    p0 = torch.randn(n)
    pa = p0[:n_split].clone().requires_grad_(True)
    pb = p0[n_split:].clone().requires_grad_(True)
    opt = torch.optim.SGD([{"params": [pa], "lr": 1e-3, "weight_decay": 5e-4, "momentum": 0.9},
                           {"params": [pb], "lr": 1e-2, "weight_decay": 5e-4, "momentum": 0.9}])
    from pixelpick_amd import _lib
    L = _lib.lib()
    p = p0.clone().to(DEV)
    buf = torch.full((n,), 123.0, device=DEV)             # garbage: step 1 must overwrite, not read it
    for step in range(1, 5):
        g = torch.randn(n) * 0.1
        pa.grad, pb.grad = g[:n_split].clone(), g[n_split:].clone()
        opt.step()
        gg = (g * 2).to(DEV)                              # grad_scale 0.5 (the 1/world of a 2-rank all-reduce)
        rc = L.pp_sgd_step_flat(p.data_ptr(), gg.data_ptr(), buf.data_ptr(), n, n_split, 1e-3, 1e-2, 0.9, 5e-4, step, 0.5, None,
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        ref = torch.cat([pa.detach(), pb.detach()])
        assert (p.cpu() - ref).abs().max().item() < 2e-6
    assert L.pp_sgd_step_flat(p.data_ptr(), gg.data_ptr(), None, n, n_split, 1e-3, 1e-2, 0.9, 5e-4, 1, 1.0, None,
                              torch.cuda.current_stream().cuda_stream) != 0


# ---- LDS-DMA convolution kernels (csrc/conv_igemm.hip: conv_igemm_dma_kernel, conv_wgrad_dma_kernel) vs the register-staged ones
DMA_CONV_SHAPES = [  # B, H, W, Cin, Cout, k, stride, pad, dil
    (4, 64, 128, 304, 256, 3, 1, 1, 1),     # SegmentHead conv 1 (decoders.py:107): 128x128 tiles, ragged Cin
    (2, 64, 128, 256, 256, 3, 1, 1, 1),     # SegmentHead conv 2
    (4, 16, 32, 160, 960, 1, 1, 0, 1),      # MobileNetV2 expand: 64x64 tiles, ragged Cout tile in backward-data, split-K
    (4, 16, 32, 1280, 256, 1, 1, 0, 1),     # ASPP fuse (aspp.py:73): split-K
    (4, 16, 32, 320, 256, 3, 1, 12, 12),    # ASPP atrous branch: dead taps
    (3, 37, 53, 132, 260, 3, 1, 1, 1),      # ragged rows, ragged channel chunks (132 = 8*16 + 4)
    (2, 45, 61, 136, 192, 3, 2, 1, 1),      # stride 2: backward-data stays on the register-staged kernel
    (2, 32, 64, 64, 64, 3, 1, 1, 1),        # one 64x64 tile column
    (1, 9, 11, 72, 68, 1, 1, 0, 1),         # tiny: fewer K steps than pipeline stages
]


@pytest.mark.parametrize("shape", DMA_CONV_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_lds_dma_conv_kernels_match_register_staged_kernels(shape):
    """Same tiles and MFMA order, so forward, backward-data and weight gradient must agree bit for bit (to 1e-4 relative
    where the register-staged plan uses the 64-deep K step, whose summation order differs)."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    B, H, W, Ci, Co, k, st, pad, dil = shape
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Ci, device=DEV)
    w = (torch.randn(k, k, Ci, Co, device=DEV) * 0.05).requires_grad_(True)
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // st + 1, (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    dy = torch.randn(B, Ho, Wo, Co, device=DEV)

    def run(variant):
        L.pp_debug_set_conv_variant(variant)
        tape = E.Tape(True)
        xv = E.Var(x)
        y = E.conv2d(tape, xv, w, None, st, pad, dil)
        tape.backward(y, dy)
        torch.cuda.synchronize()
        return y.t.clone(), xv.grad.clone(), tape.param_grads[id(w)].clone()
    try:
        ref = run(256 | 262144 | (1 << 20))                 # every LDS-DMA kernel off
        for variant in (0, 2 << 20, 32768):                 # the default mix; + 64x64 weight gradients; backward-data on fewer layers
            got = run(variant)
            for name, a, b in zip(("y", "dx", "dw"), got, ref):
                if torch.equal(a, b):
                    continue
                close(a, b, tol=1e-4, what=f"{name} (variant {variant})")
                assert max(Ci, Co) >= 256 and B * H * W >= 4096, f"{name}: only the 64-deep-K baseline may differ in rounding"
    finally:
        L.pp_debug_set_conv_variant(0)


@pytest.mark.parametrize("shape,stride,pad,dil,act,res", [
    ((4, 16, 32, 960), 1, 1, 1, E.ACT_RELU6, False),      # 1/16-resolution MobileNetV2 block (mobilenet_v2.py:38-40)
    ((2, 33, 47, 96), 2, 1, 1, E.ACT_RELU6, False),       # stride 2, ragged size
    ((2, 18, 34, 192), 1, 0, 1, E.ACT_RELU6, False),      # pre-padded input, padding 0 (fixed_padding, mobilenet_v2.py:15-21)
    ((2, 20, 36, 64), 1, 2, 2, E.ACT_RELU6, False),       # dilation 2 (features.17)
    ((1, 9, 11, 8), 1, 1, 1, E.ACT_NONE, True),           # tiny, with a residual
    ((4, 128, 256, 32), 1, 1, 1, E.ACT_RELU6, False),     # the largest map of the network
])
def test_dwconv_fused_into_training_batchnorm_is_bit_identical(monkeypatch, bn_fused, shape, stride, pad, dil, act, res):
    """pp_dwconv3x3_bn_train_fwd_fused (the depthwise convolution computed inside the single-launch BatchNorm) against the
    two launches it replaces: outputs, saved statistics, running statistics and every gradient must agree bit for bit."""
    B, H, W, C = shape
    torch.manual_seed(4)
    x = torch.randn(B, H, W, C, device=DEV)
    w = (torch.randn(3, 3, C, device=DEV) * 0.3).requires_grad_(True)
    gamma = (torch.rand(C, device=DEV) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=DEV) * 0.1).requires_grad_(True)
    Ho, Wo = (H + 2 * pad - 2 * dil - 1) // stride + 1, (W + 2 * pad - 2 * dil - 1) // stride + 1
    r = torch.randn(B, Ho, Wo, C, device=DEV) if res else None
    dy = torch.randn(B, Ho, Wo, C, device=DEV)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(E, "_FUSE_DW_BN", fuse)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        tape = E.Tape(True)
        xv = E.Var(x.clone())
        rvar = E.Var(r.clone()) if res else None
        d = E.dwconv3x3(tape, xv, w, stride, pad, dil)
        assert (d._pending is not None) == fuse
        y = E.batch_norm_act(tape, d, gamma, beta, rm, rv, True, act, rvar)
        assert d._pending is None
        tape.backward(y, dy.clone())
        torch.cuda.synchronize()
        outs.append((y.t.clone(), d.t.clone(), rm, rv, xv.grad.clone(), tape.param_grads[id(w)].clone(),
                     tape.param_grads[id(gamma)].clone(), tape.param_grads[id(beta)].clone()) + ((rvar.grad.clone(),) if res else ()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
