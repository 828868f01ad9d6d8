"""GPU parity of network BLOCKS (InvertedResidual incl. the padded-border quirk, ASPP, decoder) against a
plain PyTorch fp32 CPU restatement built from torch.nn.functional ops with the same weights."""
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from pixelpick_amd import engine as E
from pixelpick_amd.networks.aspp import ASPP
from pixelpick_amd.networks.layers import BatchNorm2d, Dropout
from pixelpick_amd.networks.mobilenet_v2 import InvertedResidual

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-12)


def _check(name, got, ref32, ref_noisy, ref64, tol=1e-3):
    """Flip-tolerant comparison against the torch restatement evaluated three ways: fp32, fp32 with a 1e-6
    input perturbation, and fp64.  A ReLU/ReLU6 unit whose pre-activation sits within an fp32 ulp of the
    threshold takes different branches in equally valid evaluations (measured: torch fp32 vs torch fp64 ASPP
    at 23x30 differ on 17 % of a weight gradient's elements at rel. L2 4e-3 because of ONE such unit), so each
    element is compared with the closest of the three references: per-element tolerance 1e-3 of the tensor
    scale; at most 1 % of the elements may exceed it and the relative L2 error must stay below 3e-2.  A wrong
    kernel moves most elements away from all three."""
    g = got.double()
    refs = [ref32.double(), ref_noisy.double(), ref64.double()]
    scale = refs[2].abs().max().item()
    err = torch.stack([(g - r).abs() for r in refs]).min(dim=0).values
    frac_bad = (err > tol * scale).double().mean().item()
    l2 = min((g - r).norm().item() / max(r.norm().item(), 1e-30) for r in refs)
    assert frac_bad <= 0.01 and l2 <= 3e-2, f"{name}: {100 * frac_bad:.2f}% elements out of tolerance, rel L2 {l2:.2e}, max err {err.max().item():.3e}"


def _formula(mod):
    mod.load_state_dict(fi.formula_state_dict(mod.state_dict()))
    for m in mod.modules():
        if isinstance(m, Dropout):
            m.p = 0.0
    return mod


def t_bn(x, sd, prefix, training=True):
    rm, rv = sd[prefix + "running_mean"].clone(), sd[prefix + "running_var"].clone()
    return F.batch_norm(x, rm, rv, sd[prefix + "weight"], sd[prefix + "bias"], training, 0.1, 1e-5)


def _grads_oihw(mod, tape):
    out = {}
    for k, p in mod.named_parameters():
        g = tape.param_grads[id(p)]
        if g.dim() == 4:
            g = g.permute(3, 2, 0, 1)
        elif g.dim() == 3:
            g = g.permute(2, 0, 1).unsqueeze(1)
        out[k] = g.cpu()
    return out


@pytest.mark.parametrize("cfg", [(32, 16, 1, 1, 1), (24, 24, 1, 1, 6), (24, 32, 2, 1, 6), (160, 320, 1, 2, 6), (16, 24, 2, 1, 6)],
                         ids=["t1", "res", "s2", "dil2", "s2-odd"])
def test_inverted_residual(cfg):
    inp, oup, stride, dil, t = cfg
    blk = _formula(InvertedResidual(inp, oup, stride, dil, t, BatchNorm2d)).to(DEV).train()
    B, H, W = 2, 13, 18
    x0 = fi.fill((B, inp, H, W), "xb", -1, 1)

    def ref(xin, dt=torch.float32):                 # torch restatement of mobilenet_v2.py:60-66
        sd = {k: (v.detach().cpu().clone().to(dt) if v.dtype.is_floating_point else v.detach().cpu().clone())
              .requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in blk.state_dict().items()}
        x = xin.clone().to(dt).requires_grad_(True)
        h = F.pad(x, (dil, dil, dil, dil))
        i = 0
        if t != 1:
            h = F.relu6(t_bn(F.conv2d(h, sd["conv.0.weight"]), sd, "conv.1."))
            i = 3
        h = F.relu6(t_bn(F.conv2d(h, sd[f"conv.{i}.weight"], None, stride, 0, dil, groups=h.shape[1]), sd, f"conv.{i+1}."))
        h = t_bn(F.conv2d(h, sd[f"conv.{i+3}.weight"]), sd, f"conv.{i+4}.")
        yr = x + h if (stride == 1 and inp == oup) else h
        dy = fi.fill(tuple(yr.shape), "dyb", -1, 1)
        yr.backward(dy.to(dt))
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
        grads["input"] = x.grad
        return yr.detach(), dy, grads

    yr, dy, gr = ref(x0)
    yn, _, gn = ref(x0 * (1 + 1e-6 * fi.fill(tuple(x0.shape), "nz", -1, 1)))
    yd, _, gd = ref(x0, torch.float64)
    tape = E.Tape()
    xv = E.Var(nhwc(x0))
    yv = blk.run(tape, xv)
    _check("output", nchw(yv.t), yr, yn, yd)
    tape.backward(yv, nhwc(dy))
    got = _grads_oihw(blk, tape)
    got["input"] = nchw(xv.grad)
    for k in gr:
        _check(k, got[k], gr[k], gn[k], gd[k])


@pytest.mark.parametrize("hw", [(4, 6), (16, 32), (23, 30)])
def test_aspp(hw):
    H, W = hw
    aspp = _formula(ASPP("mobilenet", 16, BatchNorm2d)).to(DEV).train()
    B = 2
    x0 = fi.fill((B, 320, H, W), "xa", -1, 1)

    def ref(xin, dt=torch.float32):                 # torch restatement of aspp.py:64-79
        sd = {k: (v.detach().cpu().clone().to(dt) if v.dtype.is_floating_point else v.detach().cpu().clone())
              .requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in aspp.state_dict().items()}
        x = xin.clone().to(dt).requires_grad_(True)
        brs = [F.relu(t_bn(F.conv2d(x, sd["aspp1.atrous_conv.weight"]), sd, "aspp1.bn."))]
        for n, d in (("aspp2", 6), ("aspp3", 12), ("aspp4", 18)):
            brs.append(F.relu(t_bn(F.conv2d(x, sd[f"{n}.atrous_conv.weight"], None, 1, d, d), sd, f"{n}.bn.")))
        p = F.adaptive_avg_pool2d(x, 1)
        p = F.relu(t_bn(F.conv2d(p, sd["global_avg_pool.1.weight"]), sd, "global_avg_pool.2."))
        brs.append(F.interpolate(p, size=(H, W), mode="bilinear", align_corners=True))
        yr = F.relu(t_bn(F.conv2d(torch.cat(brs, dim=1), sd["conv1.weight"]), sd, "bn1."))
        dy = fi.fill(tuple(yr.shape), "dya", -1, 1)
        yr.backward(dy.to(dt))
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad}
        grads["input"] = x.grad
        return yr.detach(), dy, grads

    yr, dy, gr = ref(x0)
    yn, _, gn = ref(x0 * (1 + 1e-6 * fi.fill(tuple(x0.shape), "nz", -1, 1)))
    yd, _, gd = ref(x0, torch.float64)
    tape = E.Tape()
    xv = E.Var(nhwc(x0))
    yv = aspp.run(tape, xv)
    _check("output", nchw(yv.t), yr, yn, yd)
    tape.backward(yv, nhwc(dy))
    got = _grads_oihw(aspp, tape)
    got["input"] = nchw(xv.grad)
    for k in gr:
        _check(k, got[k], gr[k], gn[k], gd[k])


@pytest.mark.parametrize("backbone,stride,hw", [("mobilenet", 16, (16, 32)), ("mobilenet", 16, (23, 30)), ("resnet", 8, (16, 24))])
def test_aspp_merged_backward_equals_the_per_layer_form(backbone, stride, hw, monkeypatch):
    """aspp.py:49-57,64-67: the four branches read one input.  Its gradient as ONE backward-data launch over (branch, tap, channel)
    (engine.ConvBwdGroup / pp_conv2d_bwd_data_multi, the branch BatchNorms' backward writing straight into the merged operand) must
    equal the per-layer form - four backward-data launches, their split-K reduces and the adds - up to fp32 summation order; every
    parameter gradient is the same launch in both forms and must not move a bit."""
    from pixelpick_amd.networks import aspp as aspp_mod
    H, W = hw
    mod = _formula(ASPP(backbone, stride, BatchNorm2d)).to(DEV).train()
    # parameters in ONE buffer, as trainer.FlatTrainer holds them (the merged launch addresses the branch weights from the lowest pointer)
    ps = list(mod.parameters())
    flat = torch.empty(sum((p.numel() + 3) // 4 * 4 for p in ps), device=DEV)
    off = 0
    for p in ps:
        flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
        p.data = flat[off:off + p.numel()].view_as(p)
        off += (p.numel() + 3) // 4 * 4
    cin = mod.aspp1.atrous_conv.in_channels
    x0 = fi.fill((2, cin, H, W), "xm", -1, 1)
    dy = fi.fill((2, 256, H, W), "dym", -1, 1)
    res = {}
    for merged in (False, True):
        monkeypatch.setattr(aspp_mod, "_ASPP_MERGE", merged)
        tape = E.Tape()
        xv = E.Var(nhwc(x0))
        if merged:
            specs = [(m.atrous_conv.weight, m.atrous_conv.kernel_size, m.atrous_conv.dilation) for m in (mod.aspp1, mod.aspp2, mod.aspp3, mod.aspp4)]
            assert E.ConvBwdGroup.offered(xv, specs) > 0, "the merged form is not offered for this shape: the test would compare the per-layer form with itself"
        yv = mod.run(tape, xv)
        tape.backward(yv, nhwc(dy))
        torch.cuda.synchronize()
        res[merged] = (nchw(yv.t), nchw(xv.grad), _grads_oihw(mod, tape))
    assert torch.equal(res[True][0], res[False][0])
    assert rel(res[True][1], res[False][1]) <= 2e-5
    for k in res[False][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), k


def test_conv_bwd_data_multi_op():
    """pp_conv2d_bwd_data_multi against the sum of pp_conv2d_bwd_data results: branch gradients as channel slices of one buffer,
    dead taps of a dilation larger than the map, accumulate into an existing gradient, weights at arbitrary distances."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(5)
    B, H, W, Cin, Cout = 2, 16, 24, 64, 32
    specs = [(1, 1), (3, 6), (3, 12), (3, 18)]
    ws_ = [torch.randn(k, k, Cin, Cout, device=DEV) * 0.1 for k, _ in specs]
    dbuf = torch.randn(B, H, W, 4 * Cout + 8, device=DEV)[..., :4 * Cout]          # a slice of a wider buffer: ld != n * Cout
    nws = int(L.pp_conv2d_bwd_data_multi_workspace_bytes(B, H, W, Cin, Cout, 4, 1, 1, 3, 6, 3, 12, 3, 18))
    assert nws > 0
    ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
    ref = torch.zeros(B, H, W, Cin, device=DEV)
    for b, ((k, d), w) in enumerate(zip(specs, ws_)):
        dyb = dbuf[..., b * Cout:(b + 1) * Cout].contiguous()
        dxb = torch.empty(B, H, W, Cin, device=DEV)
        w1 = torch.empty(int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, k, k, 1, d * (k - 1) // 2, d)) + 256, dtype=torch.uint8, device=DEV)
        _lib.check(L.pp_conv2d_bwd_data(dyb.data_ptr(), Cout, B, H, W, Cout, w.data_ptr(), k, k, 1, d * (k - 1) // 2, d, dxb.data_ptr(), Cin, H, W, Cin, 0,
                                        w1.data_ptr(), w1.numel(), st), "ref")
        ref += dxb
    for acc in (0, 1):
        dx = torch.full((B, H, W, Cin), 0.5, device=DEV)
        args = []
        for (k, d), w in zip(specs, ws_):
            args += [w.data_ptr(), k, d]
        _lib.check(L.pp_conv2d_bwd_data_multi(dbuf.data_ptr(), dbuf.stride(2), B, H, W, Cout, 4, *args, dx.data_ptr(), Cin, Cin, acc, ws.data_ptr(), ws.numel(), st), "multi")
        assert rel(dx.cpu(), (ref + 0.5 * acc).cpu()) <= 2e-5, acc
    # two branches only, and the refusals
    assert int(L.pp_conv2d_bwd_data_multi_workspace_bytes(B, H, W, Cin, Cout, 1, 3, 6, 0, 0, 0, 0, 0, 0)) == 0          # one branch: nothing to merge
    assert int(L.pp_conv2d_bwd_data_multi_workspace_bytes(B, H, W, Cin, Cout, 2, 2, 1, 3, 6, 0, 0, 0, 0)) == 0          # even kernel size: no "same" padding
    dx2 = torch.empty(B, H, W, Cin, device=DEV)
    _lib.check(L.pp_conv2d_bwd_data_multi(dbuf.data_ptr(), dbuf.stride(2), B, H, W, Cout, 2, ws_[0].data_ptr(), 1, 1, ws_[1].data_ptr(), 3, 6, None, 0, 0, None, 0, 0,
                                          dx2.data_ptr(), Cin, Cin, 0, ws.data_ptr(), ws.numel(), st), "multi2")
    ref2 = torch.zeros(B, H, W, Cin, device=DEV)
    for b in (0, 1):
        k, d = specs[b]
        dyb = dbuf[..., b * Cout:(b + 1) * Cout].contiguous()
        dxb = torch.empty(B, H, W, Cin, device=DEV)
        w1 = torch.empty(int(L.pp_conv2d_bwd_data_workspace_bytes(B, H, W, Cin, Cout, k, k, 1, d * (k - 1) // 2, d)) + 256, dtype=torch.uint8, device=DEV)
        _lib.check(L.pp_conv2d_bwd_data(dyb.data_ptr(), Cout, B, H, W, Cout, ws_[b].data_ptr(), k, k, 1, d * (k - 1) // 2, d, dxb.data_ptr(), Cin, H, W, Cin, 0,
                                        w1.data_ptr(), w1.numel(), st), "ref2")
        ref2 += dxb
    assert rel(dx2.cpu(), ref2.cpu()) <= 2e-5
