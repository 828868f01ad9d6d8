"""ASPP — mirror of networks/aspp.py on the HIP engine.

The five branches write straight into their channel slice of one [B,H,W,1280] buffer (the reference's
torch.cat, aspp.py:73, costs nothing) and the image-pooling branch's bilinear upsample of a 1x1 map
(aspp.py:69-70) is a broadcast.
"""
import os

import torch
import torch.nn as nn

from .. import engine as E
from .layers import BatchNorm2d, Conv2d, Dropout, ReLU

_ASPP_MERGE = os.environ.get("PIXELPICK_ASPP_MERGE", "1") != "0"


class _ASPPModule(nn.Module):
    """aspp.py:6-29."""

    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm):
        super().__init__()
        self.atrous_conv = Conv2d(inplanes, planes, kernel_size, stride=1, padding=padding, dilation=dilation, bias=False)
        self.bn = BatchNorm(planes)
        self.relu = ReLU()

    def run(self, tape, x, dst=None, bwd_group=None):
        return self.bn.run(tape, self.atrous_conv.run(tape, x, bwd_group=bwd_group), E.ACT_RELU, dst=dst)


class ASPP(nn.Module):
    """aspp.py:32-88."""

    def __init__(self, backbone, output_stride, BatchNorm=None):
        super().__init__()
        BatchNorm = BatchNorm or BatchNorm2d
        inplanes = {"drn": 512, "mobilenet": 320}.get(backbone, 2048)
        if output_stride == 16:
            dilations = [1, 6, 12, 18]
        elif output_stride == 8:
            dilations = [1, 12, 24, 36]
        else:
            raise NotImplementedError
        self.aspp1 = _ASPPModule(inplanes, 256, 1, padding=0, dilation=dilations[0], BatchNorm=BatchNorm)
        self.aspp2 = _ASPPModule(inplanes, 256, 3, padding=dilations[1], dilation=dilations[1], BatchNorm=BatchNorm)
        self.aspp3 = _ASPPModule(inplanes, 256, 3, padding=dilations[2], dilation=dilations[2], BatchNorm=BatchNorm)
        self.aspp4 = _ASPPModule(inplanes, 256, 3, padding=dilations[3], dilation=dilations[3], BatchNorm=BatchNorm)
        # nn.Sequential(AdaptiveAvgPool2d, Conv2d, BatchNorm, ReLU): indices 1 and 2 carry the parameters
        self.global_avg_pool = nn.Sequential(nn.Identity(), Conv2d(inplanes, 256, 1, stride=1, bias=False), BatchNorm(256), ReLU())
        self.conv1 = Conv2d(1280, 256, 1, bias=False)
        self.bn1 = BatchNorm(256)
        self.relu = ReLU()
        self.dropout = Dropout(0.5)

    def run(self, tape, x):
        B, H, W, _ = x.t.shape
        buf = torch.empty((B, H, W, 1280), dtype=torch.float32, device=x.t.device)
        # training: the four branches read one input, so its gradient is ONE backward-data launch over (branch, tap, channel)
        # (engine.ConvBwdGroup / pp_conv2d_bwd_data_multi) instead of four launches + three split-K reduces + three adds;
        # PIXELPICK_ASPP_MERGE=0: the per-layer form
        grp = None
        if tape.enabled and self.training and x.needs_grad and _ASPP_MERGE:
            branches = (self.aspp1, self.aspp2, self.aspp3, self.aspp4)
            specs = [(m.atrous_conv.weight, m.atrous_conv.kernel_size, m.atrous_conv.dilation) for m in branches]
            same = all(m.atrous_conv.stride == 1 and m.atrous_conv.bias is None and m.atrous_conv.weight.requires_grad and
                       m.atrous_conv.padding == m.atrous_conv.dilation * (m.atrous_conv.kernel_size - 1) // 2 for m in branches)
            nws = E.ConvBwdGroup.offered(x, specs) if same else 0
            if nws:
                grp = E.ConvBwdGroup(x, specs, nws)
        g = (lambda i: (grp, i)) if grp is not None else (lambda i: None)
        x1 = self.aspp1.run(tape, x, dst=buf[..., 0:256], bwd_group=g(0))
        x2 = self.aspp2.run(tape, x, dst=buf[..., 256:512], bwd_group=g(1))
        x3 = self.aspp3.run(tape, x, dst=buf[..., 512:768], bwd_group=g(2))
        x4 = self.aspp4.run(tape, x, dst=buf[..., 768:1024], bwd_group=g(3))
        pooled = E.global_avg_pool(tape, x)
        x5 = self.global_avg_pool[2].run(tape, self.global_avg_pool[1].run(tape, pooled), E.ACT_RELU)
        x5 = E.broadcast_hw(tape, x5, H, W, dst=buf[..., 1024:1280])
        cat = E.concat_alias(tape, buf, [x1, x2, x3, x4, x5])
        return self.bn1.run(tape, self.conv1.run(tape, cat), E.ACT_RELU, dropout=self.dropout)


def build_aspp(backbone, output_stride, BatchNorm=None):
    return ASPP(backbone, output_stride, BatchNorm)
