"""Decoders — mirror of networks/decoders.py on the HIP engine: FPNDecoder / UpsampleBlock (:6-101) and
SegmentHead (:104-131)."""
import math

import torch
import torch.nn as nn

from .. import engine as E
from .layers import BatchNorm2d, Conv2d, Dropout, ReLU


class GroupNorm(nn.Module):
    """nn.GroupNorm(num_groups, num_channels) parameters; executed fused with the following ReLU."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def run(self, tape, x, relu=True):
        return E.group_norm_relu(tape, x, self.weight, self.bias, self.num_groups, relu, self.eps)

    def forward(self, x):
        raise RuntimeError("pixelpick_amd layers execute through run(tape, x)")


class UpsampleBlock(nn.Module):
    """decoders.py:87-101: 3x3 conv (bias) + GroupNorm(32) + ReLU, then bilinear x2 (align_corners=False)."""

    def __init__(self, in_channels, out_channels=128, kernel_size=3, padding=1, n_groups=32, scale_factor=2):
        super().__init__()
        self.block = nn.Sequential(Conv2d(in_channels, out_channels, kernel_size, padding=padding),
                                   GroupNorm(n_groups, out_channels), ReLU(inplace=True))
        self.scale_factor = scale_factor

    def run(self, tape, x, upsample=True):
        h = self.block[1].run(tape, self.block[0].run(tape, x), relu=True)
        if not upsample:                 # FPNDecoder.run(lowres=True): the x2 interpolation of the LAST block of a branch is applied later
            return h
        _, H, W, _ = h.t.shape
        return E.bilinear(tape, h, (int(math.floor(H * self.scale_factor)), int(math.floor(W * self.scale_factor))), False,
                          float(self.scale_factor))


class FPNDecoder(nn.Module):
    """decoders.py:6-84."""

    def __init__(self, args):
        super().__init__()
        n_classes, wm = args.n_classes, args.width_multiplier
        if args.n_layers in [18, 34]:
            chans = [int(512 * wm), int(256 * wm), int(128 * wm), int(64 * wm)]
        elif args.n_layers in [50, 101]:
            chans = [int(2048 * wm), int(1024 * wm), int(512 * wm), int(256 * wm)]
        else:
            raise ValueError(args.n_layers)
        self.lat_layer_0 = Conv2d(chans[0], 256, 1)
        self.lat_layer_1 = Conv2d(chans[1], 256, 1)
        self.lat_layer_2 = Conv2d(chans[2], 256, 1)
        self.lat_layer_3 = Conv2d(chans[3], 256, 1)
        self.upsample_blocks_0 = nn.Sequential(UpsampleBlock(256, 128), UpsampleBlock(128, 128), UpsampleBlock(128, 128))
        self.upsample_blocks_1 = nn.Sequential(UpsampleBlock(256, 128), UpsampleBlock(128, 128), UpsampleBlock(128, 128))
        self.upsample_blocks_2 = nn.Sequential(UpsampleBlock(256, 128), UpsampleBlock(128, 128), UpsampleBlock(128, 128))
        self.upsample_blocks_3 = nn.Sequential(UpsampleBlock(256, 128), UpsampleBlock(128, 128))
        self.classifier = Conv2d(128, n_classes, 1)
        for m in self.modules():                       # decoders.py:84-86: kaiming_normal_(nonlinearity='relu') on every conv
            if isinstance(m, Conv2d):
                fan_in = m.in_channels * m.kernel_size * m.kernel_size
                with torch.no_grad():
                    m.weight.normal_(0, math.sqrt(2.0 / fan_in))

    @staticmethod
    def _upsample_add(tape, x, y):
        _, h, w, _ = y.t.shape
        return E.add(tape, E.bilinear(tape, x, (h, w), False, 0.0), y)

    @staticmethod
    def _seq(tape, seq, x, last_upsample=True):
        n = len(seq)
        for i, blk in enumerate(seq):
            x = blk.run(tape, x, upsample=last_upsample or i + 1 < n)
        return x

    def run(self, tape, feats, lowres=False):
        """lowres (training with sparse labels, trainer.FlatTrainer): the four branches all end in the SAME x2 bilinear interpolation
        (decoders.py:101), the branch sum (:79) and the 1x1 classifier (:81) are linear and the interpolation weights sum to one, so
        pred = classifier(sum_i up2(q_i)) = up2(classifier(sum_i q_i)) - identical in exact arithmetic.  Returns the classifier
        output at HALF resolution ("pred") and no "emb"; the loss interpolates it at the labelled pixels only
        (engine.cross_entropy_lowres, align_corners False), so the four 268 MB full-resolution maps of a 4 x 256 x 512 batch, their three
        sums, the full-resolution classifier and logits never exist.  forward() / acquisition keep the reference's operation order."""
        c2, c3, c4, c5 = feats
        c5 = self.lat_layer_0.run(tape, c5)
        c4 = self.lat_layer_1.run(tape, c4)
        c3 = self.lat_layer_2.run(tape, c3)
        c2 = self.lat_layer_3.run(tape, c2)
        p5 = c5
        p4 = self._upsample_add(tape, p5, c4)
        p3 = self._upsample_add(tape, p4, c3)
        p2 = self._upsample_add(tape, p3, c2)
        p5 = self._seq(tape, self.upsample_blocks_0, p5, not lowres)
        p4 = self._seq(tape, self.upsample_blocks_1, p4, not lowres)
        p3 = self._seq(tape, self.upsample_blocks_2, p3, not lowres)
        p2 = self._seq(tape, self.upsample_blocks_3, p2, not lowres)
        emb = E.add(tape, E.add(tape, E.add(tape, p2, p3), p4), p5)
        if lowres:
            return {"emb": None, "pred": self.classifier.run(tape, emb)}
        return {"emb": emb, "pred": self.classifier.run(tape, emb)}


class SegmentHead(nn.Module):
    def __init__(self, args, openset=False):
        super().__init__()
        self.segment_head = nn.Sequential(Conv2d(304, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                          BatchNorm2d(256), ReLU(), Dropout(0.5),
                                          Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False),
                                          BatchNorm2d(256), ReLU(), Dropout(args.mc_dropout_p))
        self.classifier = Conv2d(256, args.n_classes, 1)
        self.n_classes = args.n_classes

    def run(self, tape, x):
        s = self.segment_head
        h = s[1].run(tape, s[0].run(tape, x), E.ACT_RELU, dropout=s[3])
        emb = s[5].run(tape, s[4].run(tape, h), E.ACT_RELU, dropout=s[7])
        pred = self.classifier.run(tape, emb)
        return {"emb": emb, "pred": pred}
