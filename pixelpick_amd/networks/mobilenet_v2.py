"""MobileNetV2 encoder (output_stride 16) — mirror of networks/mobilenet_v2.py on the HIP engine.

Same constructor arguments, attribute names (`features`, `low_level_features`, `high_level_features`)
and state_dict keys.  Parity-critical quirk kept: every InvertedResidual zero-pads its
INPUT (fixed_padding, mobilenet_v2.py:15-21,61) and runs the expand conv + BN + ReLU6 on the padded map,
then a depthwise 3x3 with padding 0.
"""
import os
import warnings

import torch
import torch.nn as nn

from .. import engine as E
from .layers import BatchNorm2d, Conv2d, Dropout2d, ReLU6


FOLD_FIXED_PADDING = os.environ.get("PIXELPICK_FOLD_PAD", "1") != "0"


def conv_bn(inp, oup, stride, BatchNorm):
    """mobilenet_v2.py:7-12."""
    return nn.Sequential(Conv2d(inp, oup, 3, stride, 1, bias=False), BatchNorm(oup), ReLU6(inplace=True))


def fixed_padding_amounts(kernel_size, dilation):
    """mobilenet_v2.py:15-21 -> (pad_beg, pad_end)."""
    kernel_size_effective = kernel_size + (kernel_size - 1) * (dilation - 1)
    pad_total = kernel_size_effective - 1
    pad_beg = pad_total // 2
    return pad_beg, pad_total - pad_beg


def _run_conv_bn_act_chain(tape, seq, x, residual=None, first_pad=0, out_consumers=0):
    """Execute an nn.Sequential of [Conv2d, BatchNorm2d, (ReLU6)] groups; the last BN may take a residual.
    first_pad: zero padding of x folded into the first convolution.  out_consumers: how many ops read the chain's output, the first
    of them a dense convolution (BatchNorm2d.run `consumers`); 0: unknown."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        conv, bn = mods[i], mods[i + 1]
        has_act = i + 2 < len(mods) and isinstance(mods[i + 2], ReLU6)
        is_last = (i + (3 if has_act else 2)) >= len(mods)
        x = conv.run(tape, x, extra_pad=first_pad if i == 0 else 0)
        # an inner BatchNorm's only consumer is the next convolution of the chain: when that one can apply scale / shift /
        # activation where it loads its input, the BatchNorm is split over its neighbours (engine._bn_on_load)
        nxt = i + (3 if has_act else 2)
        lazy_ok = (not is_last) and tape.enabled and mods[nxt].accepts_lazy_input(E.shape_of(x))
        x = bn.run(tape, x, E.ACT_RELU6 if has_act else E.ACT_NONE, residual if is_last else None, lazy_ok=lazy_ok,
                   single_consumer=not is_last, consumers=out_consumers if is_last else 0)
        i = nxt
    return x


class InvertedResidual(nn.Module):
    """mobilenet_v2.py:24-66."""

    def __init__(self, inp, oup, stride, dilation, expand_ratio, BatchNorm):
        super().__init__()
        self.stride = stride
        assert stride in [1, 2]
        hidden_dim = round(inp * expand_ratio)
        self.use_res_connect = self.stride == 1 and inp == oup
        self.kernel_size = 3
        self.dilation = dilation
        layers = []
        if expand_ratio != 1:
            layers += [Conv2d(inp, hidden_dim, 1, 1, 0, 1, bias=False), BatchNorm(hidden_dim), ReLU6(inplace=True)]
        layers += [Conv2d(hidden_dim, hidden_dim, 3, stride, 0, dilation, groups=hidden_dim, bias=False),
                   BatchNorm(hidden_dim), ReLU6(inplace=True),
                   Conv2d(hidden_dim, oup, 1, 1, 0, 1, bias=False), BatchNorm(oup)]
        self.conv = nn.Sequential(*layers)

    def takes_input_through_a_dense_conv(self):
        """The block's input is read by its expand convolution first (with the fixed padding folded into it), then - in a residual
        block - by the add behind the project BatchNorm: the producer of that input may be told so (`out_consumers`)."""
        pad_beg, pad_end = fixed_padding_amounts(self.kernel_size, self.dilation)
        return FOLD_FIXED_PADDING and pad_beg == pad_end and not self.conv[0].depthwise

    def run(self, tape, x, out_consumers=0):
        pad_beg, pad_end = fixed_padding_amounts(self.kernel_size, self.dilation)
        res = x if self.use_res_connect else None
        if FOLD_FIXED_PADDING and pad_beg == pad_end:
            # F.pad(x) followed by a bias-free 1x1 conv (or the depthwise 3x3 of the t=1 block) is that convolution
            # with padding=pad: the border outputs are the same zeros, the BN statistics over the padded map too,
            # and neither the padded copy (forward) nor its cropped gradient (backward) is materialised.
            return _run_conv_bn_act_chain(tape, self.conv, x, residual=res, first_pad=pad_beg, out_consumers=out_consumers)
        x_pad = E.pad2d(tape, x, pad_beg, pad_end)
        return _run_conv_bn_act_chain(tape, self.conv, x_pad, residual=res, out_consumers=out_consumers)


class MobileNetV2(nn.Module):
    """mobilenet_v2.py:69-155.  `pretrained=True` in the reference downloads ImageNet weights from a URL
    (:140); there is no network here, so weights are loaded from $PIXELPICK_MNV2_WEIGHTS when that file
    exists and left at their random initialisation otherwise."""

    def __init__(self, output_stride=8, BatchNorm=None, width_mult=1., pretrained=True, mc_dropout=False, mc_dropout_p=0.2):
        super().__init__()
        BatchNorm = BatchNorm or BatchNorm2d
        block = InvertedResidual
        input_channel = 32
        current_stride = 1
        rate = 1
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
        input_channel = int(input_channel * width_mult)
        features = [conv_bn(3, input_channel, 2, BatchNorm)]
        current_stride *= 2
        for t, c, n, s in setting:
            if current_stride == output_stride:
                stride, dilation = 1, rate
                rate *= s
            else:
                stride, dilation = s, 1
                current_stride *= s
            output_channel = int(c * width_mult)
            for i in range(n):
                features.append(block(input_channel, output_channel, stride if i == 0 else 1, dilation, t, BatchNorm))
                input_channel = output_channel
        if mc_dropout:                                   # mobilenet_v2.py:114-115: last feature, for MC train
            features.append(Dropout2d(p=mc_dropout_p))
        self.features = nn.Sequential(*features)
        if pretrained:
            self._load_pretrained_model()
        self.low_level_features = self.features[0:4]
        self.high_level_features = self.features[4:]
        self.dropout = Dropout2d(p=mc_dropout_p)         # mobilenet_v2.py:127: on the low-level features, for MC test
        self.mc_dropout = mc_dropout

    def _load_pretrained_model(self):
        """mobilenet_v2.py:139-147: the reference downloads `mobilenet_v2-6a65762b.pth` (torchvision key layout: `features.N...`
        plus the ImageNet head `features.18.*`, `classifier.1.*`) and copies every entry whose key exists in its own
        state_dict.  There is no network here: PIXELPICK_MNV2_WEIGHTS names a local copy of that file.  Starting from a RANDOM
        backbone is never silent: it must be asked for (args.weight_type == "random" -> DeepLab passes pretrained=False, or
        PIXELPICK_MNV2_WEIGHTS=random); a missing file raises."""
        path = os.environ.get("PIXELPICK_MNV2_WEIGHTS", "")
        if path == "random":
            return
        if not path:
            raise FileNotFoundError("MobileNetV2 ImageNet weights: the reference downloads them (mobilenet_v2.py:140), which is "
                                    "impossible offline - set PIXELPICK_MNV2_WEIGHTS to a local mobilenet_v2-6a65762b.pth, or ask "
                                    "for a random backbone explicitly (args.weight_type='random' / PIXELPICK_MNV2_WEIGHTS=random)")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"PIXELPICK_MNV2_WEIGHTS={path}: no such file")
        pretrain_dict = torch.load(path, map_location="cpu", weights_only=True)
        state_dict = self.state_dict()
        state_dict.update({k: v for k, v in pretrain_dict.items() if k in state_dict})      # mobilenet_v2.py:142-146
        self.load_state_dict(state_dict)

    # features[LATE_FEATURES_FROM:] (the 96- / 160- / 320-channel blocks) hold 87 % of the encoder's parameters, and when backward()
    # gets back to this point ~1.5 ms of encoder backward are still ahead: data-parallel training starts their gradients'
    # all-reduce here (Tape.mark "encoder_late_done", trainer.FlatTrainer), so that only the first 0.9 MB wait for the join
    LATE_FEATURES_FROM = 11

    def late_modules(self):
        """The modules whose parameter gradients are complete when backward() reaches Tape.mark("encoder_late_done")."""
        return list(self.features)[self.LATE_FEATURES_FROM:]

    @staticmethod
    def _run_features(tape, seq, x, mark_before=None):
        mods = list(seq)
        for j, m in enumerate(mods):
            if mark_before is not None and j == mark_before:
                tape.mark("encoder_late_done")
            if isinstance(m, InvertedResidual):
                # the block behind (inside THIS Sequential: the last block's output leaves it) reads the output through its expand
                # convolution first, then - a residual block - through its add: its backward-data takes this block's project
                # BatchNorm backward along (engine._conv2d_bwd)
                nxt = mods[j + 1] if j + 1 < len(mods) else None
                nc = 0
                if tape.enabled and isinstance(nxt, InvertedResidual) and nxt.takes_input_through_a_dense_conv():
                    nc = 2 if nxt.use_res_connect else 1
                x = m.run(tape, x, out_consumers=nc)
            elif isinstance(m, Dropout2d):
                x = m.run(tape, x)
            else:  # stem conv_bn; its only consumer is the depthwise conv of the t=1 block behind it (no residual there)
                nxt = mods[j + 1] if j + 1 < len(mods) else None
                lazy_ok = (tape.enabled and isinstance(nxt, InvertedResidual) and not nxt.use_res_connect and nxt.conv[0].depthwise
                           and FOLD_FIXED_PADDING)
                x = m[1].run(tape, m[0].run(tape, x), E.ACT_RELU6, lazy_ok=lazy_ok)
        return x

    def run(self, tape, x):
        low = self._run_features(tape, self.low_level_features, x)
        n_low = len(self.low_level_features)
        high = self._run_features(tape, self.high_level_features, low,
                                  mark_before=self.LATE_FEATURES_FROM - n_low if self.LATE_FEATURES_FROM > n_low else None)
        if self.mc_dropout:                              # mobilenet_v2.py:133-134
            low = self.dropout.run(tape, low)
        return high, low
