"""Dilated ResNet backbone — mirror of networks/backbones/resnet_models.py (Bottleneck/ResNet) and
resnet_backbone.py (DilatedResnetBackbone, ResNetBackbone) for the one architecture the reference's
configs select: `resnet{50,101}_dilated8` (encoder.py:28-33).  Same attribute names / state_dict keys
(`prefix.conv1`, `prefix.bn1`, `layerN.M.conv1..bn3`, `layerN.0.downsample.0/1`).
"""
import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from ... import engine as E
from ..layers import BatchNorm2d, Conv2d, ReLU


class Bottleneck(nn.Module):
    """resnet_models.py:58-94."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, tape, x):
        out = self.bn1.run(tape, self.conv1.run(tape, x), E.ACT_RELU)
        out = self.bn2.run(tape, self.conv2.run(tape, out), E.ACT_RELU)
        residual = x
        if self.downsample is not None:
            residual = self.downsample[1].run(tape, self.downsample[0].run(tape, x))
        # out = relu(bn3(conv3(out)) + residual)   (resnet_models.py:85-92)
        return self.bn3.run(tape, self.conv3.run(tape, out), E.ACT_RELU, residual=residual)


class ResNet(nn.Module):
    """resnet_models.py:97-170 (non-deep-base stem; avgpool/fc are never used by the backbone and are omitted —
    their keys are stripped by the reference as well when it wraps the model, resnet_backbone.py:62-67)."""

    def __init__(self, block, layers, width_multiplier=1.0):
        super().__init__()
        self.inplanes = int(64 * width_multiplier)
        self.prefix = nn.Sequential(OrderedDict([
            ('conv1', Conv2d(3, self.inplanes, 7, stride=2, padding=3, bias=False)),
            ('bn1', BatchNorm2d(self.inplanes)),
            ('relu', ReLU(inplace=False))]))
        self.maxpool = nn.Identity()   # nn.MaxPool2d(3, 2, 1): no parameters; executed by E.max_pool2d
        self.layer1 = self._make_layer(block, int(64 * width_multiplier), layers[0])
        self.layer2 = self._make_layer(block, int(128 * width_multiplier), layers[1], stride=2)
        self.layer3 = self._make_layer(block, int(256 * width_multiplier), layers[2], stride=2)
        self.layer4 = self._make_layer(block, int(512 * width_multiplier), layers[3], stride=2)
        for m in self.modules():                       # resnet_models.py:131-137
            if isinstance(m, Conv2d):
                n = m.kernel_size * m.kernel_size * m.out_channels
                with torch.no_grad():
                    m.weight.normal_(0, math.sqrt(2. / n))

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                                       BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


class DilatedResnetBackbone(nn.Module):
    """resnet_backbone.py:46-104: strides of layer3/4 removed and replaced by dilation (output stride 8)."""

    def __init__(self, orig_resnet, dilate_scale=8, multi_grid=None):
        super().__init__()
        self.num_features = 2048
        if dilate_scale == 8:
            orig_resnet.layer3.apply(partial(self._nostride_dilate, dilate=2))
            if multi_grid is None:
                orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=4))
            else:
                for i, r in enumerate(multi_grid):
                    orig_resnet.layer4[i].apply(partial(self._nostride_dilate, dilate=int(4 * r)))
        elif dilate_scale == 16:
            if multi_grid is None:
                orig_resnet.layer4.apply(partial(self._nostride_dilate, dilate=2))
            else:
                for i, r in enumerate(multi_grid):
                    orig_resnet.layer4[i].apply(partial(self._nostride_dilate, dilate=int(2 * r)))
        self.prefix = orig_resnet.prefix
        self.maxpool = orig_resnet.maxpool
        self.layer1, self.layer2 = orig_resnet.layer1, orig_resnet.layer2
        self.layer3, self.layer4 = orig_resnet.layer3, orig_resnet.layer4

    @staticmethod
    def _nostride_dilate(m, dilate):
        if isinstance(m, Conv2d):
            if m.stride == 2:
                m.stride = 1
                if m.kernel_size == 3:
                    m.dilation = dilate // 2
                    m.padding = dilate // 2
            elif m.kernel_size == 3:
                m.dilation = dilate
                m.padding = dilate

    def get_num_features(self):
        return self.num_features

    def run(self, tape, x):
        feats = []
        x = self.prefix.bn1.run(tape, self.prefix.conv1.run(tape, x), E.ACT_RELU)
        x = E.max_pool2d(tape, x, 3, 2, 1)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            if layer is self.layer3:
                # layer3 + layer4 hold 22 of ResNet50's 23.5 M parameters; when backward() is back here, layer2 / layer1 / the stem are
                # still ahead: data-parallel training starts the all-reduce of their gradients now (trainer.FlatTrainer)
                tape.mark("encoder_late_done")
            for blk in layer:
                x = blk.run(tape, x)
            feats.append(x)
        return feats

    def late_modules(self):
        """The modules whose parameter gradients are complete when backward() reaches Tape.mark("encoder_late_done")."""
        return [self.layer3, self.layer4]


def ResNetBackbone(backbone=None, width_multiplier=1.0, pretrained=None, multi_grid=None, norm_type='batchnorm'):
    """resnet_backbone.py:107-192, restricted to the dilated Bottleneck variants the configs use."""
    depths = {'resnet50_dilated8': [3, 4, 6, 3], 'resnet101_dilated8': [3, 4, 23, 3]}
    if backbone not in depths:
        raise Exception('Architecture undefined!' if backbone is None else
                        f"{backbone}: only resnet50_dilated8 / resnet101_dilated8 are built (the reference's configs select no other)")
    orig = ResNet(Bottleneck, depths[backbone], width_multiplier=width_multiplier)
    if pretrained is not None:
        sd = torch.load(pretrained, map_location="cpu", weights_only=True)
        own = orig.state_dict()
        mapped = {}
        for k, v in sd.items():                     # module_helper.py:102-106 load_model: conv1/bn1 live under prefix.*
            k2 = "prefix." + k if "prefix." + k in own else k
            if k2 in own:
                mapped[k2] = v                      # (the ImageNet head fc.* has no counterpart: this ResNet keeps no fc)
        missing = [k for k in own if k not in mapped and not k.endswith("num_batches_tracked")]
        if missing:
            raise KeyError(f"{pretrained}: {len(missing)} backbone tensors missing from the file, e.g. {missing[:4]}")
        orig.load_state_dict(mapped, strict=False)
    return DilatedResnetBackbone(orig, dilate_scale=8, multi_grid=multi_grid)
