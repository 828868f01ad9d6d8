"""Encoder — mirror of networks/encoder.py:18-56."""
import os

import torch.nn as nn

from .backbones.resnet_backbone import ResNetBackbone

resnet = {18: "../networks/backbones/pretrained/resnet18-pytorch.pth", 34: "../networks/backbones/pretrained/resnet34-pytorch.pth",
          50: "../networks/backbones/pretrained/resnet50-pytorch.pth", 101: "../networks/backbones/pretrained/resnet101-pytorch.pth"}


class Encoder(nn.Module):
    def __init__(self, args, load_pretrained):
        super().__init__()
        weight_type = args.weight_type
        n_layers = args.n_layers
        if load_pretrained:
            if weight_type == "supervised":
                path = resnet[n_layers]
                self.base = ResNetBackbone(backbone=f'resnet{n_layers}_dilated8', pretrained=path if os.path.isfile(path) else None)
                print("Encoder initialised with supervised weights." if os.path.isfile(path) else
                      f"Encoder: {path} not found, random initialisation.")
            else:
                self.base = ResNetBackbone(backbone='resnet50_dilated8', pretrained=None)
        else:
            self.base = ResNetBackbone(backbone=f'resnet{n_layers}_dilated8', pretrained=None,
                                       width_multiplier=args.width_multiplier)
        self.weight_type = weight_type
        self.use_fpn = args.use_dilated_resnet

    def get_backbone_params(self):
        return self.base.parameters()

    def run(self, tape, x):
        return self.base.run(tape, x)
