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
                # encoder.py:28 + module_helper.py:86-107: a torchvision-layout ResNet file at a path relative to scripts/;
                # PIXELPICK_RESNET_WEIGHTS overrides the path.  A missing file raises, as in the reference (module_helper.py:95)
                path = os.environ.get("PIXELPICK_RESNET_WEIGHTS", "") or resnet[n_layers]
                if not os.path.isfile(path):
                    raise FileNotFoundError(f"{path} not exists. (weight_type='supervised'; set PIXELPICK_RESNET_WEIGHTS or use "
                                            f"weight_type='random')")
                self.base = ResNetBackbone(backbone=f'resnet{n_layers}_dilated8', pretrained=path)
                print("Encoder initialised with supervised weights.")
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
