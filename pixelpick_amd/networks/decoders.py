"""SegmentHead — mirror of networks/decoders.py:104-131 on the HIP engine."""
import torch
import torch.nn as nn

from .. import engine as E
from .layers import BatchNorm2d, Conv2d, Dropout, ReLU


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
        h = s[3].run(tape, s[1].run(tape, s[0].run(tape, x), E.ACT_RELU))
        emb = s[7].run(tape, s[5].run(tape, s[4].run(tape, h), E.ACT_RELU))
        pred = self.classifier.run(tape, emb)
        return {"emb": emb, "pred": pred}
