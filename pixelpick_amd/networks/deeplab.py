"""DeepLabv3+ (MobileNetV2) — mirror of networks/deeplab.py on the HIP engine.

`forward(inputs[B,3,H,W]) -> {"pred": logits [B,C,H,W], "emb": ...}` like deeplab.py:43-61, so
model.py:113-121, query.py:190 and eval loops drive it unchanged; `.backbone / .aspp / .low_level_conv /
.seg_head` feed the optimiser groups of utils/utils.py:125-141.  The whole network is ONE autograd node:
its forward records the HIP kernels on an engine tape and its backward replays the matching backward
kernels, so `loss.backward()` with any torch loss/optimizer works, while pixelpick_amd.trainer bypasses
torch.autograd entirely.  "emb" (deeplab.py:58-59, a 134 MB/image dead output no caller reads) is
computed lazily on first access.
"""
import torch
import torch.nn as nn

from .. import engine as E
from .aspp import ASPP
from .decoders import SegmentHead
from .layers import BatchNorm2d, Conv2d, Dropout, ReLU
from .mobilenet_v2 import MobileNetV2


class _LazyOutputs(dict):
    """dict_outputs whose "emb" entry is materialised on first access."""

    def __init__(self, pred, emb_low, size):
        super().__init__(pred=pred)
        self._emb_low, self._size = emb_low, size

    def _emb(self):
        if not dict.__contains__(self, "emb"):
            v = E.bilinear(E.Tape(False), E.Var(self._emb_low), self._size, True, 0.0, out_nchw=True)
            dict.__setitem__(self, "emb", v.t)
        return dict.__getitem__(self, "emb")

    def __getitem__(self, k):
        return self._emb() if k == "emb" else dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "emb" or dict.__contains__(self, k)

    def keys(self):
        return ["pred", "emb"]


class _NetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = tape recording, backward = tape replay."""

    @staticmethod
    def forward(ctx, inputs, model, *params):
        tape = E.Tape(enabled=True)
        pred, emb_low = model._run(tape, inputs)
        ctx.tape, ctx.pred_var, ctx.params = tape, pred, params
        ctx.model = model
        model._last_emb_low = emb_low.t
        return pred.t

    @staticmethod
    def backward(ctx, dpred):
        tape = ctx.tape
        tape.backward(ctx.pred_var, dpred.contiguous())
        grads = tuple(tape.param_grads.get(id(p)) if p.requires_grad else None for p in ctx.params)
        ctx.tape = None
        return (None, None) + grads


class DeepLab(nn.Module):
    """backbone='mobilenet' is the reference's model (deeplab.py:19 hard-wires MobileNetV2).  backbone='resnet' is the EXTRA
    of SURVEY.md 0.1: a DeepLabv3+-ResNet50 assembled from reference PARTS - `ResNetBackbone('resnet50_dilated8')`
    (resnet_backbone.py:141-144; c2 = low-level feature, c5 at 1/8 = high-level), `ASPP('resnet', 8)` (aspp.py:38-39,43-44:
    2048 input channels, rates 1/12/24/36), `SegmentHead` - wired as deeplab.py:43-59 wires its MobileNetV2.  The reference
    never instantiates it (its "ResNet50" model is FPNSeg), so it is pinned per component and against the same assembly of
    the imported reference parts (tools/gen_golden_net.py --r50)."""
    LOWRES_LOGITS = True      # _run(..., upsample=False) stops in front of the final x4 bilinear (deeplab.py:55-56)
    LOWRES_ALIGN_CORNERS = True

    def __init__(self, args, backbone='mobilenet', output_stride=16):
        super().__init__()
        self._resnet = backbone == 'resnet'
        if self._resnet:
            from .backbones.resnet_backbone import ResNetBackbone
            if output_stride != 8:
                raise NotImplementedError("the dilated ResNet backbone has output stride 8 (resnet_backbone.py:53-58)")
            self.backbone = ResNetBackbone(backbone='resnet50_dilated8', pretrained=None)
            low_level_inplanes = 256
        else:
            # deeplab.py:19 always starts from the ImageNet backbone (MobileNetV2(pretrained=True) downloads it); here the
            # file comes from PIXELPICK_MNV2_WEIGHTS and a random backbone has to be asked for (weight_type "random")
            self.backbone = MobileNetV2(output_stride, BatchNorm2d, mc_dropout=args.use_mc_dropout,
                                        pretrained=getattr(args, "weight_type", None) != "random")
            low_level_inplanes = 24
        self.aspp = ASPP(backbone, output_stride, BatchNorm2d)
        self.low_level_conv = nn.Sequential(Conv2d(low_level_inplanes, 48, 1, bias=False), BatchNorm2d(48), ReLU())
        self.seg_head = SegmentHead(args)
        self.return_features = False
        self.return_attention = False
        self._last_emb_low = None

    # deeplab.py:33-41
    def turn_on_dropout(self):
        for m in self.modules():
            if isinstance(m, Dropout):
                m.train()

    def turn_off_dropout(self):
        for m in self.modules():
            if isinstance(m, Dropout):
                m.eval()

    def set_return_features(self, return_features):
        self.return_features = return_features

    def set_return_attention(self, return_attention):
        self.return_attention = return_attention

    # ---- the graph (deeplab.py:43-59) ---------------------------------------------------------------------
    def _run(self, tape, inputs, upsample=True):
        B, _, H, W = inputs.shape
        x = E.nchw_to_nhwc(inputs)
        if self._resnet:
            feats = self.backbone.run(tape, x)                  # [c2, c3, c4, c5]
            high, low = feats[3], feats[0]
        else:
            high, low = self.backbone.run(tape, x)
        tape.mark("encoder_done")             # backward: every aspp / low-level / head gradient is enqueued at this point
        a = self.aspp.run(tape, high)
        _, Hl, Wl, _ = low.t.shape
        cat_buf = torch.empty((B, Hl, Wl, 304), dtype=torch.float32, device=inputs.device)
        up = E.bilinear(tape, a, (Hl, Wl), True, 0.0, dst=cat_buf[..., 0:256])
        llc = self.low_level_conv
        low_ = llc[1].run(tape, llc[0].run(tape, low), E.ACT_RELU, dst=cat_buf[..., 256:304])
        cat = E.concat_alias(tape, cat_buf, [up, low_])
        outs = self.seg_head.run(tape, cat)
        if not upsample:
            return outs["pred"], outs["emb"]
        pred = E.bilinear(tape, outs["pred"], (H, W), True, 0.0, out_nchw=True)
        return pred, outs["emb"]

    def forward_lowres(self, inputs):
        """Inference forward that stops in front of deeplab.py:55-56: -> (classifier logits [B,H/4,W/4,C] channels-last,
        (H, W)).  `acquisition.score_topk_lowres` interpolates them on the fly, so an acquisition
        round never writes the full-resolution logits.  Same module state semantics as forward() under no_grad."""
        if not inputs.is_cuda:
            raise RuntimeError("pixelpick_amd.DeepLab runs on the GPU only (no CPU fallback)")
        pred_v, _ = self._run(E.Tape(enabled=False), inputs.to(torch.float32), upsample=False)
        return pred_v.t, tuple(inputs.shape[2:])

    def forward(self, inputs):
        if not inputs.is_cuda:
            raise RuntimeError("pixelpick_amd.DeepLab runs on the GPU only (no CPU fallback); move the model and "
                               "inputs to cuda")
        inputs = inputs.to(torch.float32)
        params = [p for p in self.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            pred = _NetFunction.apply(inputs, self, *params)
            emb_low = self._last_emb_low
        else:
            pred_v, emb_v = self._run(E.Tape(enabled=False), inputs)
            pred, emb_low = pred_v.t, emb_v.t
        return _LazyOutputs(pred, emb_low, tuple(inputs.shape[2:]))
