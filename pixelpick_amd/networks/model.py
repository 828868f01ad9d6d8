"""FPNSeg (dilated ResNet encoder + FPN decoder) — mirror of networks/model.py:6-14 on the HIP engine.
`forward(x[B,3,H,W]) -> {"emb": [B,128,H,W], "pred": [B,C,H,W]}`; `.encoder` / `.decoder` feed the optimiser groups
of utils/utils.py:117-123."""
import torch
import torch.nn as nn

from .. import engine as E
from .decoders import FPNDecoder
from .encoder import Encoder


class _FpnOutputs(dict):
    def __init__(self, pred, emb_nhwc):
        super().__init__(pred=pred)
        self._emb_nhwc = emb_nhwc

    def __getitem__(self, k):
        if k == "emb" and not dict.__contains__(self, "emb"):
            dict.__setitem__(self, "emb", E.nhwc_to_nchw(E.Tape(False), E.Var(self._emb_nhwc)).t)
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "emb" or dict.__contains__(self, k)

    def keys(self):
        return ["emb", "pred"]


class _FpnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, model, *params):
        tape = E.Tape(enabled=True)
        pred, emb = model._run(tape, inputs)
        ctx.tape, ctx.pred_var, ctx.params = tape, pred, params
        model._last_emb = emb.t
        return pred.t

    @staticmethod
    def backward(ctx, dpred):
        tape = ctx.tape
        tape.backward(ctx.pred_var, dpred.contiguous())
        grads = tuple(tape.param_grads.get(id(p)) if p.requires_grad else None for p in ctx.params)
        ctx.tape = None
        return (None, None) + grads


class FPNSeg(nn.Module):
    # trainer.FlatTrainer: _run(..., upsample=False) stops in front of the last x2 interpolation (FPNDecoder.run(lowres=True)); the
    # logits it returns are at 1/2 resolution and interpolate with align_corners False (F.interpolate(scale_factor=2), decoders.py:101)
    LOWRES_LOGITS = True
    LOWRES_ALIGN_CORNERS = False
    LOWRES_SCALE_FACTOR = 2.0

    def __init__(self, args, load_pretrained=True):
        super().__init__()
        self.encoder = Encoder(args, load_pretrained)
        self.decoder = FPNDecoder(args)
        self._last_emb = None

    def turn_on_dropout(self):      # the FPN model has no nn.Dropout; kept for QuerySelector (query.py:152)
        pass

    def turn_off_dropout(self):
        pass

    def _run(self, tape, inputs, upsample=True):
        x = E.nchw_to_nhwc(inputs)
        feats = self.encoder.run(tape, x)
        tape.mark("encoder_done")             # backward: every decoder gradient is enqueued at this point
        outs = self.decoder.run(tape, feats, lowres=not upsample)
        if not upsample:
            return outs["pred"], None         # [B, H/2, W/2, C] channels-last
        return E.nhwc_to_nchw(tape, outs["pred"]), outs["emb"]

    def forward_lowres(self, inputs):
        """Inference forward that stops in front of the decoder's last x2 interpolation: the four branches end in the same
        F.interpolate(scale_factor=2) (decoders.py:101), the branch sum (:75) and the 1x1 classifier (:77) are linear and the
        interpolation weights sum to one, so classifier(sum_i up2(q_i)) == up2(classifier(sum_i q_i)) in exact arithmetic
        (FPNDecoder.run(lowres=True), the order the train step has used since round 4).  -> (classifier logits
        [B,H/2,W/2,C] channels-last, (H, W) = the size the reference's "pred" has).  `acquisition.score_topk_lowres(...,
        align_corners=False)` interpolates them on the fly, so an acquisition round never writes the four 128-channel
        full-resolution branch maps, their sums ("emb": 1.07 GB per 1024x2048 image) or the full-size logits."""
        if not inputs.is_cuda:
            raise RuntimeError("pixelpick_amd.FPNSeg runs on the GPU only (no CPU fallback)")
        pred_v, _ = self._run(E.Tape(enabled=False), inputs.to(torch.float32), upsample=False)
        low = pred_v.t
        return low, (2 * low.shape[1], 2 * low.shape[2])

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("pixelpick_amd.FPNSeg runs on the GPU only (no CPU fallback)")
        x = x.to(torch.float32)
        params = list(self.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            pred = _FpnFunction.apply(x, self, *params)
            emb = self._last_emb
        else:
            p, e = self._run(E.Tape(enabled=False), x)
            pred, emb = p.t, e.t
        return _FpnOutputs(pred, emb)
