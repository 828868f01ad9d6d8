"""Parameter-holding building blocks with the reference's state_dict surface.

The modules mirror `torch.nn.Conv2d / BatchNorm2d / ReLU / ReLU6 / Dropout` as used by the reference
(networks/*.py) in NAME and state_dict FORMAT (so released checkpoints, README.md:113-118, load), but
they keep convolution weights in the kernel layout (HWIO / [3,3,C]) and execute through the HIP engine
(`run(tape, x, ...)`) instead of ATen.
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from .. import engine as E


def _bump_nbt(d):
    d["_nbt_pending"] += 1


def _kaiming_normal_oihw(cout, cin_per_group, kh, kw, nonlinearity_gain=math.sqrt(2.0)):
    """torch.nn.init.kaiming_normal_ (fan_in) on an OIHW tensor (mobilenet_v2.py:152, aspp.py:25, decoders.py:128)."""
    fan_in = cin_per_group * kh * kw
    std = nonlinearity_gain / math.sqrt(fan_in)
    return torch.randn(cout, cin_per_group, kh, kw) * std


class Conv2d(nn.Module):
    """nn.Conv2d(groups=1 or groups=C depthwise 3x3).  `weight` is stored HWIO ([kh,kw,Cin,Cout]) or
    [3,3,C] for depthwise; state_dict()/load_state_dict() speak the reference's OIHW."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        assert groups in (1, in_channels), "only dense and depthwise convolutions occur in the reference"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation, self.groups = kernel_size, stride, padding, dilation, groups
        self.depthwise = groups != 1
        k = kernel_size
        if self.depthwise:
            assert k == 3 and in_channels == out_channels
            w = _kaiming_normal_oihw(out_channels, 1, k, k)[:, 0].permute(1, 2, 0).contiguous()        # [3,3,C]
        else:
            w = _kaiming_normal_oihw(out_channels, in_channels, k, k).permute(2, 3, 1, 0).contiguous()   # HWIO
        self.weight = nn.Parameter(w)
        if bias:
            bound = 1.0 / math.sqrt(in_channels // groups * k * k)      # nn.Conv2d default bias init
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    # ---- reference (OIHW) <-> kernel layout ------------------------------------------------------
    def weight_oihw(self) -> torch.Tensor:
        w = self.weight.detach()
        if self.depthwise:
            return w.permute(2, 0, 1).unsqueeze(1).contiguous()
        return w.permute(3, 2, 0, 1).contiguous()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.weight_oihw()
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + "weight"
        if key in state_dict:
            w = state_dict[key]
            if w.dim() == 4 and not self.depthwise and tuple(w.shape) == (self.out_channels, self.in_channels, self.kernel_size, self.kernel_size):
                w = w.permute(2, 3, 1, 0).contiguous()
            elif w.dim() == 4 and self.depthwise and tuple(w.shape) == (self.out_channels, 1, 3, 3):
                w = w[:, 0].permute(1, 2, 0).contiguous()
            elif tuple(w.shape) != tuple(self.weight.shape):
                error_msgs.append(f"size mismatch for {key}: got {tuple(w.shape)}")
                w = None
            if w is not None:
                with torch.no_grad():
                    self.weight.copy_(w)
        elif strict:
            missing_keys.append(key)
        bkey = prefix + "bias"
        if self.bias is not None:
            if bkey in state_dict:
                with torch.no_grad():
                    self.bias.copy_(state_dict[bkey])
            elif strict:
                missing_keys.append(bkey)
        elif bkey in state_dict and strict:
            unexpected_keys.append(bkey)

    def accepts_lazy_input(self, in_shape, extra_pad: int = 0) -> bool:
        """Can this convolution apply its producer's skipped BatchNorm on load (engine._BN_ON_LOAD)?  in_shape = (B, H, W, C)."""
        if self.depthwise:
            return True
        B, H, W, C = in_shape
        k = self.kernel_size
        return E.conv_accepts_lazy_input(B, H, W, C, self.out_channels, k, k, self.stride, self.padding + extra_pad, self.dilation)

    def run(self, tape, x, dst=None, extra_pad: int = 0, bwd_group=None):
        """extra_pad: zero padding applied to x in front of this convolution (F.pad(x, extra_pad) then conv), folded
        into the kernel's own bounds handling instead of materialising the padded tensor.
        bwd_group: (engine.ConvBwdGroup, index) - several convolutions of one input share one backward-data launch."""
        if self.depthwise:
            return E.dwconv3x3(tape, x, self.weight, self.stride, self.padding + extra_pad, self.dilation)
        return E.conv2d(tape, x, self.weight, self.bias, self.stride, self.padding + extra_pad, self.dilation, dst=dst, bwd_group=bwd_group)

    def forward(self, x):
        raise RuntimeError("pixelpick_amd layers execute through run(tape, x); call the network's forward()")

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, groups={self.groups}, bias={self.bias is not None}")


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1, affine, track_running_stats) parameters and buffers."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._nbt_pending = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._nbt_pending:
            self.num_batches_tracked += self._nbt_pending
            self._nbt_pending = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def run(self, tape, x, act=E.ACT_NONE, residual=None, dst=None, dropout=None, lazy_ok=False, single_consumer=False, consumers=0):
        """dropout: the nn.Dropout module that follows BN -> activation in the reference's Sequential (applied here so
        that it can ride in the BatchNorm kernel when both are in training mode).
        lazy_ok: the caller guarantees that the only consumer takes a skipped BatchNorm apply (engine.batch_norm_act).
        single_consumer: the caller guarantees that exactly one op reads the output (the next convolution of a Sequential): that
        convolution's backward-data launch may then run this BatchNorm's backward too (engine._conv2d_bwd).
        consumers: the caller guarantees that exactly this many ops read the output and that the FIRST of them is a dense convolution
        (mobilenet_v2.py:63-66: the next block's expand convolution, then its residual add): that convolution's backward is the last
        to run and takes this BatchNorm's backward along.  0: unknown."""
        training = self.training
        if training:
            B, H, W, _ = E.shape_of(x)          # (does not launch a deferred depthwise convolution)
            if B * H * W <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.t.shape)}")
            _lib.plan_note_host(_bump_nbt, self.__dict__)   # folded into the num_batches_tracked buffer when it is read (plain
                                                       # dict write: nn.Module.__setattr__ costs ~2 us x 60 BN layers per step)
        drop_p = dropout.p if (dropout is not None and dropout.training and dropout.p > 0.0) else 0.0
        return E.batch_norm_act(tape, x, self.weight, self.bias, self.running_mean, self.running_var, training, act,
                                residual, self.eps, self.momentum, dst=dst, dropout_p=drop_p, lazy_ok=lazy_ok,
                                single_consumer=single_consumer, consumers=consumers)

    def forward(self, x):
        raise RuntimeError("pixelpick_amd layers execute through run(tape, x)")

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


class _Placeholder(nn.Module):
    """Keeps the reference's nn.Sequential indices (state_dict key numbering); the activation itself is
    fused into the preceding BatchNorm2d.run()."""

    def forward(self, x):
        return x


class ReLU(_Placeholder):
    def __init__(self, inplace=False):
        super().__init__()


class ReLU6(_Placeholder):
    def __init__(self, inplace=False):
        super().__init__()


class Dropout2d(nn.Module):
    """nn.Dropout2d(p): whole channels of a sample are dropped.  NOT an instance of `Dropout`, so turn_on_dropout() /
    turn_off_dropout() (deeplab.py:33-41, `isinstance(m, torch.nn.Dropout)`) leave it alone, exactly as in the reference:
    it is active in train mode only."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def run(self, tape, x):
        return E.dropout2d(tape, x, self.p, self.training)

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"p={self.p}"


class Dropout(nn.Module):
    """nn.Dropout(p).  isinstance(m, Dropout) is what turn_on_dropout() toggles (deeplab.py:33-41)."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def run(self, tape, x):
        return E.dropout(tape, x, self.p, self.training)

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"p={self.p}"
