"""Minimal define-by-run tape over the network-layer entry points of the C ABI.

The reference relies on torch.autograd over ATen ops (model.py:113-121).  Here every forward AND
backward arithmetic step is a hand-written HIP kernel behind include/pixelpick_hip.h; this module only
records which kernel produced which tensor so that the reverse sweep can call the matching backward
kernels, and accumulates gradients of multiply-consumed tensors with pp_add2d.  torch provides device
memory (caching allocator) and the current stream — no torch arithmetic is used on activations.

Activations are NHWC `torch.Tensor`s [B,H,W,C]; channel slices of a wider buffer are allowed
(pixel stride `ld` = stride(2)).  Weights are HWIO (dense) / [3,3,C] (depthwise).
"""
import contextlib
import math
import ctypes
import os
import weakref
from typing import List, Optional, Sequence

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


class Var:
    """A tensor on the tape.  `grad` is filled during Tape.backward().

    In inference (tape disabled) a convolution's output may be DEFERRED: `_pending` holds the launch arguments and the
    tensor does not exist yet.  If the next consumer is an eval-mode BatchNorm, conv + BN (+ residual) + activation go
    out as ONE launch (pp_conv2d_fwd_bn_act / pp_dwconv3x3_fwd_bn_act); any other consumer reads `.t`, which launches
    the plain convolution first."""
    __slots__ = ("_t", "grad", "needs_grad", "_pending", "_lazy", "_bn_bwd_ctx", "_closed", "_grad_dst")

    def __init__(self, t: Optional[torch.Tensor], needs_grad: bool = True):
        self._t = t
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self._pending = None
        # training BatchNorm whose apply pass was skipped (_BN_ON_LOAD): (raw [B,H,W,C], scale [C], shift [C], act) - the value
        # is act(raw * scale + shift); consumers that take the pair apply it where they load, `.t` materialises it for the rest
        self._lazy = None
        # output of a training BatchNorm (+ residual, activation) whose consumers the caller named: (input Var, gamma, beta, mean,
        # invstd, act, residual Var or None, consumers) - the dense convolution that reads it FIRST (so whose backward runs last) may
        # run this BatchNorm's backward inside its own backward-data launch.  _closed: that has happened - a gradient arriving
        # afterwards means the consumer count was wrong, and raises
        self._bn_bwd_ctx = None
        self._closed = False
        # where this Var's gradient should be WRITTEN by the (single) op that produces it: a channel slice of a wider buffer (the
        # ASPP branches' output gradients side by side, ConvBwdGroup).  None: the producer allocates.
        self._grad_dst = None

    @property
    def t(self) -> torch.Tensor:
        if self._pending is not None:
            _launch_deferred(self, None)
        if self._t is None and self._lazy is not None:
            self._t = _materialise(self._lazy)
        return self._t

    @t.setter
    def t(self, value):
        self._t = value


def shape_of(v: "Var"):
    """(B, H, W, C) of a Var WITHOUT launching a deferred convolution or materialising a skipped BatchNorm apply."""
    if v._pending is None:
        if v._t is None and v._lazy is not None:
            return tuple(v._lazy[0].shape)
        return tuple(v._t.shape)
    kind, x, w, _, stride, pad, dil = v._pending
    B, H, W, C = shape_of(x)
    kh, kw = (w.shape[0], w.shape[1]) if kind == "conv" else (3, 3)
    return (B, out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil), w.shape[3] if kind == "conv" else C)


# PIXELPICK_FUSE_EVAL=0 keeps the three-launch inference form (conv, bn_eval_affine, scale_shift_act) for A/B.
_FUSE_EVAL = os.environ.get("PIXELPICK_FUSE_EVAL", "1") != "0"
# PIXELPICK_FUSE_DW_BN=1: compute a depthwise convolution inside the single-launch training BatchNorm behind it
# (pp_dwconv3x3_bn_train_fwd_fused; bit-identical).  Off by default: measured 7.16-7.18 vs 7.05 ms/step - the BatchNorm grid
# (384 blocks) is too small for the nine-tap gather, the 17 saved launches do not pay for it.
_FUSE_DW_BN = os.environ.get("PIXELPICK_FUSE_DW_BN", "0")          # "0" never, "1" always, "auto" where the rows stay in registers
_FUSE_DW_BN = {"0": False, "1": True}.get(_FUSE_DW_BN, _FUSE_DW_BN)
_ROWS_CACHED = {}


def _dw_bn_fusable(M: int, C: int) -> bool:
    if _FUSE_DW_BN != "auto":
        return bool(_FUSE_DW_BN)
    key = (M, C)
    r = _ROWS_CACHED.get(key)
    if r is None:
        r = _ROWS_CACHED[key] = bool(_lib.lib().pp_bn_fused_rows_cached(M, C)) and _BN_FUSED
    return r


def _dw_out_rows(x: "Var", stride, pad, dil):
    B, H, W, C = shape_of(x)
    return B * out_size(H, 3, stride, pad, dil) * out_size(W, 3, stride, pad, dil), C
# PIXELPICK_CONV_BN_STATS=1: a training BatchNorm behind a dense convolution takes its statistics from partial sums the
# convolution's epilogue (or its split-K reduce) wrote, and only applies - no statistics pass, no exchange between blocks, no
# co-residency requirement.  OFF by default: measured 6.96-6.98 vs 6.86-6.87 ms/step (profiles/r02_train_ablation.txt) - every
# block re-reducing the 64..128 partial rows of its channel strip costs what the exchange of the single-launch kernel costs
# (7-10 us per layer on the 1/16-resolution maps either way), and the extra allocations add 0.35 ms of host time per step.
_CONV_BN_STATS = os.environ.get("PIXELPICK_CONV_BN_STATS", "0") == "1"


# PIXELPICK_BN_ON_LOAD=1: a training BatchNorm whose producer is a convolution and whose only consumer is a depthwise convolution
# or an eligible pointwise convolution (the two inner BatchNorms of every InvertedResidual, mobilenet_v2.py:42-56, and the stem's:
# 33 of the 60) is SPLIT over its neighbours - statistics from the producer's epilogue, one small finalize launch, scale / shift /
# activation applied where the consumer loads its input (SURVEY.md 7 hard part (b)); the normalised tensor is never written.  Only
# the callers that know the consumer ask for it (`lazy_ok=`, networks/mobilenet_v2.py); tests/layerwise.py switches it off while
# it overwrites activations.  OFF by default: built, parity-tested (tests/test_bn_on_load_gpu.py) and measured SLOWER -
# 6.62-6.68 vs 6.46-6.51 ms/step, no map-size threshold wins (profiles/r03_bn_on_load.txt): the finalize launch costs what the
# single-launch BatchNorm costs on the small maps (a kernel boundary is the price of both), and the one-item-per-thread depthwise
# kernel that keeps column sums is slower than the four-outputs-per-thread one it replaces.
_BN_ON_LOAD = os.environ.get("PIXELPICK_BN_ON_LOAD", "0") == "1"
_BN_ON_LOAD_MIN_BYTES = int(os.environ.get("PIXELPICK_BN_ON_LOAD_MIN_BYTES", "0"))
_DW_WGRAD_AFFINE = os.environ.get("PIXELPICK_DW_WGRAD_AFFINE", "0") == "1"


def _materialise(lazy) -> torch.Tensor:
    raw, scale, shift, act = lazy
    B, H, W, C, ldx = _geom(raw)
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=raw.device)
    rc = _lib.lib().pp_scale_shift_act(raw.data_ptr(), ldx, B * H * W, C, scale.data_ptr(), shift.data_ptr(), None, 0, act, y.data_ptr(), C,
                                       _stream())
    _lib.check(rc, "pp_scale_shift_act")
    return y


_ACCEPTS_AFFINE = {}
_MEMO_EPOCH = [0]


def _memo_epoch():
    """The memo tables below hold answers that depend on the library's plan for a shape; a pp_debug_set_* knob may have
    changed it since (_lib.knob_epoch): start over."""
    e = _lib.knob_epoch[0]
    if e != _MEMO_EPOCH[0]:
        _MEMO_EPOCH[0] = e
        _ACCEPTS_AFFINE.clear()
        _WS_BYTES.clear()
        _CONV_WS_BYTES.clear()


def conv_accepts_lazy_input(B, H, W, Cin, Cout, kh, kw, stride, pad, dil) -> bool:
    """True when pp_conv2d_fwd_affine_in has a kernel for this dense convolution (memoised)."""
    key = (B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
    _memo_epoch()
    r = _ACCEPTS_AFFINE.get(key)
    if r is None:
        r = _ACCEPTS_AFFINE[key] = bool(_lib.lib().pp_conv2d_fwd_accepts_affine_in(*key))
    return r


def _launch_dw_fused(v: "Var", want_stats: bool):
    """v is a DEFERRED depthwise convolution: launch it through pp_dwconv3x3_fwd_fused - its input's skipped BatchNorm applied on
    load, its output's column statistics written for the BatchNorm behind it.  -> (stats, rows) | None."""
    _, x, w, _, stride, pad, dil = v._pending
    v._pending = None
    aff = x._lazy if (x._t is None and x._lazy is not None) else None
    xin = aff[0] if aff is not None else x.t
    B, H, W, C, ldx = _geom(xin)
    Ho, Wo = out_size(H, 3, stride, pad, dil), out_size(W, 3, stride, pad, dil)
    dev = xin.device
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
    stats, rows = None, 0
    if want_stats:
        rows = _wsbytes("pp_dwconv3x3_fwd_stats_rows", B, H, W, C, stride, pad, dil)
        stats = torch.empty((rows, 2, C), dtype=torch.float32, device=dev)
    rc = _lib.lib().pp_dwconv3x3_fwd_fused(xin.data_ptr(), ldx, B, H, W, C, w.data_ptr(), stride, pad, dil,
                                           aff[1].data_ptr() if aff is not None else None, aff[2].data_ptr() if aff is not None else None,
                                           aff[3] if aff is not None else 0, y.data_ptr(), C,
                                           stats.data_ptr() if stats is not None else None, stats.numel() if stats is not None else 0,
                                           _stream())
    _lib.check(rc, "pp_dwconv3x3_fwd_fused")
    v._t = y
    return (stats, rows) if want_stats else None


def _launch_deferred(v: "Var", bn):
    """Launch the convolution held in v._pending; bn = None (plain) or (gamma, beta, mean, var, eps, act, residual, dst)."""
    kind, x, w, bias, stride, pad, dil = v._pending
    if bn is None and x._t is None and x._lazy is not None:
        # the input is a BatchNorm whose apply pass was skipped: consumers that take (raw, scale, shift, act) apply it on load
        if kind == "dw":
            _launch_dw_fused(v, False)
            return
        raw, scale, shift, act = x._lazy
        B, H, W, Cin, ldx = _geom(raw)
        kh, kw, _, Cout = w.shape
        if conv_accepts_lazy_input(B, H, W, Cin, Cout, kh, kw, stride, pad, dil):
            v._pending = None
            y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=raw.device)
            ws, wsn = _conv_ws(False, raw.device, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
            rc = _lib.lib().pp_conv2d_fwd_affine_in(raw.data_ptr(), ldx, B, H, W, Cin, scale.data_ptr(), shift.data_ptr(), act, w.data_ptr(),
                                                    bias.data_ptr() if bias is not None else None, kh, kw, stride, pad, dil, y.data_ptr(),
                                                    Cout, Cout, ws, wsn, _stream())
            _lib.check(rc, "pp_conv2d_fwd_affine_in")
            v._t = y
            return
    v._pending = None
    L = _lib.lib()
    B, H, W, Cin, ldx = _geom(x.t)
    dev = x.t.device
    g = be = rm = rv = rptr = None
    eps, act, ldr, dst = 0.0, 0, 0, None
    if bn is not None:
        gamma, beta, mean, var, eps, act, residual, dst = bn
        g, be, rm, rv = gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr()
        if residual is not None:
            _, _, _, _, ldr = _geom(residual.t)
            rptr = residual.t.data_ptr()
    if kind == "conv":
        kh, kw, _, Cout = w.shape
        Ho, Wo = out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil)
        y = dst if dst is not None else torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
        _, _, _, _, ldy = _geom(y)
        ws, wsn = _conv_ws(False, dev, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
        rc = L.pp_conv2d_fwd_bn_act(x.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    kh, kw, stride, pad, dil, g, be, rm, rv, eps, rptr, ldr, act, y.data_ptr(), ldy, Cout,
                                    ws, wsn, _stream())
        _lib.check(rc, "pp_conv2d_fwd_bn_act")
    else:
        Ho, Wo = out_size(H, 3, stride, pad, dil), out_size(W, 3, stride, pad, dil)
        y = dst if dst is not None else torch.empty((B, Ho, Wo, Cin), dtype=torch.float32, device=dev)
        _, _, _, _, ldy = _geom(y)
        rc = L.pp_dwconv3x3_fwd_bn_act(x.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), stride, pad, dil, g, be, rm, rv, eps,
                                       rptr, ldr, act, y.data_ptr(), ldy, _stream())
        _lib.check(rc, "pp_dwconv3x3_fwd_bn_act")
    v._t = y


_BIG_WGRAD_MAIN = os.environ.get("PIXELPICK_BIG_WGRAD_MAIN", "1") != "0"
_BIG_WGRAD_FLOP = float(os.environ.get("PIXELPICK_BIG_WGRAD_FLOP", "8e9"))      # (the bf16x3 weight-gradient threshold of the library)
_side_streams = {}
N_SIDE_STREAMS = int(os.environ.get("PIXELPICK_SIDE_STREAMS", "1"))
SIDE_PRIORITY = int(os.environ.get("PIXELPICK_SIDE_PRIORITY", "0"))      # HIP stream priority of the weight-gradient stream(s)


def _side_stream(device, i: int = 0) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), i)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=SIDE_PRIORITY)
        _side_streams[key] = st
    return st


def _MARK():        # sentinel node kind: Tape.mark(name)
    pass


class Tape:
    # Weight-gradient kernels have no consumer until the optimiser step, and most layers of this network are too
    # small to fill 256 CUs on their own: in backward() they run on a second HIP stream, concurrently with the
    # data-gradient chain on the main stream (joined at the end of backward()).
    overlap_wgrad = os.environ.get("PIXELPICK_OVERLAP_WGRAD", "1") != "0"
    trace = None                       # debugging: a list collects (label, torch.cuda.Event) phase marks (tools/phase_times.py)

    def __init__(self, enabled: bool = True):
        refresh_stream()
        self.enabled = enabled
        self.nodes: List = []          # (backward_fn, ctx_tuple, output Var)
        self.param_grads = {}          # id(param tensor) -> grad tensor
        self.param_grad_dst = None     # optional callable(param) -> preallocated grad tensor to write into
        self._side = None
        self._side_rr = 0
        self._keepalive = []
        self.hooks = {}                # name -> callable run when backward() gets back to Tape.mark(name)
        self._jobs_first = 0           # deferred weight-gradient reduces: slots [_jobs_first, _jobs_next) of _JOB_POOL are pending
        self._jobs_next = 0
        self._arena_off = 0

    def mark(self, name: str):
        """Forward: remember this point.  backward() calls hooks[name] (if set) when every node recorded AFTER this point
        has been processed, i.e. when all their kernels are enqueued (e.g. "encoder_done": the decoder-side gradients
        are complete on their streams and can start their all-reduce while the encoder's backward still runs)."""
        if self.enabled:
            self.nodes.append((_MARK, name, None))

    def side_streams_in_use(self):
        return list(self._side.values()) if self._side else []

    def record(self, fn, ctx, out: Var):
        if self.enabled:
            self.nodes.append((fn, ctx, out))

    def grad_buffer_for(self, param: torch.Tensor) -> torch.Tensor:
        if self.param_grad_dst is not None:
            g = self.param_grad_dst(param)
            if g is not None:
                return g
        return torch.empty_like(param)

    def set_param_grad(self, param: torch.Tensor, g: torch.Tensor):
        key = id(param)
        if key in self.param_grads:   # parameter used twice on the tape: accumulate
            self.flush_reduces(g.device)          # (a deferred reduce may still owe either summand)
            prev = self.param_grads[key]
            with (self.side_stream_for(prev, g) if _role[0] == 0 else _NULL_CTX):     # both summands were written on the side stream
                add2d_(prev.view(1, -1), g.view(1, -1), prev.view(1, -1))
        else:
            self.param_grads[key] = g

    def side_stream_for(self, *tensors):
        """Context manager: run the enclosed launches on the side stream, after everything queued so far on the
        main stream.  `tensors` are kept alive until the join (they were allocated for the main stream; the caching
        allocator must not recycle them while the side stream reads them).  No torch stream context is entered:
        the enclosed code takes its stream handle from _stream() and its scratch from _ws(), both switched here
        (the torch context manager + wait_stream + record_stream cost ~30 us of host time per weight gradient)."""
        if not Tape.overlap_wgrad:
            return _NULL_CTX
        dev = tensors[0].device
        i = self._side_rr
        self._side_rr = (i + 1) % N_SIDE_STREAMS
        side = _side_stream(dev, i)
        ev = _fork_event(dev)
        _lib.plan_note(ev.record, _main_stream_obj())         # (plan_note: also part of a recorded launch plan, _lib.LaunchPlan)
        _lib.plan_note(side.wait_event, ev)
        self._keepalive.extend(tensors)
        if self._side is None:
            self._side = {}
        self._side[i] = side
        _SIDE_CTX.handle = side.cuda_stream
        _SIDE_CTX.role = 1 + i
        return _SIDE_CTX

    # ---- deferred reduces (pp_*_bwd_weight_partials + pp_wgrad_reduce_batch) ------------------------------------------------
    # A weight gradient is a partial-sum kernel followed by a small reduce launch; ~60 of those reduces per step are 0.41 ms of
    # launches of a few hundred blocks on the weight-gradient queue.  With PIXELPICK_BATCH_REDUCE=1 the layers on the side
    # stream run only their partial kernel - partials in a slice of an arena that stays untouched until the flush - and ONE
    # launch per <= 64 layers reduces them: at every Tape.mark (the gradients behind a mark must be final before its hook
    # starts their all-reduce), every PIXELPICK_REDUCE_FLUSH_MB of pending partials, and at the join.  Bit-identical to the
    # per-layer reduces (tested).  OFF by default - measured (profiles/r03_side_queue.txt): the second queue loses 57 launches and
    # 0.35 ms of kernel time and the step gets SLOWER (5.77-5.78 -> 5.80-5.93 ms): the per-layer reduces ran in the shadow of
    # the main queue on partials that were still in L2, the batch reads them back cold and its last launch (110 us when
    # everything is flushed at the join) sits between the last weight gradient and the optimiser.
    def defer_slot(self, nbytes: int, device):
        """(job struct, arena pointer, arena bytes) for one deferred layer, or None: not deferrable right now."""
        if not (_BATCH_REDUCE and Tape.overlap_wgrad and N_SIDE_STREAMS == 1) or nbytes > _ARENA_BYTES:
            return None
        if (self._jobs_next >= len(_JOB_POOL) or self._arena_off + nbytes > _ARENA_BYTES or self._jobs_next - self._jobs_first >= 64
                or self._arena_off >= _FLUSH_BYTES):
            self.flush_reduces(device)
            if self._jobs_next >= len(_JOB_POOL):
                return None
        return _JOB_POOL[self._jobs_next], _arena(device).data_ptr() + self._arena_off, nbytes

    def defer_commit(self, job):
        """The partial kernel of `job` is enqueued: keep the slot and the bytes its partials occupy (the workspace query is an
        upper bound over every slice count; the job says how many slices there are) if something was left to reduce."""
        if job.kind != 0:
            used = job.splits * job.cn * 4 * (1 if job.kind == 3 else job.ntaps)
            self._arena_off = (self._arena_off + used + 255) & ~255
            self._jobs_next += 1

    def flush_reduces(self, device):
        n = self._jobs_next - self._jobs_first
        if n > 0:
            args = (ctypes.addressof(_JOB_POOL[self._jobs_first]), n)
            if _role[0] != 0:              # called from inside a side-stream context (the single _SIDE_CTX does not nest)
                rc = _lib.lib().pp_wgrad_reduce_batch(*args, _stream())
            else:
                with self.side_stream_for(_arena(device)):
                    rc = _lib.lib().pp_wgrad_reduce_batch(*args, _stream())
            _lib.check(rc, "pp_wgrad_reduce_batch")
        self._jobs_first = self._jobs_next
        self._arena_off = 0

    def backward(self, out: Var, dout: torch.Tensor):
        refresh_stream()               # autograd may call this from its own thread / stream
        if Tape.trace is not None:
            Tape.trace.append(("bwd_begin", _mark()))
        out.grad = dout
        for fn, ctx, o in reversed(self.nodes):
            if fn is _MARK:
                hook = self.hooks.get(ctx)
                if hook is not None:
                    self.flush_reduces(dout.device)
                    hook(self)
                continue
            if o.grad is None:
                continue
            fn(self, o.grad, *ctx)
            o.grad = None             # free as we go
        self.nodes = []
        self.flush_reduces(dout.device)
        if Tape.trace is not None:
            Tape.trace.append(("bwd_main_end", _mark()))
        if self._side is not None:    # join: the optimiser / all-reduce must see every weight gradient
            for side in self._side.values():
                _lib.plan_note(_main_stream_obj().wait_stream, side)
            self._side = None
        if Tape.trace is not None:
            Tape.trace.append(("joined", _mark()))
        self._keepalive = []


# ------------------------------------------------------------------------------------------------- helpers
# torch.cuda.current_stream() costs ~7 us per call and a train step issues ~550 launches: the raw stream handle is
# cached (refreshed whenever a Tape is created / replayed and swapped by the side-stream context).
_stream_cache = [None]
_stream_obj_cache = [None]
_role = [0]                 # 0: launches go to the main stream, 1: to the side stream (selects the scratch buffer)
_fork_events = {}
_NULL_CTX = contextlib.nullcontext()


def refresh_stream():
    st = torch.cuda.current_stream()
    _stream_obj_cache[0] = st
    _stream_cache[0] = st.cuda_stream


def _stream():
    if _stream_cache[0] is None:
        refresh_stream()
    return _stream_cache[0]


def _main_stream_obj():
    if _stream_obj_cache[0] is None:
        refresh_stream()
    return _stream_obj_cache[0]


def _mark():
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(_main_stream_obj())
    return ev


def _fork_event(device):
    key = (device.type, device.index)
    ev = _fork_events.get(key)
    if ev is None:
        ev = torch.cuda.Event()
        _fork_events[key] = ev
    return ev


class _SideCtx:
    """Switches _stream() / _ws() to the side stream for the enclosed launches (re-entrant use is not needed)."""
    __slots__ = ("handle", "prev", "role")

    def __enter__(self):
        self.prev = _stream_cache[0]
        _stream_cache[0] = self.handle
        _role[0] = self.role

    def __exit__(self, *exc):
        _stream_cache[0] = self.prev
        _role[0] = 0
        return False


_SIDE_CTX = _SideCtx()


def _geom(t: torch.Tensor):
    """[B,H,W,C] NHWC (possibly a channel slice of a wider buffer) -> (B,H,W,C,ld)."""
    B, H, W, C = t.shape
    if t.is_contiguous():
        if t.dtype != torch.float32 or not t.is_cuda:
            raise AssertionError("NHWC float32 GPU tensor expected")
        return B, H, W, C, C
    assert t.dim() == 4 and t.dtype == torch.float32 and t.is_cuda, "NHWC float32 GPU tensor expected"
    if C > 1 and t.stride(3) != 1:
        raise ValueError("channel axis must be contiguous")
    if W > 1:
        ld = t.stride(2)
    elif H > 1:
        ld = t.stride(1)
    elif B > 1:
        ld = t.stride(0)
    else:
        ld = C
    if (W > 1 and H > 1 and t.stride(1) != W * ld) or (B > 1 and (H > 1 or W > 1) and t.stride(0) != H * W * ld):
        raise ValueError(f"tensor is not a pixel-strided NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return B, H, W, C, ld


# Kernel scratch (reduction partials, split-K partials): one grow-only buffer per (device, stream role).  Launches
# on one stream are serialised, so consecutive layers share it; a buffer that is outgrown is retired, never freed
# (the side stream may still be reading it and the allocator would hand it to the main stream).
_SCRATCH = {}
_SCRATCH_RETIRED = []
_WS_BYTES = {}


# bf16x3 operand planes shared inside a layer (pp_x3_split + the *_pre entry points): the forward's split of x is kept for the
# weight gradient, the backward splits dy once for backward-data AND the weight gradient - 2 instead of 4 activation splits per
# MFMA-bound layer (DeepLab-MNv2: 8 -> 4 launches of 17.7 us; DeepLabv3+-R50: 73 -> ~37 of 15.5 us).  Costs 1.5x the activation in
# bf16 planes kept from forward to backward.  PIXELPICK_X3_SHARE=0: every call splits for itself.
_X3_SHARE = os.environ.get("PIXELPICK_X3_SHARE", "1") != "0"


def _x3_planes(which: int, B, H, W, Cin, Cout, kh, kw, stride, pad, dil) -> int:
    return _wsbytes("pp_conv2d_x3_planes_bytes", which, B, H, W, Cin, Cout, kh, kw, stride, pad, dil) if _X3_SHARE else 0


def _x3_split(t: torch.Tensor, ld: int, rows: int, C: int) -> torch.Tensor:
    nb = _wsbytes("pp_x3_planes_bytes", rows, C)
    planes = torch.empty(nb, dtype=torch.uint8, device=t.device)
    _lib.check(_lib.lib().pp_x3_split(t.data_ptr(), ld, rows, C, planes.data_ptr(), nb, _stream()), "pp_x3_split")
    return planes


_BATCH_REDUCE = os.environ.get("PIXELPICK_BATCH_REDUCE", "0") != "0"
_ARENA_BYTES = int(os.environ.get("PIXELPICK_REDUCE_ARENA_MB", "768")) << 20
_FLUSH_BYTES = int(os.environ.get("PIXELPICK_REDUCE_FLUSH_MB", "768")) << 20      # flush once this many bytes of partials are pending
_ARENAS = {}
_JOB_POOL = (_lib.ReduceJob * 512)()      # fixed addresses: a recorded launch plan replays calls that point into it


def _arena(device) -> torch.Tensor:
    """Partial sums of the deferred weight gradients of one flush interval (side stream only; allocated once per device)."""
    key = (device.type, device.index)
    a = _ARENAS.get(key)
    if a is None:
        a = _ARENAS[key] = torch.empty(_ARENA_BYTES, dtype=torch.uint8, device=device)
    return a


def _ws(nbytes: int, device) -> torch.Tensor:
    key = (device.type, device.index, _role[0])
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _SCRATCH_RETIRED.append(buf)
        buf = torch.empty(max(int(nbytes), 2 * (buf.numel() if buf is not None else 0), 8 << 20), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    return buf


def _wsbytes(fn_name: str, *args) -> int:
    """Memoised pp_*_workspace_bytes query (a ctypes round trip per layer per step otherwise)."""
    key = (fn_name,) + args
    _memo_epoch()
    n = _WS_BYTES.get(key)
    if n is None:
        n = int(getattr(_lib.lib(), fn_name)(*args))
        _WS_BYTES[key] = n
    return n


# Single-launch BatchNorm (pp_bn_train_fwd_fused / pp_bn_bwd_fused).  PIXELPICK_BN_FUSED=0 selects the
# three-launch form (partials -> finalize -> apply) for A/B timing.
# PIXELPICK_BN_FUSED_DIST=0: three-launch form whenever torch.distributed runs with more than one rank (the fallback for a
# multi-GPU box on which the single-launch kernels' sibling waits misbehave under a concurrent RCCL kernel)
_BN_FUSED = os.environ.get("PIXELPICK_BN_FUSED", "1") != "0"
_BN_FUSED_DIST = os.environ.get("PIXELPICK_BN_FUSED_DIST", "1") != "0"


def disable_fused_bn_for_collectives():
    """Called by the trainer when it runs data-parallel and PIXELPICK_BN_FUSED_DIST=0."""
    global _BN_FUSED
    _BN_FUSED = False
_BN_FUSED_MAXM = int(os.environ.get("PIXELPICK_BN_FUSED_MAXM", str(1 << 62)))
_BN_XCHG = {}
_BN_XCHG_SYNC_INTS = 1 << 14          # 64 KiB of arrival counters (2048 strips: C <= 65536)
_BN_XCHG_PART_BYTES = 1 << 20         # partial-sum exchange area (<= 1024 blocks x 64 floats = 256 KiB needed)


class _FineGrainedUnavailable(RuntimeError):
    pass


class _Raw:
    """data_ptr()/numel() view of a raw device allocation."""
    __slots__ = ("ptr", "n")

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n

    def data_ptr(self):
        return self.ptr

    def numel(self):
        return self.n


def _bn_exchange(device):
    """(sync, part): the arrival counters and the partial-sum area of the single-launch BatchNorm, in FINE-GRAINED device
    memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), allocated once per device and never freed.

    The blocks of one launch sit on different XCDs, whose L2s are not coherent with each other for ordinary
    (coarse-grained) allocations: a partial written on one XCD can be served stale from another XCD's L2 even through
    agent-scope (sc1) loads - observed as run-to-run differences of a whole train step once the scratch became a
    persistent buffer (same addresses every launch).  Fine-grained memory is kept coherent by the hardware, so the
    exchange needs neither the L2 write-back nor the invalidate that a release/acquire pair on ordinary memory costs
    (11.9 vs 8.5 ms per step)."""
    # one area per (device, stream role): launches on one stream are ordered, launches on different streams may overlap
    # and must not share partial slots or the launch epoch
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), _role[0])
    ex = _BN_XCHG.get(key)
    if ex is None:
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        nbytes = _BN_XCHG_SYNC_INTS * 4 + _BN_XCHG_PART_BYTES
        ptr = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), nbytes, 1)          # hipDeviceMallocFinegrained
            if rc != 0 or not ptr.value:
                raise _FineGrainedUnavailable(f"hipExtMallocWithFlags(finegrained, {nbytes}) -> {rc}")
            rc = hip.hipMemset(ptr, 0, nbytes)
            if rc != 0:
                raise _lib.PixelPickHipError(f"hipMemset -> {rc}")
            torch.cuda.synchronize(device)
        ex = (_Raw(ptr.value, _BN_XCHG_SYNC_INTS), _Raw(ptr.value + _BN_XCHG_SYNC_INTS * 4, _BN_XCHG_PART_BYTES))
        _BN_XCHG[key] = ex
    return ex


def _bn_exchange_ok(device) -> bool:
    """False (once, with a warning) when fine-grained memory cannot be allocated: BatchNorm then uses the three-launch
    kernels, which need no cross-block exchange."""
    global _BN_FUSED
    try:
        _bn_exchange(device)
        return True
    except _FineGrainedUnavailable as e:
        import warnings
        warnings.warn(f"single-launch BatchNorm disabled: {e}")
        _BN_FUSED = False
        return False


def _acc(v: Var, g: torch.Tensor):
    """Accumulate gradient g into v (pp_add2d when v already holds one)."""
    if not v.needs_grad:
        return
    if v._closed:
        raise RuntimeError("a gradient arrived at a BatchNorm output whose backward already ran inside its consumer's backward-data "
                           "launch: the `consumers` hint given to batch_norm_act was too small")
    if v.grad is None:
        v.grad = g
        return
    B, H, W, C, lda = _geom(v.grad)
    _, _, _, _, ldb = _geom(g)
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
    rc = _lib.lib().pp_add2d(v.grad.data_ptr(), lda, g.data_ptr(), ldb, out.data_ptr(), C, B * H * W, C, _stream())
    _lib.check(rc, "pp_add2d")
    out._pp_owned = True
    v.grad = out


def add2d_(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor):
    n = a.numel()
    rc = _lib.lib().pp_add2d(a.data_ptr(), n, b.data_ptr(), n, out.data_ptr(), n, 1, n, _stream())
    _lib.check(rc, "pp_add2d")


def out_size(n: int, k: int, stride: int, pad: int, dil: int) -> int:
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ------------------------------------------------------------------------------------------------- layout
def nchw_to_nhwc(x: torch.Tensor) -> Var:
    """Network input [B,C,H,W] -> NHWC Var (no gradient: the image is a leaf, model.py:105)."""
    assert x.dim() == 4 and x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
    rc = _lib.lib().pp_nchw_to_nhwc(x.data_ptr(), B, C, H * W, y.data_ptr(), C, _stream())
    _lib.check(rc, "pp_nchw_to_nhwc")
    return Var(y, needs_grad=False)


def nhwc_to_nchw(tape: Tape, x: Var) -> Var:
    """NHWC Var -> [B,C,H,W] tensor (the FPN model's outputs); backward converts the NCHW gradient back."""
    B, H, W, C, ldx = _geom(x.t)
    y = torch.empty((B, C, H, W), dtype=torch.float32, device=x.t.device)
    rc = _lib.lib().pp_nhwc_to_nchw(x.t.data_ptr(), ldx, B, C, H * W, y.data_ptr(), _stream())
    _lib.check(rc, "pp_nhwc_to_nchw")
    out = Var(y)
    tape.record(_to_nchw_bwd, (x,), out)
    return out


def _to_nchw_bwd(tape: Tape, dy, x: Var):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    sparse = getattr(dy, "_pp_sparse_rows", False)
    dy = dy.contiguous()
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    rc = _lib.lib().pp_nchw_to_nhwc(dy.data_ptr(), B, C, H * W, dx.data_ptr(), C, _stream())
    _lib.check(rc, "pp_nchw_to_nhwc")
    if sparse and _SPARSE_ROWS and x.grad is None:
        # the gradient of a sparsely labelled loss (cross_entropy_nchw marks it): non-zero in 20 rows per image.  The flags ride on the
        # tensor; the classifier's backward-data and weight gradient visit those rows only
        flags = torch.empty(B * H * W, dtype=torch.uint8, device=dy.device)
        _lib.check(_lib.lib().pp_row_flags(dx.data_ptr(), C, B * H * W, C, flags.data_ptr(), _stream()), "pp_row_flags")
        dx._pp_rowflags = flags
    _acc(x, dx)


# ------------------------------------------------------------------------------------------------- elementwise add
def add(tape: Tape, a: Var, b: Var) -> Var:
    """a + b (FPN top-down pathway decoders.py:82, `emb = p2+p3+p4+p5` decoders.py:75)."""
    B, H, W, C, lda = _geom(a.t)
    _, _, _, _, ldb = _geom(b.t)
    assert a.t.shape == b.t.shape
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=a.t.device)
    rc = _lib.lib().pp_add2d(a.t.data_ptr(), lda, b.t.data_ptr(), ldb, y.data_ptr(), C, B * H * W, C, _stream())
    _lib.check(rc, "pp_add2d")
    out = Var(y)
    tape.record(_add_bwd, (a, b), out)
    return out


def _add_bwd(tape: Tape, dy, a: Var, b: Var):
    if getattr(dy, "_pp_owned", False):
        dy._pp_owned = False             # shared by both inputs from here on: nobody may add into it in place
    _acc(a, dy)
    _acc(b, dy)


# ------------------------------------------------------------------------------------------------- group norm / max pool
def group_norm_relu(tape: Tape, x: Var, gamma, beta, groups: int, relu: bool = True, eps: float = 1e-5) -> Var:
    """nn.GroupNorm(groups, C) [+ nn.ReLU] (decoders.py:92-94)."""
    L = _lib.lib()
    B, H, W, C, ldx = _geom(x.t)
    dev = x.t.device
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    mean = torch.empty(B * groups, dtype=torch.float32, device=dev)
    rstd = torch.empty(B * groups, dtype=torch.float32, device=dev)
    ws = _ws(_wsbytes("pp_groupnorm_workspace_bytes", B, H * W, C), dev)
    rc = L.pp_groupnorm_relu_fwd(x.t.data_ptr(), ldx, B, H * W, C, groups, gamma.data_ptr(), beta.data_ptr(), eps, int(relu),
                                 y.data_ptr(), C, mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "pp_groupnorm_relu_fwd")
    out = Var(y)
    tape.record(_gn_bwd, (x, gamma, beta, groups, mean, rstd, relu, out), out)
    return out


def _gn_bwd(tape: Tape, dy, x: Var, gamma, beta, groups, mean, rstd, relu, out: Var):
    assert relu, "GroupNorm without ReLU does not occur in the reference"
    L = _lib.lib()
    B, H, W, C, ldx = _geom(x.t)
    _, _, _, _, lddy = _geom(dy)
    dev = dy.device
    dgamma, dbeta = tape.grad_buffer_for(gamma), tape.grad_buffer_for(beta)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    ws = _ws(_wsbytes("pp_groupnorm_workspace_bytes", B, H * W, C), dev)
    rc = L.pp_groupnorm_relu_bwd(x.t.data_ptr(), ldx, dy.data_ptr(), lddy, out.t.data_ptr(), C, B, H * W, C, groups, mean.data_ptr(),
                                 rstd.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dx.data_ptr(), C,
                                 ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "pp_groupnorm_relu_bwd")
    if gamma.requires_grad:
        tape.set_param_grad(gamma, dgamma)
    if beta.requires_grad:
        tape.set_param_grad(beta, dbeta)
    _acc(x, dx)


def max_pool2d(tape: Tape, x: Var, ksize: int = 3, stride: int = 2, pad: int = 1) -> Var:
    """nn.MaxPool2d(ksize, stride, pad) (resnet_models.py:121)."""
    B, H, W, C, ldx = _geom(x.t)
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    dev = x.t.device
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
    idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=dev)
    rc = _lib.lib().pp_maxpool2d_fwd(x.t.data_ptr(), ldx, B, H, W, C, ksize, stride, pad, y.data_ptr(), C, idx.data_ptr(), _stream())
    _lib.check(rc, "pp_maxpool2d_fwd")
    out = Var(y)
    tape.record(_maxpool_bwd, (x, idx, ksize, stride, pad), out)
    return out


def _maxpool_bwd(tape: Tape, dy, x: Var, idx, ksize, stride, pad):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    _, _, _, _, lddy = _geom(dy)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    rc = _lib.lib().pp_maxpool2d_bwd(dy.data_ptr(), lddy, idx.data_ptr(), B, H, W, C, ksize, stride, pad, dx.data_ptr(), C, _stream())
    _lib.check(rc, "pp_maxpool2d_bwd")
    _acc(x, dx)


# ------------------------------------------------------------------------------------------------- dense conv
_CONV_WS_BYTES = {}      # (bwd, shape...) -> split-K workspace bytes (0: never splits)
_CONV_WS_BUF = {}        # device -> one grow-only scratch buffer; forward / backward-data run on the main stream only


_CONV_WS_MIN = 64 << 20      # first allocation of the shared convolution scratch (tests lower it to force a re-allocation)


def _conv_ws(bwd: bool, device, *shape):
    """(pointer, bytes) of the split-K scratch for this conv shape, or (None, 0)."""
    key = (bwd,) + shape
    _memo_epoch()
    need = _CONV_WS_BYTES.get(key)
    if need is None:
        L = _lib.lib()
        need = int((L.pp_conv2d_bwd_data_workspace_bytes if bwd else L.pp_conv2d_fwd_workspace_bytes)(*shape))
        _CONV_WS_BYTES[key] = need
    if need == 0:
        return None, 0
    dk = (device.type, device.index)
    buf = _CONV_WS_BUF.get(dk)
    if buf is None or buf.numel() < need:
        if buf is not None:
            # a recorded launch plan (LaunchPlan / NativePlan) holds this block's ADDRESS: keep it alive, like _ws() does - handing
            # it back to the caching allocator would let later replays write split-K partials / bf16x3 planes into memory that
            # belongs to other tensors (an eager forward at a larger shape between two replays is enough to get here)
            _SCRATCH_RETIRED.append(buf)
        buf = torch.empty(max(need, 2 * (buf.numel() if buf is not None else 0), _CONV_WS_MIN), dtype=torch.uint8, device=device)
        _CONV_WS_BUF[dk] = buf
    return buf.data_ptr(), buf.numel()


class ConvBwdGroup:
    """Backward-data of several convolutions that read ONE input (the ASPP branches, aspp.py:49-57,64-67), issued as one launch
    (pp_conv2d_bwd_data_multi) by the last of them whose backward runs.  The branches' output gradients sit side by side in `dbuf`
    [B,H,W,n*Cout] - each branch's BatchNorm backward writes its slice there directly (Var._grad_dst) - and the input's gradient is
    one implicit GEMM over (branch, tap, channel) instead of n launches, n split-K reduces and n - 1 adds."""

    def __init__(self, x: Var, specs, ws_bytes: int):
        B, H, W, Cin = shape_of(x)
        self.x, self.specs, self.ws_bytes = x, specs, ws_bytes         # specs: [(w, k, dil)] in branch order
        self.n = len(specs)
        self.Cout = specs[0][0].shape[3]
        self.dbuf = torch.empty((B, H, W, self.n * self.Cout), dtype=torch.float32, device=x.t.device)
        self.arrived = [False] * self.n

    def slice(self, i: int) -> torch.Tensor:
        return self.dbuf[..., i * self.Cout:(i + 1) * self.Cout]

    @staticmethod
    def offered(x: Var, specs) -> int:
        """Workspace bytes of the merged launch for these branches, 0 when it is not offered (odd kernel sizes with "same"
        padding, stride 1, equal Cout, 2..4 branches, a shape the 64 x 64 LDS-DMA kernel takes)."""
        if not (2 <= len(specs) <= 4):
            return 0
        B, H, W, Cin = shape_of(x)
        Cout = specs[0][0].shape[3]
        flat = []
        for w, k, d in specs:
            if tuple(w.shape) != (k, k, Cin, Cout) or w.data_ptr() & 15:
                return 0
            flat += [k, d]
        flat += [0, 0] * (4 - len(specs))
        # the merged launch addresses every branch's weights from the lowest pointer with 31-bit element offsets: the trainer's flat
        # parameter buffer always qualifies, separately allocated tensors only when the allocator put them within 8 GiB
        ptrs = [w.data_ptr() for w, _, _ in specs]
        if max(ptrs) - min(ptrs) + max(w.numel() for w, _, _ in specs) * 4 >= (1 << 33) - (1 << 26):
            return 0
        # (memoised per shape, dropped when a planner knob changes: an eager step asks on every forward)
        return _wsbytes("pp_conv2d_bwd_data_multi_workspace_bytes", B, H, W, Cin, Cout, len(specs), *flat)

    def arrive(self, tape: "Tape", i: int, dy: torch.Tensor):
        if self.arrived[i]:
            raise RuntimeError("ConvBwdGroup: a branch's backward ran twice")
        sl = self.slice(i)
        if dy.data_ptr() != sl.data_ptr() or dy.stride() != sl.stride():
            sl.copy_(dy)                                   # the producer did not write in place (not the BatchNorm backward): one copy
        self.arrived[i] = True
        if not all(self.arrived):
            return
        x = self.x
        if not x.needs_grad:
            return
        if getattr(x, "_closed", False):
            # as _issue_dx: the input's BatchNorm backward was already folded into another consumer's backward-data launch, so
            # a gradient arriving now would be dropped silently.  (Group inputs must not carry _bn_bwd_ctx.)
            raise RuntimeError("ConvBwdGroup: the input's gradient was already consumed by a fused BatchNorm backward")
        B, H, W, Cin = shape_of(x)
        dev = self.dbuf.device
        acc_into = x.grad if (x.grad is not None and getattr(x.grad, "_pp_owned", False) and x.grad.is_contiguous()
                              and tuple(x.grad.shape) == (B, H, W, Cin)) else None
        dx = acc_into if acc_into is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
        ws = _ws(self.ws_bytes, dev)
        args = []
        for w, k, d in self.specs:
            args += [w.data_ptr(), k, d]
        args += [None, 0, 0] * (4 - self.n)
        rc = _lib.lib().pp_conv2d_bwd_data_multi(self.dbuf.data_ptr(), self.n * self.Cout, B, H, W, self.Cout, self.n, *args, dx.data_ptr(), Cin, Cin,
                                                 1 if acc_into is not None else 0, ws.data_ptr(), ws.numel(), _stream())
        _lib.check(rc, "pp_conv2d_bwd_data_multi")
        tape._keepalive.append(self.dbuf)
        if acc_into is None:
            dx._pp_owned = True
            _acc(x, dx)


def conv2d(tape: Tape, x: Var, w: torch.Tensor, bias: Optional[torch.Tensor], stride=1, pad=0, dil=1,
           dst: Optional[torch.Tensor] = None, bwd_group=None) -> Var:
    """nn.Conv2d (groups=1).  w: HWIO [kh,kw,Cin,Cout].  dst: optional NHWC view to write into.
    bwd_group: (ConvBwdGroup, branch index) - this convolution's backward-data is left to the group's merged launch."""
    B, H, W, Cin = shape_of(x)          # (no launch: x may be a deferred convolution or a skipped BatchNorm apply)
    kh, kw, wcin, Cout = w.shape
    assert wcin == Cin, f"conv2d: Cin {Cin} vs weight {tuple(w.shape)}"
    if _FUSE_EVAL and not tape.enabled and dst is None:
        out = Var(None, needs_grad=False)
        out._pending = ("conv", x, w, bias, stride, pad, dil)
        return out
    if (tape.enabled and dst is None and (_CONV_BN_STATS or _BN_ON_LOAD or
                                           (_CONV_BN_FUSE and bias is None and _conv_bn_fusable(B, H, W, Cin, Cout, kh, kw, stride, pad, dil)))):
        # training: deferred as well - a training-mode BatchNorm right behind it launches the convolution with a statistics
        # epilogue (pp_conv2d_fwd_stats) and then only applies (pp_bn_train_fwd_partials); any other consumer's `.t`
        # launches the plain convolution
        out = Var(None)
        out._pending = ("conv", x, w, bias, stride, pad, dil)
        if bwd_group is not None:
            out._grad_dst = bwd_group[0].slice(bwd_group[1])
        tape.record(_conv2d_bwd, (x, w, bias, stride, pad, dil, None, bwd_group), out)
        return out
    _, _, _, _, ldx = _geom(x.t)
    Ho, Wo = out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil)
    y = dst if dst is not None else torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=x.t.device)
    _, _, _, _, ldy = _geom(y)
    ws, wsn = _conv_ws(False, x.t.device, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
    xp = None
    if (tape.enabled and w.requires_grad and _x3_planes(0, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
            and _x3_planes(2, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)):
        xp = _x3_split(x.t, ldx, B * H * W, Cin)          # read by this forward and, in backward(), by the weight gradient
    # weight planes of the step (split at begin_step on the second queue) for every layer a bf16x3 kernel serves - with the caller's
    # activation planes (xp) or, without them, through the in-kernel split (csrc/conv_x3f.hip: no x3_split launch at all)
    if xp is not None or (tape.enabled and w.requires_grad and _x3_planes(3, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)):
        wp = _weight_planes(w, 1)
        _register_weight_planes(w, 1)
        rc = _lib.lib().pp_conv2d_fwd_pre2(x.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           kh, kw, stride, pad, dil, y.data_ptr(), ldy, Cout, ws, wsn, xp.data_ptr() if xp is not None else None, wp, _stream())
    else:
        rc = _lib.lib().pp_conv2d_fwd(x.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      kh, kw, stride, pad, dil, y.data_ptr(), ldy, Cout, ws, wsn, _stream())
    _lib.check(rc, "pp_conv2d_fwd")
    out = Var(y)
    if bwd_group is not None and tape.enabled:
        out._grad_dst = bwd_group[0].slice(bwd_group[1])
    tape.record(_conv2d_bwd, (x, w, bias, stride, pad, dil, xp, bwd_group), out)
    return out


def _conv2d_bwd(tape: Tape, dy: torch.Tensor, x: Var, w, bias, stride, pad, dil, xp=None, bwd_group=None):
    L = _lib.lib()
    lazy_in = x._lazy if (x._t is None and x._lazy is not None) else None
    B, H, W, Cin, ldx = _geom(lazy_in[0] if lazy_in is not None else x.t)
    if lazy_in is not None:
        ldx = Cin                      # the weight gradient reads the materialised (contiguous) activation
    _, Ho, Wo, Cout, lddy = _geom(dy)
    kh, kw = w.shape[0], w.shape[1]
    dev = dy.device
    dyp = None
    if (w.requires_grad and x.needs_grad and lazy_in is None and _x3_planes(1, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
            and _x3_planes(2, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)):
        dyp = _x3_split(dy, lddy, B * Ho * Wo, Cout)       # one split of dy for backward-data and the weight gradient (main stream, before the fork)
    big = _BIG_WGRAD_MAIN and 2.0 * B * Ho * Wo * Cin * Cout * kh * kw >= _BIG_WGRAD_FLOP

    def _issue_wgrad():
        dw = tape.grad_buffer_for(w)
        db = tape.grad_buffer_for(bias) if (bias is not None and bias.requires_grad) else None
        # MFMA-bound layers (>= 8 GFLOP: the SegmentHead / ResNet 3x3 convolutions): their backward-data runs conv_x3_kernel with one
        # 512-thread block and 108-144 KiB of LDS per CU, which cannot become resident beside the weight-gradient blocks (647 / 751 us
        # in the trace instead of 210 / 350 alone) - and two MFMA-bound kernels gain nothing from sharing the matrix pipes anyway:
        # the weight gradient of such a layer goes out on the MAIN stream, behind nothing it could overlap with
        with (_NULL_CTX if big else tape.side_stream_for(lazy_in[0] if lazy_in is not None else x.t, dy, dw, db)):
            if lazy_in is not None:
                # the weight gradient needs act(bn(raw)), which the forward never wrote: one elementwise launch on the
                # weight-gradient stream (idle between the encoder's small weight gradients), kept alive until the join
                tape._keepalive.append(x.t)
            nws = _wsbytes("pp_conv2d_bwd_weight_workspace_bytes", B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
            slot = tape.defer_slot(nws, dev) if (not big and db is None) else None
            flags = getattr(dy, "_pp_rowflags", None) if (_SPARSE_ROWS and _SPARSE_WGRAD) else None
            nsp = (_wsbytes("pp_conv1x1_bwd_weight_sparse_workspace_bytes", B * H * W, Cin, Cout)
                   if (flags is not None and kh == 1 and kw == 1 and stride == 1 and pad == 0 and lazy_in is None and xp is None and dyp is None
                       and flags.numel() == B * H * W) else 0)
            if nsp:
                # the classifier behind a sparsely labelled loss: only the flagged rows of dy are non-zero (80 of 32768 for DeepLab, of
                # 524288 for FPNSeg) - gather those instead of streaming x and dy whole
                ws = _ws(nsp, dev)
                rc = L.pp_conv1x1_bwd_weight_sparse(x.t.data_ptr(), ldx, B * H * W, Cin, dy.data_ptr(), lddy, Cout, flags.data_ptr(),
                                                    dw.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), ws.numel(), _stream())
                tape._keepalive.append(flags)
            elif slot is not None:
                job, wptr, _ = slot
                rc = L.pp_conv2d_bwd_weight_partials(x.t.data_ptr(), ldx, B, H, W, Cin, dy.data_ptr(), lddy, Cout, kh, kw, stride, pad, dil,
                                                     dw.data_ptr(), None, wptr, nws, ctypes.addressof(job), _stream())
                tape.defer_commit(job)
            elif xp is not None or dyp is not None:
                ws = _ws(nws, dev)
                tape._keepalive.extend(t for t in (xp, dyp) if t is not None)
                rc = L.pp_conv2d_bwd_weight_pre(x.t.data_ptr(), ldx, B, H, W, Cin, dy.data_ptr(), lddy, Cout, kh, kw, stride, pad, dil,
                                                dw.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), ws.numel(),
                                                xp.data_ptr() if xp is not None else None, dyp.data_ptr() if dyp is not None else None, _stream())
            else:
                ws = _ws(nws, dev)
                rc = L.pp_conv2d_bwd_weight(x.t.data_ptr(), ldx, B, H, W, Cin, dy.data_ptr(), lddy, Cout, kh, kw, stride, pad, dil,
                                            dw.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(), ws.numel(),
                                            _stream())
        _lib.check(rc, "pp_conv2d_bwd_weight")
        tape.set_param_grad(w, dw)
        if db is not None:
            tape.set_param_grad(bias, db)
    def _issue_dx():
        if x._closed:
            raise RuntimeError("a second convolution's backward reached a BatchNorm output whose backward already ran (wrong `consumers` hint)")
        bctx = x._bn_bwd_ctx if (x.needs_grad and _CONV_BN_FUSE_BWD and lazy_in is None) else None
        if bctx is not None:
            # every OTHER consumer's gradient must be here already (their backward ran earlier: they read x later in the forward);
            # one other consumer = the residual add of the block behind, whose gradient is a contiguous tensor of x's shape
            have = x.grad is not None
            if bctx[7] != (2 if have else 1) or (have and not (x.grad.is_contiguous() and tuple(x.grad.shape) == (B, H, W, Cin))):
                bctx = None
        if (bctx is not None and bctx[0].needs_grad and not ((dy.data_ptr() | w.data_ptr()) & 15) and lddy % 4 == 0
                and _conv_bn_bwd_fusable(B, H, W, Cin, Cout, kh, kw, stride, pad, dil) and _bn_exchange_ok(dev)):
            # x is the output of a training BatchNorm (+ residual, activation) and this convolution's backward is the last of its
            # consumers' to run: the BatchNorm's backward runs inside this backward-data launch (pp_conv2d_bwd_data_bn_bwd) - the BatchNorm
            # node then finds no gradient on its output and is skipped
            bin_, g_, b_, mean_, invstd_, act_, res_, _ = bctx
            _, _, _, _, ldb = _geom(bin_.t)
            dxbn = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
            dres = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev) if (res_ is not None and res_.needs_grad) else None
            gin = x.grad
            dg_, db_ = tape.grad_buffer_for(g_), tape.grad_buffer_for(b_)
            sync, xws = _bn_exchange(dev)
            rc = L.pp_conv2d_bwd_data_bn_bwd(dy.data_ptr(), lddy, B, Ho, Wo, Cout, w.data_ptr(), kh, kw, stride, pad, dil, H, W, Cin,
                                             bin_.t.data_ptr(), ldb, mean_.data_ptr(), invstd_.data_ptr(), g_.data_ptr(), b_.data_ptr(), act_,
                                             dg_.data_ptr(), db_.data_ptr(), dxbn.data_ptr(), Cin,
                                             gin.data_ptr() if gin is not None else None, Cin, dres.data_ptr() if dres is not None else None, Cin,
                                             xws.data_ptr(), xws.numel(), sync.data_ptr(), sync.numel(), _stream())
            _lib.check(rc, "pp_conv2d_bwd_data_bn_bwd")
            tape.set_param_grad(g_, dg_)
            tape.set_param_grad(b_, db_)
            if gin is not None:
                tape._keepalive.append(gin)
            x.grad = None                    # consumed here: the BatchNorm node is skipped
            x._closed = True
            dxbn._pp_owned = True
            _acc(bin_, dxbn)
            if dres is not None:
                dres._pp_owned = True
                _acc(res_, dres)
        elif x.needs_grad:
            # x already has a gradient from another consumer (the residual branch): add into it in the kernel's epilogue
            acc_into = x.grad if (x.grad is not None and getattr(x.grad, "_pp_owned", False) and x.grad.is_contiguous()
                                  and tuple(x.grad.shape) == (B, H, W, Cin)) else None
            flags = getattr(dy, "_pp_rowflags", None) if _SPARSE_ROWS else None
            if (flags is not None and kh == 1 and kw == 1 and stride == 1 and pad == 0 and acc_into is None and lazy_in is None and Cin % 4 == 0
                    and flags.numel() == B * H * W and dyp is None):
                # pointwise convolution behind a sparse gradient: rows stay rows - zeros for the unflagged ones, the flags travel on
                dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
                rc = L.pp_conv1x1_bwd_data_sparse(dy.data_ptr(), lddy, B * H * W, Cout, w.data_ptr(), Cin, flags.data_ptr(), dx.data_ptr(), Cin, _stream())
                _lib.check(rc, "pp_conv1x1_bwd_data_sparse")
                dx._pp_rowflags = flags              # (not _pp_owned: nobody may add into it in place, the flags would go stale)
                _acc(x, dx)
                return
            dx = acc_into if acc_into is not None else torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev)
            ws, wsn = _conv_ws(True, dev, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
            if dyp is not None or (stride == 1 and w.requires_grad and _x3_planes(4, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)):
                wp = _weight_planes(w, 0) if stride == 1 else None
                if stride == 1:
                    _register_weight_planes(w, 0)
                rc = L.pp_conv2d_bwd_data_pre2(dy.data_ptr(), lddy, B, Ho, Wo, Cout, w.data_ptr(), kh, kw, stride, pad, dil,
                                               dx.data_ptr(), Cin, H, W, Cin, 1 if acc_into is not None else 0, ws, wsn,
                                               dyp.data_ptr() if dyp is not None else None, wp, _stream())
            else:
                rc = L.pp_conv2d_bwd_data(dy.data_ptr(), lddy, B, Ho, Wo, Cout, w.data_ptr(), kh, kw, stride, pad, dil,
                                          dx.data_ptr(), Cin, H, W, Cin, 1 if acc_into is not None else 0, ws, wsn, _stream())
            _lib.check(rc, "pp_conv2d_bwd_data")
            if acc_into is None:
                dx._pp_owned = True              # fresh tensor referenced by x.grad only: later consumers may add in place
                _acc(x, dx)

    # PIXELPICK_WGRAD_LATE (default off; measured, profiles/r03_side_queue.txt): backward-data first, then the fork and the weight
    # gradient - on the second queue it would then start when this layer's backward-data has finished and overlap the memory-bound
    # BatchNorm backward behind it instead of the backward-data of its own layer.  Slower both eager and replayed (+0.05 ms).
    if bwd_group is not None:
        # the input gradient comes from the group's merged launch (issued by the last branch to arrive); the weight gradient is this layer's own
        if w.requires_grad:
            _issue_wgrad()
        bwd_group[0].arrive(tape, bwd_group[1], dy)
        return
    if w.requires_grad and (big or not _WGRAD_LATE):
        _issue_wgrad()
        _issue_dx()
    else:
        _issue_dx()
        if w.requires_grad:
            _issue_wgrad()


# ------------------------------------------------------------------------------------------------- depthwise conv
def dwconv3x3(tape: Tape, x: Var, w: torch.Tensor, stride=1, pad=0, dil=1) -> Var:
    """Depthwise 3x3 (groups=C).  w: [3,3,C]."""
    if _FUSE_EVAL and not tape.enabled:
        out = Var(None, needs_grad=False)
        out._pending = ("dw", x, w, None, stride, pad, dil)
        return out
    if tape.enabled and (_BN_ON_LOAD or (_FUSE_DW_BN and _dw_bn_fusable(*_dw_out_rows(x, stride, pad, dil)))):
        # training: defer as well - a training-mode BatchNorm right behind it computes the convolution inside its own
        # single launch (pp_dwconv3x3_bn_train_fwd_fused); any other consumer's `.t` launches the plain convolution
        out = Var(None)
        out._pending = ("dw", x, w, None, stride, pad, dil)
        tape.record(_dwconv_bwd, (x, w, stride, pad, dil), out)
        return out
    B, H, W, C, ldx = _geom(x.t)
    Ho, Wo = out_size(H, 3, stride, pad, dil), out_size(W, 3, stride, pad, dil)
    y = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=x.t.device)
    rc = _lib.lib().pp_dwconv3x3_fwd(x.t.data_ptr(), ldx, B, H, W, C, w.data_ptr(), stride, pad, dil, y.data_ptr(), C, _stream())
    _lib.check(rc, "pp_dwconv3x3_fwd")
    out = Var(y)
    tape.record(_dwconv_bwd, (x, w, stride, pad, dil), out)
    return out


def _dwconv_bwd(tape: Tape, dy, x: Var, w, stride, pad, dil):
    L = _lib.lib()
    lazy_in = x._lazy if (x._t is None and x._lazy is not None) else None
    B, H, W, C, ldx = _geom(lazy_in[0] if lazy_in is not None else x.t)
    if lazy_in is not None and not _DW_WGRAD_AFFINE:
        # (measured: applying the affine inside the nine-tap weight-gradient kernels costs 3x their time; one elementwise launch
        # on the weight-gradient stream is cheaper)
        with tape.side_stream_for(lazy_in[0], dy):
            tape._keepalive.append(x.t)
        lazy_in, ldx = None, C
    _, Ho, Wo, _, lddy = _geom(dy)
    dev = dy.device
    if w.requires_grad:
        dw = tape.grad_buffer_for(w)
        if lazy_in is not None:
            raw, scale, shift, act = lazy_in
            with tape.side_stream_for(raw, dy, dw):
                ws = _ws(_wsbytes("pp_colreduce_workspace_bytes", B * Ho * Wo, C), dev)
                rc = L.pp_dwconv3x3_bwd_weight_affine_in(raw.data_ptr(), ldx, B, H, W, C, scale.data_ptr(), shift.data_ptr(), act,
                                                         dy.data_ptr(), lddy, stride, pad, dil, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         _stream())
            _lib.check(rc, "pp_dwconv3x3_bwd_weight_affine_in")
        else:
            with tape.side_stream_for(x.t, dy, dw):
                nws = _wsbytes("pp_colreduce_workspace_bytes", B * Ho * Wo, C)
                slot = tape.defer_slot(nws, dev)
                if slot is not None:
                    job, wptr, _ = slot
                    rc = L.pp_dwconv3x3_bwd_weight_partials(x.t.data_ptr(), ldx, B, H, W, C, dy.data_ptr(), lddy, stride, pad, dil,
                                                            dw.data_ptr(), wptr, nws, ctypes.addressof(job), _stream())
                    tape.defer_commit(job)
                else:
                    ws = _ws(nws, dev)
                    rc = L.pp_dwconv3x3_bwd_weight(x.t.data_ptr(), ldx, B, H, W, C, dy.data_ptr(), lddy, stride, pad, dil, dw.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), _stream())
            _lib.check(rc, "pp_dwconv3x3_bwd_weight")
        tape.set_param_grad(w, dw)
    if x.needs_grad:
        dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        rc = L.pp_dwconv3x3_bwd_data(dy.data_ptr(), lddy, B, H, W, C, w.data_ptr(), stride, pad, dil, dx.data_ptr(), C, _stream())
        _lib.check(rc, "pp_dwconv3x3_bwd_data")
        _acc(x, dx)


def _dw_bn_train_fused(tape: Tape, x: Var, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst):
    """x is a DEFERRED depthwise convolution: run it inside the single-launch training BatchNorm.  Returns the output Var,
    or None when the shape is outside the fused kernel's range (the caller then takes the ordinary path)."""
    _, xin, w, _, stride, pad, dil = x._pending
    B, H, W, C, ld_in = _geom(xin.t)
    Ho, Wo = out_size(H, 3, stride, pad, dil), out_size(W, 3, stride, pad, dil)
    M = B * Ho * Wo
    if M > _BN_FUSED_MAXM or M >= (1 << 31) or C > 65536 or C % 4 or ld_in % 4:
        return None
    dev = xin.t.device
    xbuf = torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    y = dst if dst is not None else torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
    _, _, _, _, ldy = _geom(y)
    rptr, ldr = (None, 0)
    if residual is not None:
        _, _, _, _, ldr = _geom(residual.t)
        rptr = residual.t.data_ptr()
    sync, ws = _bn_exchange(dev)
    rc = _lib.lib().pp_dwconv3x3_bn_train_fwd_fused(
        xin.t.data_ptr(), ld_in, B, H, W, C, w.data_ptr(), stride, pad, dil, xbuf.data_ptr(), C, gamma.data_ptr(), beta.data_ptr(),
        eps, momentum, running_mean.data_ptr() if running_mean is not None else None,
        running_var.data_ptr() if running_var is not None else None, mean.data_ptr(), invstd.data_ptr(), rptr, ldr, act,
        y.data_ptr(), ldy, ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), _stream())
    _lib.check(rc, "pp_dwconv3x3_bn_train_fwd_fused")
    x._pending = None
    x._t = xbuf                                   # the convolution output exists now (BatchNorm backward reads it)
    out = Var(y)
    tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, residual, out, 1.0), out)
    return out


_CONV_BN_STATS_MAX_ROWS = int(os.environ.get("PIXELPICK_CONV_BN_STATS_MAX_ROWS", "160"))


# PIXELPICK_CONV_BN_FUSE (default on): a training BatchNorm right behind a dense convolution whose whole grid is co-resident (the
# 1/16- and 1/8-resolution pointwise layers) is finished in the convolution's own epilogue - pp_conv2d_fwd_bn_train: one launch
# instead of two, no statistics pass, no second read of the convolution's output.  The convolution is DEFERRED until its consumer is
# known; any other consumer's `.t` launches it plain.
_CONV_BN_FUSE = os.environ.get("PIXELPICK_CONV_BN_FUSE", "1") != "0"


def _conv_bn_fusable(B, H, W, Cin, Cout, kh, kw, stride, pad, dil) -> bool:
    if not (_BN_FUSED and Cout % 32 == 0):
        return False
    if not _wsbytes("pp_conv2d_fwd_bn_train_ok", B, H, W, Cin, Cout, kh, kw, stride, pad, dil):
        return False
    return _wsbytes("pp_conv2d_fwd_bn_train_xchg_bytes", B, H, W, Cin, Cout, kh, kw, stride, pad, dil) <= _BN_XCHG_PART_BYTES


_CONV_BN_FUSE_BWD = os.environ.get("PIXELPICK_CONV_BN_FUSE_BWD", "1") != "0"
# PIXELPICK_SPARSE_ROWS (default on): the loss gradient's non-zero rows are flagged (cross_entropy_lowres) and the flags follow the
# gradient through the classifier's backward-data (pp_conv1x1_bwd_data_sparse) into the BatchNorm backward (pp_bn_bwd_fused_sparse)
_SPARSE_ROWS = os.environ.get("PIXELPICK_SPARSE_ROWS", "1") != "0"
_WGRAD_LATE = os.environ.get("PIXELPICK_WGRAD_LATE", "0") != "0"
# PIXELPICK_SPARSE_WGRAD (default on): the weight gradient of a pointwise convolution behind flagged rows gathers those rows
# (pp_conv1x1_bwd_weight_sparse) instead of running the dense kernels
_SPARSE_WGRAD = os.environ.get("PIXELPICK_SPARSE_WGRAD", "1") != "0"


def _conv_bn_bwd_fusable(B, H, W, Cin, Cout, kh, kw, stride, pad, dil) -> bool:
    """A backward-data convolution of this shape can also run the BatchNorm backward of the layer in front of it."""
    if not (_BN_FUSED and Cin % 32 == 0):
        return False
    if not _wsbytes("pp_conv2d_bwd_data_bn_bwd_ok", B, H, W, Cin, Cout, kh, kw, stride, pad, dil):
        return False
    return _wsbytes("pp_conv2d_bwd_data_bn_bwd_xchg_bytes", B, H, W, Cin, Cout, kh, kw, stride, pad, dil) <= _BN_XCHG_PART_BYTES


def _conv_bn_train_fused(tape: Tape, x: Var, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst, ncons=0):
    """x is a DEFERRED dense convolution followed by this training BatchNorm: one launch (pp_conv2d_fwd_bn_train).  None: not
    applicable (x stays deferred, the caller takes the ordinary path)."""
    _, xin, w, bias, stride, pad, dil = x._pending
    if bias is not None:
        return None
    B, H, W, Cin, ldx = _geom(xin.t)
    kh, kw, _, Cout = w.shape
    if not _conv_bn_fusable(B, H, W, Cin, Cout, kh, kw, stride, pad, dil) or ldx % 4:
        return None
    if (xin.t.data_ptr() | w.data_ptr()) & 15:
        return None                                 # the one-launch kernels are the 16-byte-vector forms only: take the two-launch path
    dev = xin.t.device
    if not _bn_exchange_ok(dev):
        return None
    Ho, Wo = out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil)
    raw = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
    y = dst if dst is not None else torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
    _, _, _, _, ldy = _geom(y)
    mean = torch.empty(Cout, dtype=torch.float32, device=dev)
    invstd = torch.empty(Cout, dtype=torch.float32, device=dev)
    rptr, ldr = (None, 0)
    if residual is not None:
        _, _, _, _, ldr = _geom(residual.t)
        rptr = residual.t.data_ptr()
    sync, ws = _bn_exchange(dev)
    rc = _lib.lib().pp_conv2d_fwd_bn_train(xin.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), kh, kw, stride, pad, dil, raw.data_ptr(), Cout,
                                           gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                           running_mean.data_ptr() if running_mean is not None else None,
                                           running_var.data_ptr() if running_var is not None else None, mean.data_ptr(), invstd.data_ptr(),
                                           rptr, ldr, act, y.data_ptr(), ldy, Cout, ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(),
                                           _stream())
    _lib.check(rc, "pp_conv2d_fwd_bn_train")
    x._pending = None
    x._t = raw
    out = Var(y)
    tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, residual, out, 1.0), out)
    if ncons:
        out._bn_bwd_ctx = (x, gamma, beta, mean, invstd, act, residual, ncons)
    return out


def _launch_conv_stats(x: Var, max_rows: Optional[int] = None):
    """x is a DEFERRED dense convolution: launch it with the BatchNorm-statistics epilogue.  -> (stats [rows,2,Cout], rows), or
    None when this shape delivers no statistics (x stays deferred)."""
    _, xin, w, bias, stride, pad, dil = x._pending
    B, H, W, Cin, ldx = _geom(xin.t)
    kh, kw, _, Cout = w.shape
    rows = _wsbytes("pp_conv2d_fwd_stats_rows", B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
    if rows <= 0 or rows > (_CONV_BN_STATS_MAX_ROWS if max_rows is None else max_rows) or Cout % 4:
        return None                                 # (large maps: thousands of partial rows - the stem BatchNorm took 93 us instead of 22)
    dev = xin.t.device
    Ho, Wo = out_size(H, kh, stride, pad, dil), out_size(W, kw, stride, pad, dil)
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=dev)
    stats = torch.empty((rows, 2, Cout), dtype=torch.float32, device=dev)
    ws, wsn = _conv_ws(False, dev, B, H, W, Cin, Cout, kh, kw, stride, pad, dil)
    rc = _lib.lib().pp_conv2d_fwd_stats(xin.t.data_ptr(), ldx, B, H, W, Cin, w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        kh, kw, stride, pad, dil, y.data_ptr(), Cout, Cout, ws, wsn, stats.data_ptr(), stats.numel(),
                                        _stream())
    _lib.check(rc, "pp_conv2d_fwd_stats")
    x._pending = None
    x._t = y
    return stats, rows


def _conv_bn_train_partials(tape: Tape, x: Var, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst, dropout_p):
    """Dense convolution -> training BatchNorm (+ residual, activation, dropout): the convolution's epilogue delivers the
    statistics, the BatchNorm kernel only applies.  None: not applicable (the caller takes the ordinary path)."""
    if dropout_p > 0.0 and act == ACT_RELU6:
        return None
    got = _launch_conv_stats(x)
    if got is None:
        return None
    stats, rows = got
    B, H, W, C, ldx = _geom(x.t)
    M = B * H * W
    dev = x.t.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    y = dst if dst is not None else torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    _, _, _, _, ldy = _geom(y)
    rptr, ldr = (None, 0)
    if residual is not None:
        _, _, _, _, ldr = _geom(residual.t)
        rptr = residual.t.data_ptr()
    dseed, sd = 0, None
    if dropout_p > 0.0:                               # same seed sequence as dropout()
        _dropout_counter[0] += 1
        dseed = (_dropout_counter[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        sd = _dropout_seed_dev[0]
    rc = _lib.lib().pp_bn_train_fwd_partials(x.t.data_ptr(), ldx, M, C, stats.data_ptr(), rows, gamma.data_ptr(), beta.data_ptr(), eps,
                                             momentum, running_mean.data_ptr() if running_mean is not None else None,
                                             running_var.data_ptr() if running_var is not None else None, mean.data_ptr(),
                                             invstd.data_ptr(), rptr, ldr, act, float(dropout_p), dseed,
                                             sd.data_ptr() if sd is not None else None, y.data_ptr(), ldy, _stream())
    _lib.check(rc, "pp_bn_train_fwd_partials")
    out = Var(y)
    tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, residual, out, 1.0 / (1.0 - dropout_p)), out)
    return out


# ------------------------------------------------------------------------------------------------- batch norm (+act, +residual)
def _bn_on_load(tape: Tape, x: Var, gamma, beta, running_mean, running_var, act, eps, momentum):
    """Training BatchNorm (+ activation) behind the DEFERRED convolution x, split over its neighbours: x is launched with the
    statistics epilogue, pp_bn_finalize_partials makes mean / invstd / running statistics / scale / shift, and the result is a
    LAZY Var (raw, scale, shift, act) whose consumer applies it on load.  None: not applicable (x stays deferred)."""
    if x._pending[0] == "conv":
        got = _launch_conv_stats(x, max_rows=1 << 30)
    else:
        got = _launch_dw_fused(x, True)
    if got is None:
        return None
    stats, rows = got
    B, H, W, C, _ = _geom(x._t)
    dev = x._t.device
    mean, invstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
    rc = _lib.lib().pp_bn_finalize_partials(stats.data_ptr(), rows, B * H * W, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                            running_mean.data_ptr() if running_mean is not None else None,
                                            running_var.data_ptr() if running_var is not None else None, mean.data_ptr(),
                                            invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _stream())
    _lib.check(rc, "pp_bn_finalize_partials")
    out = Var(None)
    out._lazy = (x._t, scale, shift, act)
    tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, None, out, 1.0), out)
    return out


def batch_norm_act(tape: Tape, x: Var, gamma, beta, running_mean, running_var, training: bool, act: int = ACT_NONE,
                   residual: Optional[Var] = None, eps: float = 1e-5, momentum: float = 0.1,
                   dst: Optional[torch.Tensor] = None, dropout_p: float = 0.0, lazy_ok: bool = False, single_consumer: bool = False,
                   consumers: int = 0) -> Var:
    """nn.BatchNorm2d -> (+ residual) -> activation [-> nn.Dropout(dropout_p), already known to be active].
    Training: batch statistics + running-stat update; the dropout rides in the single-launch kernel's apply pass.
    lazy_ok: the caller knows that the ONLY consumer is a depthwise convolution or a pointwise convolution that
    conv_accepts_lazy_input() - the BatchNorm may then be split over its neighbours (_bn_on_load)."""
    if (lazy_ok and _BN_ON_LOAD and training and tape.enabled and x._pending is not None and residual is None and dropout_p == 0.0
            and dst is None and shape_of(x)[3] % 4 == 0 and 4 * math.prod(shape_of(x)) >= _BN_ON_LOAD_MIN_BYTES):
        out = _bn_on_load(tape, x, gamma, beta, running_mean, running_var, act, eps, momentum)
        if out is not None:
            return out
    if dropout_p > 0.0 and not (training and _BN_FUSED and act != ACT_RELU6 and dst is None and _bn_exchange_ok(x.t.device)):
        y = batch_norm_act(tape, x, gamma, beta, running_mean, running_var, training, act, residual, eps, momentum, dst)
        return dropout(tape, y, dropout_p, True)
    if not training and x._pending is not None and not tape.enabled:
        _launch_deferred(x, (gamma, beta, running_mean, running_var, eps, act, residual, dst))
        return Var(x._t, needs_grad=False)
    L = _lib.lib()
    if training and x._pending is not None and x._pending[0] == "conv" and tape.enabled and _CONV_BN_FUSE and dropout_p == 0.0:
        ncons = consumers if consumers > 0 else (1 if (lazy_ok or single_consumer) else 0)
        out = _conv_bn_train_fused(tape, x, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst,
                                   ncons if (dst is None and (residual is None or act == ACT_NONE)) else 0)
        if out is not None:
            return out
    if training and x._pending is not None and x._pending[0] == "conv" and tape.enabled and _CONV_BN_STATS:
        out = _conv_bn_train_partials(tape, x, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst, dropout_p)
        if out is not None:
            return out
    if (training and x._pending is not None and x._pending[0] == "dw" and tape.enabled and dropout_p == 0.0 and _BN_FUSED
            and _FUSE_DW_BN and _dw_bn_fusable(*_dw_out_rows(x._pending[1], *x._pending[4:7])) and _bn_exchange_ok(x._pending[1].t.device)):
        fused = _dw_bn_train_fused(tape, x, gamma, beta, running_mean, running_var, act, residual, eps, momentum, dst)
        if fused is not None:
            return fused
    B, H, W, C, ldx = _geom(x.t)
    M = B * H * W
    dev = x.t.device
    if training and _BN_FUSED and M <= _BN_FUSED_MAXM and C <= 65536 and _bn_exchange_ok(dev):
        # one launch: statistics + running-stat update + affine + residual + activation
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        y = dst if dst is not None else torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
        _, _, _, _, ldy = _geom(y)
        rptr, ldr = (None, 0)
        if residual is not None:
            _, _, _, _, ldr = _geom(residual.t)
            rptr = residual.t.data_ptr()
        sync, ws = _bn_exchange(dev)
        dseed, sd = 0, None
        if dropout_p > 0.0:                           # same seed sequence as dropout()
            _dropout_counter[0] += 1
            dseed = (_dropout_counter[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            sd = _dropout_seed_dev[0]
        rc = L.pp_bn_train_fwd_fused(x.t.data_ptr(), ldx, M, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                                     running_mean.data_ptr() if running_mean is not None else None,
                                     running_var.data_ptr() if running_var is not None else None,
                                     mean.data_ptr(), invstd.data_ptr(), rptr, ldr, act, float(dropout_p), dseed,
                                     sd.data_ptr() if sd is not None else None, y.data_ptr(), ldy,
                                     ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), _stream())
        _lib.check(rc, "pp_bn_train_fwd_fused")
        out = Var(y)
        tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, residual, out, 1.0 / (1.0 - dropout_p)), out)
        ncons = consumers if consumers > 0 else (1 if (lazy_ok or single_consumer) else 0)
        if ncons and dropout_p == 0.0 and dst is None and (residual is None or act == ACT_NONE):
            out._bn_bwd_ctx = (x, gamma, beta, mean, invstd, act, residual, ncons)
        return out
    scale = torch.empty(C, dtype=torch.float32, device=dev)
    shift = torch.empty(C, dtype=torch.float32, device=dev)
    mean = invstd = None
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        ws = _ws(_wsbytes("pp_colreduce_workspace_bytes", M, C), dev)
        rc = L.pp_bn_train_fwd(x.t.data_ptr(), ldx, M, C, gamma.data_ptr(), beta.data_ptr(), eps, momentum,
                               running_mean.data_ptr() if running_mean is not None else None,
                               running_var.data_ptr() if running_var is not None else None,
                               mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                               ws.data_ptr(), ws.numel(), _stream())
        _lib.check(rc, "pp_bn_train_fwd")
    else:
        rc = L.pp_bn_eval_affine(C, gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), eps,
                                 scale.data_ptr(), shift.data_ptr(), _stream())
        _lib.check(rc, "pp_bn_eval_affine")
    y = dst if dst is not None else torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    _, _, _, _, ldy = _geom(y)
    rptr, ldr = (None, 0)
    if residual is not None:
        _, _, _, _, ldr = _geom(residual.t)
        rptr = residual.t.data_ptr()
    rc = L.pp_scale_shift_act(x.t.data_ptr(), ldx, M, C, scale.data_ptr(), shift.data_ptr(), rptr, ldr, act, y.data_ptr(), ldy, _stream())
    _lib.check(rc, "pp_scale_shift_act")
    out = Var(y)
    if training:
        tape.record(_bn_bwd, (x, gamma, beta, mean, invstd, act, residual, out, 1.0), out)
    else:
        tape.record(_bn_eval_bwd, (x, scale, act, residual, out), out)
    return out


def _bn_bwd(tape: Tape, dy, x: Var, gamma, beta, mean, invstd, act, residual, out: Var, gscale: float = 1.0):
    L = _lib.lib()
    B, H, W, C, ldx = _geom(x.t)
    M = B * H * W
    _, _, _, _, lddy = _geom(dy)
    lazy_out = out._t is None and out._lazy is not None          # the activated output was never written (_bn_on_load)
    ldya = 0 if lazy_out else _geom(out.t)[4]
    dev = dy.device
    dgamma = tape.grad_buffer_for(gamma)
    dbeta = tape.grad_buffer_for(beta)
    gd = x._grad_dst if x.grad is None else None          # the consumer of this gradient wants it in a slice of its own buffer
    if gd is not None and not (tuple(gd.shape) == (B, H, W, C) and gd.stride(3) == 1 and gd.stride(2) % 4 == 0 and gd.data_ptr() % 16 == 0
                               and gd.stride(1) == W * gd.stride(2) and gd.stride(0) == H * gd.stride(1)):
        gd = None
    dx = gd if gd is not None else torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
    lddx = dx.stride(2)
    dres = torch.empty((B, H, W, C), dtype=torch.float32, device=dev) if (residual is not None and residual.needs_grad) else None
    if _BN_FUSED and M <= _BN_FUSED_MAXM and C <= 65536 and _bn_exchange_ok(dev):
        sync, ws = _bn_exchange(dev)
        # ReLU/ReLU6 mask: from the saved output, or - no residual, no fused dropout - recomputed from x (one tensor less)
        remask = (act != ACT_NONE and residual is None and gscale == 1.0) or lazy_out
        flags = getattr(dy, "_pp_rowflags", None) if _SPARSE_ROWS else None
        if flags is not None and flags.numel() == M:
            rc = L.pp_bn_bwd_fused_sparse(x.t.data_ptr(), ldx, dy.data_ptr(), lddy, None if remask else out.t.data_ptr(), ldya, act, M, C,
                                        mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                        dx.data_ptr(), lddx, dres.data_ptr() if dres is not None else None, C, float(gscale),
                                        beta.data_ptr(), ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), flags.data_ptr(), _stream())
            tape._keepalive.append(flags)
        else:
            rc = L.pp_bn_bwd_fused(x.t.data_ptr(), ldx, dy.data_ptr(), lddy, None if remask else out.t.data_ptr(), ldya, act, M, C,
                                   mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                   dx.data_ptr(), lddx, dres.data_ptr() if dres is not None else None, C, float(gscale),
                                   beta.data_ptr(), ws.data_ptr(), ws.numel(), sync.data_ptr(), sync.numel(), _stream())
        _lib.check(rc, "pp_bn_bwd_fused")
    else:
        assert gscale == 1.0, "a fused dropout is only created together with the single-launch BatchNorm"
        ws = _ws(_wsbytes("pp_colreduce_workspace_bytes", M, C), dev)
        rc = L.pp_bn_bwd(x.t.data_ptr(), ldx, dy.data_ptr(), lddy, out.t.data_ptr(), ldya, act, M, C, mean.data_ptr(), invstd.data_ptr(),
                         gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dx.data_ptr(), lddx,
                         dres.data_ptr() if dres is not None else None, C, ws.data_ptr(), ws.numel(), _stream())
        _lib.check(rc, "pp_bn_bwd")
    if gamma.requires_grad:
        tape.set_param_grad(gamma, dgamma)
    if beta.requires_grad:
        tape.set_param_grad(beta, dbeta)
    _acc(x, dx)
    if dres is not None:
        dres._pp_owned = True
        _acc(residual, dres)


def _bn_eval_bwd(tape: Tape, dy, x: Var, scale, act, residual, out: Var):
    raise RuntimeError("backward through eval-mode BatchNorm is not part of the PixelPick path "
                       "(query.py:148-158 and model.py:176-190 run eval under no_grad)")


# ------------------------------------------------------------------------------------------------- padding
def pad2d(tape: Tape, x: Var, pad_beg: int, pad_end: int) -> Var:
    """fixed_padding (mobilenet_v2.py:15-21): zero-pad H and W by (pad_beg, pad_end)."""
    B, H, W, C, ldx = _geom(x.t)
    Hp, Wp = H + pad_beg + pad_end, W + pad_beg + pad_end
    y = torch.empty((B, Hp, Wp, C), dtype=torch.float32, device=x.t.device)
    rc = _lib.lib().pp_pad2d(x.t.data_ptr(), ldx, B, H, W, C, pad_beg, pad_beg, Hp, Wp, y.data_ptr(), C, _stream())
    _lib.check(rc, "pp_pad2d")
    out = Var(y)
    tape.record(_pad_bwd, (x, pad_beg), out)
    return out


def _pad_bwd(tape: Tape, dy, x: Var, pad_beg):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    _, Hp, Wp, _, lddy = _geom(dy)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    # fuse the accumulation with an already-present gradient (the residual branch) into the crop
    add_ptr, ldadd = (None, 0)
    if x.grad is not None:
        _, _, _, _, ldadd = _geom(x.grad)
        add_ptr = x.grad.data_ptr()
    rc = _lib.lib().pp_crop2d_add(dy.data_ptr(), lddy, B, Hp, Wp, C, pad_beg, pad_beg, add_ptr, ldadd, dx.data_ptr(), C, H, W, _stream())
    _lib.check(rc, "pp_crop2d_add")
    x.grad = dx


# ------------------------------------------------------------------------------------------------- bilinear
def bilinear(tape: Tape, x: Var, size, align_corners: bool, scale_factor: float = 0.0, out_nchw: bool = False,
             dst: Optional[torch.Tensor] = None) -> Var:
    """F.interpolate(mode='bilinear').  out_nchw -> returns a [B,C,Ho,Wo] tensor (the model output)."""
    B, H, W, C, ldx = _geom(x.t)
    Ho, Wo = int(size[0]), int(size[1])
    dev = x.t.device
    if out_nchw:
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=dev)
        ldy = 0
    else:
        y = dst if dst is not None else torch.empty((B, Ho, Wo, C), dtype=torch.float32, device=dev)
        _, _, _, _, ldy = _geom(y)
    rc = _lib.lib().pp_bilinear_fwd(x.t.data_ptr(), ldx, B, H, W, C, y.data_ptr(), ldy, Ho, Wo, int(align_corners),
                                    float(scale_factor), float(scale_factor), int(out_nchw), _stream())
    _lib.check(rc, "pp_bilinear_fwd")
    out = Var(y)
    tape.record(_bilinear_bwd, (x, (Ho, Wo), align_corners, scale_factor, out_nchw), out)
    return out


def _bilinear_bwd(tape: Tape, dy, x: Var, size, align_corners, scale_factor, out_nchw):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    Ho, Wo = size
    if out_nchw:
        dy = dy.contiguous()
        lddy = 0
    else:
        _, _, _, _, lddy = _geom(dy)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    ws_ptr, ws_n = None, 0
    if not out_nchw and Ho >= 3 * H and Wo >= 3 * W:          # separable two-gather form for large up-sampling factors
        ws = _ws(_wsbytes("pp_bilinear_bwd_workspace_bytes", B, Ho, W, C), dy.device)
        ws_ptr, ws_n = ws.data_ptr(), ws.numel()
    rc = _lib.lib().pp_bilinear_bwd(dy.data_ptr(), lddy, B, Ho, Wo, C, dx.data_ptr(), C, H, W, int(align_corners),
                                    float(scale_factor), float(scale_factor), int(out_nchw), ws_ptr, ws_n, _stream())
    _lib.check(rc, "pp_bilinear_bwd")
    _acc(x, dx)


# ------------------------------------------------------------------------------------------------- pooling / broadcast
def global_avg_pool(tape: Tape, x: Var) -> Var:
    """nn.AdaptiveAvgPool2d((1,1)) -> [B,1,1,C]."""
    B, H, W, C, ldx = _geom(x.t)
    y = torch.empty((B, 1, 1, C), dtype=torch.float32, device=x.t.device)
    rc = _lib.lib().pp_image_colsum(x.t.data_ptr(), ldx, B, H * W, C, 1.0 / (H * W), y.data_ptr(), C, _stream())
    _lib.check(rc, "pp_image_colsum")
    out = Var(y)
    tape.record(_gap_bwd, (x,), out)
    return out


def _gap_bwd(tape: Tape, dy, x: Var):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    rc = _lib.lib().pp_image_broadcast(dy.data_ptr(), C, B, H * W, C, 1.0 / (H * W), dx.data_ptr(), C, _stream())
    _lib.check(rc, "pp_image_broadcast")
    _acc(x, dx)


def broadcast_hw(tape: Tape, v: Var, H: int, W: int, dst: Optional[torch.Tensor] = None) -> Var:
    """Bilinear (align_corners=True) upsample of a 1x1 map = broadcast (aspp.py:70)."""
    B, _, _, C, ldv = _geom(v.t)
    y = dst if dst is not None else torch.empty((B, H, W, C), dtype=torch.float32, device=v.t.device)
    _, _, _, _, ldy = _geom(y)
    rc = _lib.lib().pp_image_broadcast(v.t.data_ptr(), ldv, B, H * W, C, 1.0, y.data_ptr(), ldy, _stream())
    _lib.check(rc, "pp_image_broadcast")
    out = Var(y)
    tape.record(_broadcast_bwd, (v, H, W), out)
    return out


def _broadcast_bwd(tape: Tape, dy, v: Var, H, W):
    B, _, _, C, _ = _geom(v.t)
    _, _, _, _, lddy = _geom(dy)
    dv = torch.empty((B, 1, 1, C), dtype=torch.float32, device=dy.device)
    rc = _lib.lib().pp_image_colsum(dy.data_ptr(), lddy, B, H * W, C, 1.0, dv.data_ptr(), C, _stream())
    _lib.check(rc, "pp_image_colsum")
    _acc(v, dv)


# ------------------------------------------------------------------------------------------------- dropout
_dropout_counter = [0]
_dropout_seed_dev = [None]     # optional device int64[1]: per-step base seed read by the kernels at run time


def set_dropout_seed(seed: int):
    _dropout_counter[0] = int(seed) * 1000003


def set_dropout_device_seed(t: Optional[torch.Tensor]):
    """With a device seed word installed, the per-call salts restart at 0 every step (begin_step()) so that a captured
    graph and eager execution draw identical masks for the same word; the trainer bumps the word once per step."""
    _dropout_seed_dev[0] = t


def begin_step():
    if _dropout_seed_dev[0] is not None:
        _dropout_counter[0] = 0
    _STEP_EPOCH[0] += 1
    if _X3_WPRE and _X3_WPL:
        _prefetch_weight_planes()


def end_step():
    """The optimiser has run (or a recorded step was replayed): the weights moved, so the planes split at begin_step() are stale.
    Bumping the epoch invalidates them - a forward / backward outside a trainer step (FlatTrainer.forward_backward after
    train_step, evaluation with the tape on, a load_state_dict in between) splits inline from the current weights."""
    _STEP_EPOCH[0] += 1


# ---- bf16x3 weight planes off the step's critical path ---------------------------------------------------------------------------
# The MFMA-bound convolutions (SegmentHead, decoders.py:107-114) split their fp32 weights into three bf16 planes in front of every
# launch: x3_split_w_kernel, ~10 us, four times per step on the main queue.  Weights only change in the optimiser, so every layer that
# took the bf16x3 path in an earlier step (registered by conv2d / _conv2d_bwd) has both layouts re-split at begin_step() on the
# weight-gradient stream - idle during the forward - and the step's convolutions take the planes (pp_conv2d_*_pre2) after ONE stream
# wait.  Planes are valid for the step they were made in only (_STEP_EPOCH); a forward outside a trainer step splits inline as before.
# PIXELPICK_X3_WEIGHT_PREFETCH=0: off.
_X3_WPRE = os.environ.get("PIXELPICK_X3_WEIGHT_PREFETCH", "1") != "0"
_STEP_EPOCH = [0]
_X3_WPL = {}          # id(w) -> {"w": weight, "planes": {1: forward layout, 0: backward-data layout}, "epoch": step the planes belong to}
_X3_WPL_EVENT = {}    # device key -> (event recorded behind the splits, epoch, epoch the main stream last waited for it)


def _register_weight_planes(w: torch.Tensor, transpose: int):
    if not _X3_WPRE or w.dim() != 4:
        return
    ent = _X3_WPL.get(id(w))
    if ent is None or ent["w"]() is not w:
        ent = _X3_WPL[id(w)] = {"w": weakref.ref(w), "planes": {}, "epoch": -1}      # (weak: a new model per active-learning stage, model.py:250)
    ent["used"] = _STEP_EPOCH[0]
    if transpose not in ent["planes"]:
        kh, kw, Cin, Cout = w.shape
        nb = int(_lib.lib().pp_x3_weight_planes_bytes(kh * kw, Cin, Cout, transpose))
        ent["planes"][transpose] = torch.empty(nb, dtype=torch.uint8, device=w.device)
        ent["epoch"] = -1                       # a new layout: not filled yet


def _prefetch_weight_planes():
    L = _lib.lib()
    by_dev = {}
    for key, ent in list(_X3_WPL.items()):
        w = ent["w"]()
        if w is None or not w.is_cuda:
            del _X3_WPL[key]
            continue
        # only the weights the PREVIOUS step used (begin_step and end_step each advance the epoch): the registry is process-wide, and
        # another trainer's model, or a weight a stand-alone conv2d call registered once, would otherwise be re-split - two launches and
        # a plane buffer each - at every step of this one.  An idle weight's planes simply go stale; its next step splits inline once.
        if ent.get("used", -10) < _STEP_EPOCH[0] - 3:
            continue
        by_dev.setdefault(w.device, []).append((ent, w))
    for dev, ents in by_dev.items():
        main = _main_stream_obj()
        side = _side_stream(dev, 0)
        ev = _fork_event(dev)
        _lib.plan_note(ev.record, main)              # behind the previous step's optimiser
        _lib.plan_note(side.wait_event, ev)
        for ent, w in ents:
            kh, kw, Cin, Cout = w.shape
            for tr, planes in ent["planes"].items():
                _lib.check(L.pp_x3_split_weights(w.data_ptr(), kh * kw, Cin, Cout, tr, planes.data_ptr(), planes.numel(), side.cuda_stream),
                           "pp_x3_split_weights")
            ent["epoch"] = _STEP_EPOCH[0]
        dk = (dev.type, dev.index)
        rec = _X3_WPL_EVENT.get(dk)
        done = rec[0] if rec is not None else torch.cuda.Event()
        _lib.plan_note(done.record, side)
        _X3_WPL_EVENT[dk] = [done, _STEP_EPOCH[0], -1]


def _weight_planes(w: torch.Tensor, transpose: int):
    """Device pointer of this step's pre-split planes of w in the given layout (the main stream has waited for them), or None."""
    ent = _X3_WPL.get(id(w))
    if ent is not None and ent["w"]() is w:
        ent["used"] = _STEP_EPOCH[0]
    if ent is None or ent["w"]() is not w or ent["epoch"] != _STEP_EPOCH[0] or transpose not in ent["planes"]:
        return None
    rec = _X3_WPL_EVENT.get((w.device.type, w.device.index))
    if rec is None or rec[1] != _STEP_EPOCH[0]:
        return None
    if rec[2] != _STEP_EPOCH[0]:
        _lib.plan_note(_main_stream_obj().wait_event, rec[0])
        rec[2] = _STEP_EPOCH[0]
    return ent["planes"][transpose].data_ptr()


def dropout(tape: Tape, x: Var, p: float, training: bool) -> Var:
    """nn.Dropout.  Identity in eval mode / p == 0.  The mask is regenerated from the seed in backward."""
    if not training or p <= 0.0:
        return x
    B, H, W, C, ldx = _geom(x.t)
    _dropout_counter[0] += 1
    seed = (_dropout_counter[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.t.device)
    sd = _dropout_seed_dev[0]
    rc = _lib.lib().pp_dropout(x.t.data_ptr(), ldx, y.data_ptr(), C, B * H * W, C, float(p), seed,
                               sd.data_ptr() if sd is not None else None, _stream())
    _lib.check(rc, "pp_dropout")
    out = Var(y)
    tape.record(_dropout_bwd, (x, p, seed), out)
    return out


def _dropout_bwd(tape: Tape, dy, x: Var, p, seed):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    _, _, _, _, lddy = _geom(dy)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    sd = _dropout_seed_dev[0]
    rc = _lib.lib().pp_dropout(dy.data_ptr(), lddy, dx.data_ptr(), C, B * H * W, C, float(p), seed,
                               sd.data_ptr() if sd is not None else None, _stream())
    _lib.check(rc, "pp_dropout")
    _acc(x, dx)


def dropout2d(tape: Tape, x: Var, p: float, training: bool) -> Var:
    """nn.Dropout2d: whole channels of a sample are zeroed (mobilenet_v2.py:114-115,133-134).  Identity in eval mode."""
    if not training or p <= 0.0:
        return x
    B, H, W, C, ldx = _geom(x.t)
    _dropout_counter[0] += 1
    seed = (_dropout_counter[0] * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.t.device)
    sd = _dropout_seed_dev[0]
    rc = _lib.lib().pp_dropout2d(x.t.data_ptr(), ldx, y.data_ptr(), C, B, H * W, C, float(p), seed,
                                 sd.data_ptr() if sd is not None else None, _stream())
    _lib.check(rc, "pp_dropout2d")
    out = Var(y, needs_grad=x.needs_grad)
    tape.record(_dropout2d_bwd, (x, p, seed), out)
    return out


def _dropout2d_bwd(tape: Tape, dy, x: Var, p, seed):
    if not x.needs_grad:
        return
    B, H, W, C, _ = _geom(x.t)
    _, _, _, _, lddy = _geom(dy)
    dx = torch.empty((B, H, W, C), dtype=torch.float32, device=dy.device)
    sd = _dropout_seed_dev[0]
    rc = _lib.lib().pp_dropout2d(dy.data_ptr(), lddy, dx.data_ptr(), C, B, H * W, C, float(p), seed,
                                 sd.data_ptr() if sd is not None else None, _stream())
    _lib.check(rc, "pp_dropout2d")
    _acc(x, dx)


# ------------------------------------------------------------------------------------------------- concat (zero-copy)
def concat_alias(tape: Tape, buf: torch.Tensor, parts: Sequence[Var]) -> Var:
    """torch.cat(dim=channel) without a copy: every part was produced directly into its channel slice of
    `buf` (dst= of the producer).  Backward hands each producer its slice of the gradient."""
    off = 0
    for p in parts:
        c = p.t.shape[3]
        assert p.t.data_ptr() == buf[..., off:off + c].data_ptr(), "concat_alias: part not produced in place"
        off += c
    assert off == buf.shape[3]
    out = Var(buf)
    tape.record(_concat_bwd, (tuple(parts),), out)
    return out


def _concat_bwd(tape: Tape, dy, parts):
    off = 0
    for p in parts:
        c = p.t.shape[3]
        _acc(p, dy[..., off:off + c])
        off += c


# ------------------------------------------------------------------------------------------------- loss
def cross_entropy_nchw(logits: torch.Tensor, target: torch.Tensor, ignore_index: int, want_grad: bool = True, sparse: bool = True):
    """F.cross_entropy(logits [B,C,H,W], target [B,H,W] int64, ignore_index) -> (loss [1], dlogits | None).
    sparse: the caller knows that only a few pixels are labelled (model.py:108-110) - the gradient is marked so that the backward
    behind it may skip its zero rows; False for densely labelled targets (every row would be flagged: the dense kernels are faster)."""
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4
    B, C, H, W = logits.shape
    assert logits.stride(3) == 1 and logits.stride(2) == W, "logits planes must be contiguous"
    target = target.to(logits.device, torch.int64).contiguous()
    L = _lib.lib()
    dev = logits.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.float32, device=dev)
    dl = torch.empty((B, C, H, W), dtype=torch.float32, device=dev) if want_grad else None
    ws = _ws(_wsbytes("pp_sparse_ce_workspace_bytes"), dev)
    rc = L.pp_sparse_ce_fwd_bwd(logits.data_ptr(), B, C, H * W, logits.stride(0), logits.stride(1), target.data_ptr(),
                                int(ignore_index), loss.data_ptr(), count.data_ptr(), None,
                                dl.data_ptr() if dl is not None else None, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "pp_sparse_ce_fwd_bwd")
    if dl is not None and sparse:
        dl._pp_sparse_rows = True        # zero wherever target == ignore_index (model.py:113-119: all but the labelled pixels)
    return loss, dl


def cross_entropy_lowres(low: torch.Tensor, size, target: torch.Tensor, ignore_index: int, align_corners: bool = True,
                         want_grad: bool = True, sparse: bool = True):
    """F.cross_entropy(F.interpolate(low, size, 'bilinear', align_corners), target, ignore_index) and its gradient w.r.t.
    `low`, without the full-size logits (deeplab.py:55-56 + model.py:116).  low [B,h,w,C] channels-last (the classifier
    output), target [B,H,W] int64 -> (loss [1], dlow [B,h,w,C] | None)."""
    assert low.is_cuda and low.dtype == torch.float32 and low.dim() == 4 and low.stride(3) == 1
    B, h, w, C = low.shape
    assert low.stride(1) == w * low.stride(2) and low.stride(0) == h * low.stride(1)
    H, W = int(size[0]), int(size[1])
    target = target.to(low.device, torch.int64).contiguous()
    assert tuple(target.shape) == (B, H, W)
    dev = low.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.float32, device=dev)
    dlow = torch.empty((B, h, w, C), dtype=torch.float32, device=dev) if want_grad else None
    ws = _ws(_wsbytes("pp_sparse_ce_lowres_workspace_bytes"), dev)
    rc = _lib.lib().pp_sparse_ce_lowres_fwd_bwd(low.data_ptr(), low.stride(2), B, C, h, w, H, W, int(bool(align_corners)),
                                                target.data_ptr(), int(ignore_index), loss.data_ptr(), count.data_ptr(), None,
                                                dlow.data_ptr() if dlow is not None else None, C, ws.data_ptr(), ws.numel(),
                                                _stream())
    _lib.check(rc, "pp_sparse_ce_lowres_fwd_bwd")
    if dlow is not None and _SPARSE_ROWS and sparse:
        # 20 labelled pixels per image (model.py:113-119) touch <= 4 low-resolution rows each: the gradient is zero in all other rows.
        # The flags ride on the tensor; the classifier's backward-data and the BatchNorm backward behind it skip the zero rows.
        flags = torch.empty(B * h * w, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().pp_row_flags(dlow.data_ptr(), C, B * h * w, C, flags.data_ptr(), _stream()), "pp_row_flags")
        dlow._pp_rowflags = flags
    return loss, dlow
