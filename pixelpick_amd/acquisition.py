"""Host-side functional API over the acquisition entry points of the C ABI.

Everything here only marshals torch device tensors (pointers, strides, current stream) into
include/pixelpick_hip.h calls; all arithmetic runs in the hand-written HIP kernels (csrc/acq.hip).
"""
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

STRATEGY_ID = {"entropy": 0, "least_confidence": 1, "margin_sampling": 2, "margin": 2}
# "random" (query.py:242-244) has no score kernel: its map is host RNG output (torch.rand on the CPU, as the reference);
# only its exclusion fill / top-k direction are listed here and the selection itself runs through topk_select().
LARGEST = {"entropy": True, "least_confidence": True, "margin_sampling": False, "margin": False, "random": False}
FILL = {"entropy": 0.0, "least_confidence": 0.0, "margin_sampling": 1.0, "margin": 1.0, "random": 1.0}


def strategy_id(strategy: str) -> int:
    try:
        return STRATEGY_ID[strategy]
    except KeyError:
        raise ValueError(f"no score kernel for query strategy {strategy!r} (kernels: {sorted(STRATEGY_ID)}; 'random' maps "
                         f"come from the host RNG and go through topk_select)") from None


def _require_cuda_f32(t: torch.Tensor, name: str, ndim: int):
    if not isinstance(t, torch.Tensor) or t.ndim != ndim:
        raise ValueError(f"{name} must be a {ndim}-d tensor")
    if not t.is_cuda:
        raise _lib.PixelPickHipError(f"{name} must live on the GPU: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")


def _exclude_u8(exclude, B, H, W, device):
    if exclude is None:
        return None
    if isinstance(exclude, np.ndarray):
        exclude = np.ascontiguousarray(exclude)       # also resolves negative strides (flipped views)
    ex = torch.as_tensor(exclude)
    if ex.dtype == torch.bool:
        # reinterpret, do not convert: a torch CPU op on a 131 K-element mask costs milliseconds on a many-core host
        # (OpenMP fork/join per op) - it was 90 % of the per-image time of an acquisition round
        ex = ex.contiguous().view(torch.uint8)
    elif ex.dtype != torch.uint8:
        ex = (ex != 0).contiguous().view(torch.uint8)
    ex = ex.to(device, non_blocking=True).reshape(B, H, W).contiguous()
    return ex


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


REFERENCE_ORDER = 0x100      # PP_ACQ_REFERENCE_ORDER (include/pixelpick_hip.h): OR-ed into the strategy id of one call


def score_topk(logits: torch.Tensor, exclude, strategy: str, k: int, return_map: bool = False, reference_order: bool = False
               ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """softmax -> score -> exclusion -> per-image top-k in one pass (query.py:190-204,57-61).
    reference_order: score in query.py:190,230's own operation order (p = exp(x - m) / S, then sum(-p log p)) instead of the
    default algebraic form - a per-call flag, about half the rate.

    logits [B,C,H,W] f32 on the GPU with any strides (NCHW, channels_last, cropped view).
    Returns (idx int32 [B,k] flat h*W+w value-sorted, val f32 [B,k], map f32 [B,H,W] | None).
    """
    _require_cuda_f32(logits, "logits", 4)
    B, C, H, W = logits.shape
    L = _lib.lib()
    dev = logits.device
    ex = _exclude_u8(exclude, B, H, W, dev)
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    val = torch.empty((B, k), dtype=torch.float32, device=dev)
    omap = torch.empty((B, H, W), dtype=torch.float32, device=dev) if return_map else None
    nbytes = L.pp_acq_workspace_bytes(B, C, H, W, k)
    ws = _ws(nbytes, dev)
    sB, sC, sH, sW = logits.stride()
    with torch.cuda.device(dev):
        rc = L.pp_acq_score_topk(logits.data_ptr(), B, C, H, W, sB, sC, sH, sW,
                                 ex.data_ptr() if ex is not None else None,
                                 strategy_id(strategy) | (REFERENCE_ORDER if reference_order else 0), k,
                                 idx.data_ptr(), val.data_ptr(), omap.data_ptr() if omap is not None else None,
                                 ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_acq_score_topk")
    return idx, val, omap


def score_map(logits: torch.Tensor, exclude, strategy: str, reference_order: bool = False) -> torch.Tensor:
    """[B,C,H,W] logits -> [B,H,W] uncertainty map after exclusion (query.py:190-201)."""
    _require_cuda_f32(logits, "logits", 4)
    B, C, H, W = logits.shape
    L = _lib.lib()
    dev = logits.device
    ex = _exclude_u8(exclude, B, H, W, dev)
    omap = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    sB, sC, sH, sW = logits.stride()
    with torch.cuda.device(dev):
        rc = L.pp_acq_score_map(logits.data_ptr(), B, C, H, W, sB, sC, sH, sW,
                                ex.data_ptr() if ex is not None else None,
                                strategy_id(strategy) | (REFERENCE_ORDER if reference_order else 0),
                                omap.data_ptr(), _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_acq_score_map")
    return omap


def uncertainty_from_prob(prob: torch.Tensor, strategy: str) -> torch.Tensor:
    """UncertaintySampler.__call__ on an already-softmaxed tensor (query.py:229-239,246-247)."""
    _require_cuda_f32(prob, "prob", 4)
    B, C, H, W = prob.shape
    L = _lib.lib()
    dev = prob.device
    omap = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    sB, sC, sH, sW = prob.stride()
    with torch.cuda.device(dev):
        rc = L.pp_uncertainty_from_prob(prob.data_ptr(), B, C, H, W, sB, sC, sH, sW, strategy_id(strategy),
                                        omap.data_ptr(), _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_uncertainty_from_prob")
    return omap


def mc_accumulate_(logits: torch.Tensor, prob_out: Optional[torch.Tensor], uc_out: Optional[torch.Tensor], strategy: str,
                   scale: float, accumulate: bool = True):
    """MC-dropout accumulation (query.py:181-187) in one kernel pass over logits [T,C,H,W]: prob_out [C,H,W] (+)= scale *
    sum_t softmax(logits[t]), uc_out [H,W] (+)= scale * sum_t strategy-score(softmax(logits[t])).  Either may be None."""
    _require_cuda_f32(logits, "logits", 4)
    T, C, H, W = logits.shape
    for t, shape in ((prob_out, (C, H, W)), (uc_out, (H, W))):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous() or t.dtype != torch.float32 or t.device != logits.device):
            raise ValueError(f"output must be a contiguous float32 {shape} tensor on the logits' device")
    sT, sC, sH, sW = logits.stride()
    sid = strategy_id(strategy) if uc_out is not None else 0
    with torch.cuda.device(logits.device):
        rc = _lib.lib().pp_acq_softmax_sum(logits.data_ptr(), T, C, H, W, sT, sC, sH, sW,
                                           prob_out.data_ptr() if prob_out is not None else None,
                                           uc_out.data_ptr() if uc_out is not None else None, sid, float(scale),
                                           int(bool(accumulate)), _lib.current_stream_ptr(logits.device))
    _lib.check(rc, "pp_acq_softmax_sum")


def topk_select(scores: torch.Tensor, k: int, largest: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """scores [B,N] f32 -> (idx int32 [B,k], val f32 [B,k]); uc_map.flatten().topk (query.py:57-61)."""
    _require_cuda_f32(scores, "scores", 2)
    scores = scores.contiguous()
    B, N = scores.shape
    L = _lib.lib()
    dev = scores.device
    idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    val = torch.empty((B, k), dtype=torch.float32, device=dev)
    ws = _ws(L.pp_topk_workspace_bytes(B, N, k), dev)
    with torch.cuda.device(dev):
        rc = L.pp_topk_select(scores.data_ptr(), B, N, k, int(bool(largest)), idx.data_ptr(), val.data_ptr(),
                              ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_topk_select")
    return idx, val


# ---- acquisition from the low-resolution classifier output -------------------------------
def _lowres_geom(low: torch.Tensor, size, crop):
    _require_cuda_f32(low, "low", 4)
    if low.stride(3) != 1 or low.stride(1) != low.shape[2] * low.stride(2) or low.stride(0) != low.shape[1] * low.stride(1):
        raise ValueError("low must be a dense channels-last [B,h,w,C] tensor (a channel slice of one is fine)")
    B, h, w, C = low.shape
    H, W = int(size[0]), int(size[1])
    Hc, Wc = (H, W) if crop is None else (int(crop[0]), int(crop[1]))
    return B, h, w, C, low.stride(2), H, W, Hc, Wc


def score_topk_lowres(low: torch.Tensor, size, exclude, strategy: str, k: int, crop=None, align_corners: bool = True,
                      return_map: bool = False) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]:
    """F.interpolate(low, size, 'bilinear', align_corners)[:, :, :crop_h, :crop_w] -> softmax -> score -> exclusion ->
    top-k in ONE launch, never writing the full-resolution logits (deeplab.py:55-56 + query.py:190-204,57-61).

    low [B,h,w,C] f32 channels-last on the GPU (the classifier output as the engine keeps it).  Returns
    (idx int32 [B,k] flat y*crop_w+x, val f32 [B,k], map f32 [B,crop_h,crop_w] | None); k == 0 -> (None, None, map)."""
    B, h, w, C, ldx, H, W, Hc, Wc = _lowres_geom(low, size, crop)
    L = _lib.lib()
    dev = low.device
    ex = _exclude_u8(exclude, B, Hc, Wc, dev)
    want_map = return_map or k == 0
    idx = torch.empty((B, k), dtype=torch.int32, device=dev) if k else None
    val = torch.empty((B, k), dtype=torch.float32, device=dev) if k else None
    omap = torch.empty((B, Hc, Wc), dtype=torch.float32, device=dev) if want_map else None
    ws = _ws(L.pp_acq_lowres_workspace_bytes(B, C, Hc, Wc, k), dev) if k else None
    with torch.cuda.device(dev):
        rc = L.pp_acq_lowres_score_topk(low.data_ptr(), ldx, B, C, h, w, H, W, int(bool(align_corners)), Hc, Wc,
                                        ex.data_ptr() if ex is not None else None, strategy_id(strategy), k,
                                        idx.data_ptr() if k else None, val.data_ptr() if k else None,
                                        omap.data_ptr() if omap is not None else None,
                                        ws.data_ptr() if k else None, ws.numel() if k else 0, _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_acq_lowres_score_topk")
    return idx, val, omap


def score_at_lowres(low: torch.Tensor, size, img_idx, pix_idx, strategy: str = "entropy", crop=None,
                    align_corners: bool = True) -> torch.Tensor:
    """The strategy's score (no exclusion) at listed pixels of the interpolated map: pixel i = image img_idx[i], flat
    index pix_idx[i] = y*crop_w + x.  -> f32 [n] on the GPU.  (QueryStats' entropy at the queried pixels, query.py:262-266.)"""
    B, h, w, C, ldx, H, W, Hc, Wc = _lowres_geom(low, size, crop)
    dev = low.device
    ii = torch.as_tensor(img_idx).to(torch.int32).to(dev).contiguous()
    pp = torch.as_tensor(pix_idx).to(torch.int32).to(dev).contiguous()
    if ii.shape != pp.shape or ii.ndim != 1:
        raise ValueError("img_idx and pix_idx must be 1-d and of equal length")
    n = ii.numel()
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        rc = _lib.lib().pp_acq_lowres_score_at(low.data_ptr(), ldx, B, C, h, w, H, W, int(bool(align_corners)), Hc, Wc,
                                               strategy_id(strategy), ii.data_ptr(), pp.data_ptr(), n, out.data_ptr(),
                                               _lib.current_stream_ptr(dev))
    _lib.check(rc, "pp_acq_lowres_score_at")
    return out
