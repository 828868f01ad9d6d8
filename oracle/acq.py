"""ctypes wrapper over oracle/acq_oracle.c plus a torch-CPU "port" of the reference acquisition
path used for the bench's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status: PINNED against tests/golden/acq_*.npz (generated from the imported reference).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
STRATEGY_ID = {"entropy": 0, "least_confidence": 1, "margin_sampling": 2, "margin": 2}


def build(force=False):
    src = os.path.join(_HERE, "acq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        i64, p = ctypes.c_int64, ctypes.c_void_p
        _lib.orc_acq_score_map.argtypes = [p] + [i64] * 8 + [ctypes.c_int, p]
        _lib.orc_acq_apply_exclude.argtypes = [p, p, i64, ctypes.c_int]
        _lib.orc_topk.argtypes = [p, i64, i64, ctypes.c_int, p, p]
        _lib.orc_acq_score_topk.argtypes = [p] + [i64] * 8 + [p, ctypes.c_int, i64, p, p, p]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def score_map(logits: np.ndarray, strategy: str) -> np.ndarray:
    """logits [B,C,H,W] float32, any strides -> [B,H,W] float32 (query.py:190,229-239)."""
    assert logits.dtype == np.float32 and logits.ndim == 4
    B, C, H, W = logits.shape
    sB, sC, sH, sW = (s // 4 for s in logits.strides)
    out = np.empty((B, H, W), dtype=np.float32)
    rc = lib().orc_acq_score_map(_ptr(logits), B, C, H, W, sB, sC, sH, sW, STRATEGY_ID[strategy], _ptr(out))
    assert rc == 0
    return out


def apply_exclude(m: np.ndarray, exclude: np.ndarray, strategy: str) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.float32).copy()
    ex = np.ascontiguousarray(exclude, dtype=np.uint8)
    lib().orc_acq_apply_exclude(_ptr(m), _ptr(ex), m.size, STRATEGY_ID[strategy])
    return m


def topk(scores: np.ndarray, k: int, largest: bool):
    """scores [n] -> (idx int32[k], val float32[k]); ties -> lower index, NaN first if largest."""
    s = np.ascontiguousarray(scores.reshape(-1), dtype=np.float32)
    idx = np.empty(k, dtype=np.int32)
    val = np.empty(k, dtype=np.float32)
    rc = lib().orc_topk(_ptr(s), s.size, k, int(largest), _ptr(idx), _ptr(val))
    assert rc == 0
    return idx, val


def score_topk(logits: np.ndarray, exclude, strategy: str, k: int, want_map=False):
    B, C, H, W = logits.shape
    sB, sC, sH, sW = (s // 4 for s in logits.strides)
    idx = np.empty((B, k), dtype=np.int32)
    val = np.empty((B, k), dtype=np.float32)
    m = np.empty((B, H, W), dtype=np.float32) if want_map else None
    ex = np.ascontiguousarray(exclude, dtype=np.uint8) if exclude is not None else None
    rc = lib().orc_acq_score_topk(_ptr(logits), B, C, H, W, sB, sC, sH, sW, _ptr(ex),
                                  STRATEGY_ID[strategy], k, _ptr(idx), _ptr(val), _ptr(m))
    assert rc == 0
    return (idx, val, m) if want_map else (idx, val)


def bilinear_resize(x: np.ndarray, size, align_corners: bool = True) -> np.ndarray:
    """x [B,C,h,w] float32 -> [B,C,H,W]: F.interpolate(x, size, mode='bilinear', align_corners) as deeplab.py:55-56
    calls it.  The arithmetic lives in torch (ATen upsample_bilinear2d); this restates its published rule in fp32:
      source index  src = dst * (in-1)/(out-1)                (align_corners; 0 when out == 1)
                    src = max((dst+0.5) * in/out - 0.5, 0)    (otherwise)
      i0 = floor(src) clamped to in-1, i1 = min(i0+1, in-1), l1 = src - i0, l0 = 1 - l1
      out = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11)
    pinned by tests/golden/acq_lowres.npz (s1_pred) and, in the CPU suite, against F.interpolate itself."""
    assert x.dtype == np.float32 and x.ndim == 4
    h, w = x.shape[2:]
    H, W = int(size[0]), int(size[1])

    def coords(n_in, n_out):
        d = np.arange(n_out, dtype=np.float32)
        if align_corners:
            scale = np.float32((n_in - 1) / (n_out - 1)) if n_out > 1 else np.float32(0)
            src = scale * d
        else:
            scale = np.float32(n_in / n_out)
            src = np.maximum(scale * (d + np.float32(0.5)) - np.float32(0.5), np.float32(0)).astype(np.float32)
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, np.float32(1) - l1, l1

    r0, r1, h0, h1 = coords(h, H)
    c0, c1, w0, w1 = coords(w, W)
    top = w0[None, None, None, :] * x[:, :, r0][:, :, :, c0] + w1[None, None, None, :] * x[:, :, r0][:, :, :, c1]
    bot = w0[None, None, None, :] * x[:, :, r1][:, :, :, c0] + w1[None, None, None, :] * x[:, :, r1][:, :, :, c1]
    return (h0[None, None, :, None] * top + h1[None, None, :, None] * bot).astype(np.float32)


def lowres_score_topk(low: np.ndarray, size, exclude, strategy: str, k: int, crop=None, align_corners: bool = True,
                      want_map=False):
    """SURVEY.md §8f-1 path on the CPU: low [B,C,h,w] classifier logits -> deeplab.py:55-56 interpolate -> query.py:190
    [:h,:w] crop -> score_topk above.  Flat indices refer to the crop."""
    pred = bilinear_resize(low, size, align_corners)
    if crop is not None:
        pred = np.ascontiguousarray(pred[:, :, :crop[0], :crop[1]])
    return score_topk(pred, exclude, strategy, k, want_map=want_map)


# ----------------------------------------------------------------------------------------------
# torch-CPU port of the reference acquisition path (same torch ops the reference calls), used only
# to time the CPU baseline on the GPU box's host cores (bench.py cpu_baseline.kind == "port").
def torch_port_acquire(logits_t, exclude_t, strategy: str, k: int):
    """logits_t [1,C,H,W] CPU tensor; exclude_t [H,W] bool CPU tensor.  Mirrors query.py:190-204,57-61."""
    import torch
    import torch.nn.functional as F
    prob = F.softmax(logits_t, dim=1)                                        # query.py:190
    if strategy == "entropy":
        uc = (-prob * torch.log(prob)).sum(dim=1)                             # query.py:230
    elif strategy == "least_confidence":
        uc = 1.0 - prob.max(dim=1)[0]                                         # query.py:234
    else:
        top2 = prob.topk(k=2, dim=1).values                                   # query.py:238-239
        uc = (top2[:, 0] - top2[:, 1]).abs()
    uc = uc.squeeze(0)
    uc[exclude_t] = 0.0 if strategy in ("entropy", "least_confidence") else 1.0   # query.py:198-201
    return uc.flatten().topk(k, largest=strategy in ("entropy", "least_confidence")).indices  # query.py:57-61
