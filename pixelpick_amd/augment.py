"""Training-time data augmentation on the GPU — datasets/base_dataset.py:48-141 (`_geometric_augmentations`,
`_photometric_augmentations`, `GaussianBlur`) and the `TF.normalize(TF.to_tensor(x))` of `__getitem__` (:181), SURVEY.md §8f
rank 4.  The reference does this per image in DataLoader workers through torchvision wrappers around PIL and cv2; here the
host only draws the random parameters (same generators, same order) and builds small index / coefficient tables, and the
pixels never leave the device: raw uint8 image + label map + query mask in, normalised float batch + int64 labels + masks out.

Random draws per image, in the reference's order:
  random.uniform(0.5, 2.0)                      scale                           base_dataset.py:60
  random.randint(0, h-ch), randint(0, w-cw)     crop origin (after padding)     :89
  random.random() > 0.5                         horizontal flip                 :103
  torch.rand(1) <= 0.8                          RandomApply(ColorJitter)        :126 (torchvision: `if self.p < torch.rand(1): skip`)
  torch.randperm(4), 4 x uniform_               op order; brightness, contrast, saturation in [0.2, 1.8], hue in [-0.2, 0.2]
  torch.rand(1) < 0.2                           RandomGrayscale                 :129
  np.random.random_sample() < 0.5, sigma        GaussianBlur, sigma in [0.1, 2) :207-210

Arithmetic parity: the geometric path, colour jitter and grayscale reproduce PIL's uint8 results bit for bit (tests compare
with PIL itself, which torchvision's PIL backend only wraps); the blur follows cv2's documented algorithm (float separable
kernel from getGaussianKernel, BORDER_REFLECT_101, round-half-even) but cv2 is not available offline to pin it.
There is no CPU fallback: the kernels live in csrc/augment.hip behind include/pixelpick_hip.h.
"""
import ctypes
import math
import random as _py_random
import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

_PRECISION_BITS = 22


# ------------------------------------------------------------------------------------------------- tables (host, tiny)
def pil_bilinear_tables(in_size: int, out_size: int):
    """libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1):
    -> bounds int32 [out,2] (first source index, count), kk int32 [out,ksize] (22-bit fixed point), ksize."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncation of a positive value
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    ss = 1.0 / filterscale
    w = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):
        a = np.abs((x + xmin - center + 0.5) * ss)
        wx = np.where((a < 1.0) & (x < xmax), 1.0 - a, 0.0)
        w[:, x] = wx
        ww = ww + wx                                   # sequential double sum, as the C loop
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = (0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


def identity_tables(n: int):
    bounds = np.stack([np.arange(n), np.ones(n, dtype=np.int64)], axis=1).astype(np.int32)
    kk = np.full((n, 1), 1 << _PRECISION_BITS, dtype=np.int32)
    return bounds, kk, 1


def pil_nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """PIL NEAREST resize (Geometry.c ImagingScaleAffine): index = int(xo), xo starting at a/2 and advanced by += a in double."""
    a0 = in_size / out_size
    steps = np.full(out_size, a0, dtype=np.float64)
    steps[0] = a0 * 0.5
    xo = np.cumsum(steps)                               # sequential accumulation, as the C loop
    return np.minimum(xo.astype(np.int64), in_size - 1).astype(np.int32)


def torch_nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """F.interpolate(mode='nearest') (what TF.resize does to a uint8 TENSOR): min(floor(dst * float32(in/out)), in-1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    return np.minimum(np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64), in_size - 1).astype(np.int32)


def cv2_gaussian_kernel(ksize: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0."""
    scale2x = -0.5 / (sigma * sigma)
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    cf = np.exp(scale2x * x * x).astype(np.float32)
    s = 0.0
    for v in cf:
        s += float(v)
    return (cf.astype(np.float64) * (1.0 / s)).astype(np.float32)


def cv2_gaussian_kernel_q8(ksize: int, sigma: float) -> np.ndarray:
    """The 8.8 fixed-point taps of cv2.GaussianBlur for 8-bit images (OpenCV >= 3.4.2 / 4.x, smooth.dispatch.cpp:
    getGaussianKernelBitExact in double, then getGaussianKernelFixedPoint_ED - rounding with error diffusion from the outside in,
    centre tap = 256 - the rest, so the taps sum to exactly 1.0), as uint16 for pp_aug_blur_q8."""
    if ksize % 2 != 1 or ksize < 1 or not sigma > 0:
        raise ValueError("cv2_gaussian_kernel_q8: odd ksize and sigma > 0")
    n2 = (ksize - 1) // 2
    x = (2.0 * np.arange(n2, dtype=np.float64) + (1 - ksize))                   # 2 * (i - (n-1)/2), exact
    t = np.exp((x * x) * (np.float64(-0.125) / (np.float64(sigma) * np.float64(sigma))))
    total = np.float64(0.0)
    for v in t:                                                                  # OpenCV's summation order
        total = total + v
    mul1 = np.float64(1.0) / (total * 2.0 + 1.0)
    out = np.zeros(ksize, dtype=np.int64)
    err, acc = np.float64(0.0), 0
    for i in range(n2):
        adj = (t[i] * mul1) * 256.0 + err
        v0 = int(np.rint(adj))
        err = adj - v0
        out[i] = out[ksize - 1 - i] = v0
        acc += v0
    out[n2] = 256 - 2 * acc
    return out.astype(np.uint16)


# ------------------------------------------------------------------------------------------------- the augmenter
class DeviceAugmenter:
    def __init__(self, crop_size: Sequence[int], mean: Sequence[float], std: Sequence[float], ignore_index: int,
                 geometric: Optional[dict] = None, photometric: Optional[dict] = None, device="cuda:0"):
        self.crop_size = (int(crop_size[0]), int(crop_size[1]))
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.mean_val = tuple((np.array(mean) * 255.0).astype(np.uint8).tolist())          # cityscapes.py:51
        self.ignore_index = int(ignore_index)
        self.geometric = dict(random_scale=True, crop=True, random_hflip=True)
        self.geometric.update(geometric or {})
        if not self.geometric["crop"]:
            raise ValueError("DeviceAugmenter batches its output: geometric['crop'] must be on (every training config of the reference)")
        self.photometric = dict(random_color_jitter=True, random_grayscale=True, random_gaussian_blur=True)
        self.photometric.update(photometric or {})
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PixelPickHipError("DeviceAugmenter runs on the GPU only (no CPU fallback)")
        self._mean_c = (ctypes.c_float * 3)(*self.mean)
        self._std_c = (ctypes.c_float * 3)(*self.std)
        # resampling tables are pure functions of (rule, in_size, out_size): built once on the host, kept ON THE DEVICE (a few KB
        # each; a random scale in [0.5, 2] of one dataset size yields a few hundred distinct sizes per axis), so that a warm
        # batch loop uploads nothing but the images
        self._tables = {}
        self.n_table_uploads = 0
        self.blur_arithmetic = "fixed"    # "float": the float32 separable filter of cv2 < 3.4.2 (pp_aug_blur)
        self._rng = None                  # None: the process-wide generators, as the reference's datasets draw (single-process parity)

    def use_private_rng(self, seed: int):
        """Draw the augmentation parameters from generators of this augmenter's own (python / numpy / torch, all seeded with `seed`)
        instead of the process-wide ones.  Data-parallel training: every rank holds IDENTICAL process-wide streams (the acquisition
        round needs that), which would give every rank the same scale / crop / flip / jitter sequence on its different shard;
        `Model` seeds each rank's augmenter from (args.seed, rank) - dist_utils.augment_seed - and leaves the global streams alone."""
        g = torch.Generator()
        g.manual_seed(int(seed))
        self._rng = (random.Random(int(seed)), np.random.RandomState(int(seed) % (1 << 32)), g)
        return self

    @classmethod
    def from_args(cls, args, device="cuda:0", crop_size=None):
        """The augmenter the reference's dataset classes configure from `args` (cityscapes.py:44-58, camvid.py, voc.py:
        args.augmentations["geometric" / "photometric"], args.mean / args.std, args.ignore_index; crop size by dataset)."""
        if crop_size is None:
            crop_size = getattr(args, "crop_size", None)
        if crop_size is None:
            ds = getattr(args, "dataset_name", "cs")
            crop_size = {"cs": (256, 512) if getattr(args, "downsample", 4) == 4 else (512, 1024), "cv": (360, 480), "voc": (320, 320)}[ds]
        aug = getattr(args, "augmentations", None) or {}
        return cls(crop_size, args.mean, args.std, args.ignore_index, geometric=aug.get("geometric"), photometric=aug.get("photometric"),
                   device=device)

    def _table(self, kind: str, in_size: int, out_size: int):
        key = (kind, in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            if kind == "bilinear":
                b, k, ks = pil_bilinear_tables(in_size, out_size) if in_size != out_size else identity_tables(in_size)
                t = (self._dev_i32(b), self._dev_i32(k), ks)
            elif kind == "pil_nearest":
                t = self._dev_i32(pil_nearest_table(in_size, out_size))
            else:
                t = self._dev_i32(torch_nearest_table(in_size, out_size))
            if len(self._tables) >= 8192:
                self._tables.clear()
            self._tables[key] = t
            self.n_table_uploads += 1
        return t

    # ---- random parameters: same generators and order as the reference -------------------------------------------
    def draw(self, h: int, w: int) -> dict:
        p = {"h": h, "w": w}
        ch, cw = self.crop_size
        random, nprand, g = self._rng if self._rng is not None else (_py_random, np.random, None)
        if self.geometric["random_scale"]:
            rs = random.uniform(0.5, 2.0)
            p["w_rs"], p["h_rs"] = int(w * rs), int(h * rs)
        else:
            p["w_rs"], p["h_rs"] = w, h
        pad_h, pad_w = max(ch - p["h_rs"], 0), max(cw - p["w_rs"], 0)
        hp, wp = p["h_rs"] + pad_h, p["w_rs"] + pad_w
        p["start_h"], p["start_w"] = random.randint(0, hp - ch), random.randint(0, wp - cw)
        p["flip"] = bool(self.geometric["random_hflip"] and random.random() > 0.5)
        ops = []
        if self.photometric["random_color_jitter"]:
            if not (0.8 < float(torch.rand(1, generator=g))):       # RandomApply: `if self.p < torch.rand(1): return img`
                fn_idx = torch.randperm(4, generator=g).tolist()
                b = float(torch.empty(1).uniform_(0.2, 1.8, generator=g))
                c = float(torch.empty(1).uniform_(0.2, 1.8, generator=g))
                s = float(torch.empty(1).uniform_(0.2, 1.8, generator=g))
                hue = float(torch.empty(1).uniform_(-0.2, 0.2, generator=g))
                fac = {0: b, 1: c, 2: s, 3: hue}
                ops += [(fn, fac[fn]) for fn in fn_idx]
        if self.photometric["random_grayscale"]:
            if float(torch.rand(1, generator=g)) < 0.2:
                ops.append((4, 0.0))
        p["ops"] = ops
        p["blur"] = None
        if self.photometric["random_gaussian_blur"]:
            ks = int((0.1 * min(cw, ch) // 2 * 2) + 1)               # on the cropped image: x.size after the crop
            if nprand.random_sample() < 0.5:
                sigma = (2.0 - 0.1) * nprand.random_sample() + 0.1
                p["blur"] = (ks, float(sigma))
        return p

    # ---- one image -------------------------------------------------------------------------------------------------
    def _dev_i32(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device, non_blocking=True)

    def apply(self, x_u8: torch.Tensor, y_u8: Optional[torch.Tensor], q_u8: Optional[torch.Tensor], p: dict,
              x_out: torch.Tensor, y_out: Optional[torch.Tensor], q_out: Optional[torch.Tensor],
              lq_u8: Optional[torch.Tensor] = None, lq_out: Optional[torch.Tensor] = None):
        """x_u8 [H,W,3] uint8, y_u8 / q_u8 / lq_u8 [H,W] uint8 on the device; outputs are slots of the batch tensors.
        lq: the human-labelled query map (base_dataset.py:67-68,85-86,99-101,113-114): a TENSOR in the reference, so it is resized
        with torch's nearest rule like the query mask, but padded with ignore_index like the label map."""
        L = _lib.lib()
        st = _lib.current_stream_ptr(self.device)
        h, w, h_rs, w_rs = p["h"], p["w"], p["h_rs"], p["w_rs"]
        ch, cw = self.crop_size
        assert tuple(x_u8.shape) == (h, w, 3) and x_u8.dtype == torch.uint8 and x_u8.is_contiguous() and x_u8.is_cuda
        # horizontal pass (skipped by PIL when the width is unchanged)
        if w_rs != w:
            bw_d, kw_d, ksw = self._table("bilinear", w, w_rs)
            tmp = torch.empty((h, w_rs, 3), dtype=torch.uint8, device=self.device)
            _lib.check(L.pp_aug_resample_h(x_u8.data_ptr(), h, w, bw_d.data_ptr(), kw_d.data_ptr(), ksw, w_rs, tmp.data_ptr(), st),
                       "pp_aug_resample_h")
        else:
            tmp = x_u8
        bh_d, kh_d, ksh = self._table("bilinear", h, h_rs)
        crop = torch.empty((ch, cw, 3), dtype=torch.uint8, device=self.device)
        _lib.check(L.pp_aug_vcrop(tmp.data_ptr(), bh_d.data_ptr(), kh_d.data_ptr(), ksh, h_rs, w_rs, p["start_h"], p["start_w"], ch, cw,
                                  int(p["flip"]), self.mean_val[0], self.mean_val[1], self.mean_val[2], crop.data_ptr(), st), "pp_aug_vcrop")
        if y_out is not None or q_out is not None:
            ty = tx = qy = qx = None
            if y_out is not None:
                ty, tx = self._table("pil_nearest", h, h_rs), self._table("pil_nearest", w, w_rs)
            if q_out is not None:
                qy, qx = self._table("torch_nearest", h, h_rs), self._table("torch_nearest", w, w_rs)
            _lib.check(L.pp_aug_labels(y_u8.data_ptr() if y_out is not None else None, q_u8.data_ptr() if q_out is not None else None, w,
                                       ty.data_ptr() if ty is not None else None, tx.data_ptr() if tx is not None else None,
                                       qy.data_ptr() if qy is not None else None, qx.data_ptr() if qx is not None else None,
                                       h_rs, w_rs, p["start_h"], p["start_w"], ch, cw, int(p["flip"]), self.ignore_index,
                                       y_out.data_ptr() if y_out is not None else None, q_out.data_ptr() if q_out is not None else None, st),
                       "pp_aug_labels")
        if lq_out is not None:
            # the label-map leg of the same kernel (gather + ignore_index pad) driven by torch-nearest tables
            qy, qx = self._table("torch_nearest", h, h_rs), self._table("torch_nearest", w, w_rs)
            _lib.check(L.pp_aug_labels(lq_u8.data_ptr(), None, w, qy.data_ptr(), qx.data_ptr(), None, None, h_rs, w_rs, p["start_h"],
                                       p["start_w"], ch, cw, int(p["flip"]), self.ignore_index, lq_out.data_ptr(), None, st), "pp_aug_labels")
        n = ch * cw
        scratch = None
        for op, factor in p["ops"]:
            if op == 1 and scratch is None:
                scratch = torch.empty(1, dtype=torch.int64, device=self.device)
            _lib.check(L.pp_aug_jitter(crop.data_ptr(), n, int(op), float(factor), scratch.data_ptr() if scratch is not None else None, st),
                       "pp_aug_jitter")
        if p["blur"] is not None:
            ks, sigma = p["blur"]
            if self.blur_arithmetic == "fixed":
                # cv2's 8-bit path since 3.4.2: 8.8 fixed-point taps, integer passes (what the reference's cv2.GaussianBlur runs today)
                kern = torch.from_numpy(cv2_gaussian_kernel_q8(ks, sigma).astype(np.int16)).to(self.device, non_blocking=True)
                sbuf = torch.empty(n * 3, dtype=torch.int16, device=self.device)
                _lib.check(L.pp_aug_blur_q8(crop.data_ptr(), ch, cw, kern.data_ptr(), ks, sbuf.data_ptr(), st), "pp_aug_blur_q8")
            else:
                kern = torch.from_numpy(cv2_gaussian_kernel(ks, sigma)).to(self.device, non_blocking=True)
                fbuf = torch.empty(n * 3, dtype=torch.float32, device=self.device)
                _lib.check(L.pp_aug_blur(crop.data_ptr(), ch, cw, kern.data_ptr(), ks, fbuf.data_ptr(), st), "pp_aug_blur")
        _lib.check(L.pp_aug_to_tensor(crop.data_ptr(), n, self._mean_c, self._std_c, x_out.data_ptr(), st), "pp_aug_to_tensor")
        return crop

    # ---- a batch ---------------------------------------------------------------------------------------------------
    def __call__(self, images: List, labels: Optional[List] = None, queries: Optional[List] = None, params: Optional[List[dict]] = None,
                 labelled_queries: Optional[List] = None):
        """images: list of [H,W,3] uint8 (numpy or torch, host or device) or one stacked [B,H,W,3] tensor; labels / queries /
        labelled_queries: lists (or stacked tensors) of [H,W] uint8 / bool / integer maps with values < 256.
        -> {'x': f32 [B,3,ch,cw], 'y': int64 [B,ch,cw] | None, 'queries': uint8 [B,ch,cw] | None,
            'labelled_queries': int64 [B,ch,cw] | None, 'params': [...]}  - the keys of the reference's batch dict
        (base_dataset.py:190-196)."""
        B = len(images)
        ch, cw = self.crop_size
        x = torch.empty((B, 3, ch, cw), dtype=torch.float32, device=self.device)
        y = torch.empty((B, ch, cw), dtype=torch.int64, device=self.device) if labels is not None else None
        q = torch.empty((B, ch, cw), dtype=torch.uint8, device=self.device) if queries is not None else None
        lq = torch.empty((B, ch, cw), dtype=torch.int64, device=self.device) if labelled_queries is not None else None

        def u8(t):
            t = torch.as_tensor(t)
            t = t.view(torch.uint8) if t.dtype == torch.bool else (t if t.dtype == torch.uint8 else t.to(torch.uint8))
            return t.to(self.device, non_blocking=True).contiguous()

        def stacked(v):          # one upload for a whole batch of equal-sized maps (the collated form)
            return u8(v) if (torch.is_tensor(v) or isinstance(v, np.ndarray)) else None

        xs, ys, qs, lqs = stacked(images), stacked(labels) if labels is not None else None, \
            stacked(queries) if queries is not None else None, stacked(labelled_queries) if labelled_queries is not None else None
        used = []
        for b in range(B):
            img = xs[b] if xs is not None else u8(images[b])
            lab = (ys[b] if ys is not None else u8(labels[b])) if labels is not None else None
            qq = (qs[b] if qs is not None else u8(queries[b])) if queries is not None else None
            ll = (lqs[b] if lqs is not None else u8(labelled_queries[b])) if labelled_queries is not None else None
            p = params[b] if params is not None else self.draw(int(img.shape[0]), int(img.shape[1]))
            self.apply(img, lab, qq, p, x[b], y[b] if y is not None else None, q[b] if q is not None else None,
                       ll, lq[b] if lq is not None else None)
            used.append(p)
        return {"x": x, "y": y, "queries": q, "labelled_queries": lq, "params": used}
