"""CPU restatement of the reference's training-time augmentation (datasets/base_dataset.py:48-141,181) for GIVEN random
parameters, written with the very primitives the reference reaches through torchvision's PIL backend: PIL.Image.resize /
ImageOps.expand / crop / transpose, PIL.ImageEnhance, Image.convert("L"/"HSV"), and torch's nearest interpolation / pad for
the uint8 query tensor.  TEST INFRASTRUCTURE ONLY (tests/test_augment_gpu.py); never imported by pixelpick_amd/.

Parity status: the reference's datasets/ cannot be imported here (torchvision and cv2 are absent from the image), so this file
restates torchvision.transforms.functional's PIL code paths, which are one-line wrappers:
  TF.resize(pil, (h, w), BILINEAR|NEAREST) -> pil.resize((w, h), resample)          functional_pil.resize
  TF.pad(pil, (0, 0, pw, ph), fill=f)      -> ImageOps.expand(pil, (0, 0, pw, ph), fill=f)   functional_pil.pad
  TF.crop / TF.hflip                        -> pil.crop / pil.transpose(FLIP_LEFT_RIGHT)
  TF.resize(uint8 tensor, NEAREST)          -> F.interpolate(float, size, mode="nearest") -> uint8   functional_tensor.resize
  adjust_brightness/contrast/saturation     -> ImageEnhance.Brightness/Contrast/Color(img).enhance(f)
  adjust_hue                                -> HSV split, np.uint8 H += np.uint8(f * 255) (wraps), merge, convert RGB
  RandomGrayscale                           -> img.convert("L") replicated to 3 channels
  cv2.GaussianBlur                          -> OpenCV's 8-bit path restated (below).  "PARITY UNPINNED": cv2 is absent from this
                                               image, so no cv2-generated fixture exists yet; tests hold the device kernel to THIS.

cv2.GaussianBlur on CV_8U follows OpenCV 4.5 - 4.10, modules/imgproc/src/smooth.dispatch.cpp + fixedpoint.inl.hpp + smooth.simd.hpp
(the path every default x86 build takes for 8-bit input that is not a sub-matrix: `GaussianBlurFixedPoint`):
  1. getGaussianKernelBitExact: t_i = exp(-0.5 / sigma^2 * x_i^2), x_i = i - (n-1)/2, in IEEE double; the centre tap is exactly 1;
     sum = 2 * sum(t_i, i < n/2) + 1; k_i = t_i * (1 / sum).  (OpenCV evaluates this in its `softdouble` class - IEEE-754 double in
     software, so identical on every CPU - with its own exp(); numpy's double exp can differ from it in the last ulp, which the
     8-bit quantisation below absorbs unless a scaled tap sits within 1e-13 of a rounding boundary.)
  2. getGaussianKernelFixedPoint_ED: taps as ufixedpoint16 with 8 fractional bits, rounded WITH ERROR DIFFUSION from the outside
     in - v_i = cvRound(k_i * 256 + err), err = (k_i * 256 + err) - v_i (cvRound: half to even), mirrored - and the centre tap
     = 256 - 2 * sum(v_i): the taps sum to exactly 1.0, a constant image stays constant.
  3. hlineSmooth: row pass in ufixedpoint16 (8.8): sum_j v_j * src[x + j - n/2] with BORDER_REFLECT_101 (cv2's default border);
     <= 255 * 256, never saturates.  vlineSmooth: column pass of those 8.8 values in ufixedpoint32 (16.16), then
     saturate_cast<uint8_t>: (acc + 0x8000) >> 16 - round half UP.  All integer: no dependence on summation order or fma.
Before 3.4.2 / 4.0 (and for sub-matrix inputs) cv2 ran a float32 separable filter instead: kept below as gaussian_blur_float.
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageEnhance, ImageOps


def geometric(x_u8: np.ndarray, y_u8: np.ndarray, q_u8: np.ndarray, p: dict, crop_size, mean_val, ignore_index):
    """x_u8 [H,W,3], y_u8 [H,W], q_u8 [H,W] in {0,1} -> (PIL image, y int64 [ch,cw], queries uint8 [ch,cw]) (base_dataset.py:55-118)."""
    ch, cw = crop_size
    x = Image.fromarray(x_u8)
    y = Image.fromarray(y_u8)
    q = torch.from_numpy(q_u8.astype(np.uint8)) * 255
    w, h = x.size
    h_rs, w_rs = p["h_rs"], p["w_rs"]
    if (h_rs, w_rs) != (h, w):
        x = x.resize((w_rs, h_rs), Image.BILINEAR)
        y = y.resize((w_rs, h_rs), Image.NEAREST)
        q = F.interpolate(q[None, None].float(), size=(h_rs, w_rs), mode="nearest")[0, 0].to(torch.uint8)
    pad_h, pad_w = max(ch - h_rs, 0), max(cw - w_rs, 0)
    x = ImageOps.expand(x, border=(0, 0, pad_w, pad_h), fill=tuple(mean_val))
    y = ImageOps.expand(y, border=(0, 0, pad_w, pad_h), fill=ignore_index)
    q = F.pad(q, (0, pad_w, 0, pad_h), value=0)
    sh, sw = p["start_h"], p["start_w"]
    x = x.crop((sw, sh, sw + cw, sh + ch))
    y = y.crop((sw, sh, sw + cw, sh + ch))
    q = q[sh:sh + ch, sw:sw + cw]
    if p["flip"]:
        x = x.transpose(Image.FLIP_LEFT_RIGHT)
        y = y.transpose(Image.FLIP_LEFT_RIGHT)
        q = q.flip(-1)
    return x, np.asarray(y, np.int64), (np.asarray(q, dtype=np.uint8) // 255)


def jitter(img: Image.Image, op: int, factor: float) -> Image.Image:
    if op == 0:
        return ImageEnhance.Brightness(img).enhance(factor)
    if op == 1:
        return ImageEnhance.Contrast(img).enhance(factor)
    if op == 2:
        return ImageEnhance.Color(img).enhance(factor)
    if op == 3:
        h, s, v = img.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        shift = np.uint8(int(factor * 255) & 0xFF)                  # np.uint8(hue_factor * 255) of older numpy: C cast, wraps
        with np.errstate(over="ignore"):
            np_h = (np_h + shift).astype(np.uint8)
        return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
    if op == 4:
        l = np.asarray(img.convert("L"))
        return Image.fromarray(np.stack([l, l, l], axis=-1))
    raise ValueError(op)


def gaussian_kernel_q8(ksize: int, sigma: float) -> np.ndarray:
    """The ufixedpoint16 taps (8 fractional bits, int64 here) cv2.GaussianBlur uses for 8-bit images, sigma > 0, odd ksize:
    getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (see the module docstring)."""
    assert ksize % 2 == 1 and ksize >= 1 and sigma > 0
    n2 = (ksize - 1) // 2
    scale2x = np.float64(-0.125) / (np.float64(sigma) * np.float64(sigma))       # x below is 2 * (i - (n-1)/2)
    vals = []
    total = np.float64(0.0)
    x = 1 - ksize
    for _ in range(n2):
        t = np.exp(np.float64(x * x) * scale2x)
        vals.append(t)
        total = total + t
        x += 2
    total = total * np.float64(2.0) + np.float64(1.0)
    mul1 = np.float64(1.0) / total
    k = [v * mul1 for v in vals]                                                 # outer half; the centre is 1 * mul1
    out = np.zeros(ksize, dtype=np.int64)
    err = np.float64(0.0)
    acc = 0
    for i in range(n2):
        adj = k[i] * np.float64(256.0) + err
        v0 = int(np.rint(adj))                                                   # cvRound: round half to even
        err = adj - np.float64(v0)
        out[i] = out[ksize - 1 - i] = v0
        acc += v0
    out[n2] = 256 - 2 * acc
    return out


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(idx)
    while ((idx < 0) | (idx >= n)).any():
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    return idx


def gaussian_blur(img_u8: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) on HWC uint8 as OpenCV >= 3.4.2 / 4.x computes it: 8.8 fixed-point taps, exact
    integer row pass, 16.16 column pass, round half up (see the module docstring; base_dataset.py:208)."""
    kq = gaussian_kernel_q8(ksize, sigma)
    half = ksize // 2
    a = img_u8.astype(np.int64)
    H, W = a.shape[:2]
    ap = np.take(a, _reflect101(np.arange(-half, W + half), W), axis=1)
    row = np.zeros_like(a)
    for t in range(ksize):
        row += kq[t] * ap[:, t:t + W]
    assert row.max() <= 255 * 256
    rp = np.take(row, _reflect101(np.arange(-half, H + half), H), axis=0)
    col = np.zeros_like(a)
    for t in range(ksize):
        col += kq[t] * rp[t:t + H]
    return np.clip((col + 0x8000) >> 16, 0, 255).astype(np.uint8)


def gaussian_blur_float(img_u8: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """The float32 separable filter cv2 ran for 8-bit images before 3.4.2 (and still runs for sub-matrix inputs): float kernel,
    BORDER_REFLECT_101, cvRound."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    cf = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)
    s = 0.0
    for v in cf:
        s += float(v)
    k = (cf.astype(np.float64) * (1.0 / s)).astype(np.float32)
    half = ksize // 2
    a = img_u8.astype(np.float32)
    for axis in (1, 0):
        n = a.shape[axis]
        ap = np.take(a, _reflect101(np.arange(-half, n + half), n), axis=axis)
        out = np.zeros_like(a)
        for t in range(ksize):
            sl = [slice(None)] * 3
            sl[axis] = slice(t, t + n)
            out = out + k[t] * ap[tuple(sl)]
        a = out.astype(np.float32)
    return np.clip(np.rint(a), 0, 255).astype(np.uint8)


def to_tensor_normalize(img_u8: np.ndarray, mean, std) -> torch.Tensor:
    """TF.normalize(TF.to_tensor(pil), mean, std) (base_dataset.py:181)."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).float().div(255)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    return (t - m) / s
