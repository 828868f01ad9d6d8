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
  cv2.GaussianBlur                          -> restated from OpenCV's documented algorithm ("parity unpinned": cv2 absent)
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


def gaussian_blur(img_u8: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) on HWC uint8: separable float32 filter, BORDER_REFLECT_101, cvRound."""
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
        idx = np.arange(-half, n + half)
        if n > 1:
            while ((idx < 0) | (idx >= n)).any():
                idx = np.where(idx < 0, -idx, idx)
                idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
        else:
            idx = np.zeros_like(idx)
        ap = np.take(a, idx, axis=axis)
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
