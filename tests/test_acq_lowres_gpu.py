"""GPU parity tests of the fused low-resolution acquisition (pp_acq_lowres_score_topk /
pp_acq_lowres_score_at through the C ABI):
  * against the reference-generated golden vectors (tests/golden/acq_lowres.npz, tools/gen_golden_acq.py --lowres),
  * against the CPU oracle (oracle/acq.py lowres_score_topk) on seeded inputs,
  * bit for bit against the two-launch product path it replaces (pp_bilinear_fwd -> pp_acq_score_topk).
Index work is bit-exact; scores within 2e-5 rel / 4e-6 abs of the oracle (device vs host libm + the fma in the lerp)."""
import os

import numpy as np
import pytest
import torch

from oracle import acq as orc
from pixelpick_amd import _lib
from pixelpick_amd import acquisition as acq
from pixelpick_amd import engine as E

pytestmark = pytest.mark.gpu
STRATS = ["entropy", "least_confidence", "margin_sampling"]
DEV = "cuda:0"
RTOL, ATOL = 2e-5, 4e-6


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "acq_lowres.npz"))


def _nhwc(low_nchw: np.ndarray, pad: int = 0) -> torch.Tensor:
    """[B,C,h,w] numpy -> channels-last [B,h,w,C] device tensor (optionally a channel slice of a wider buffer)."""
    t = torch.from_numpy(np.ascontiguousarray(low_nchw.transpose(0, 2, 3, 1))).to(DEV)
    if pad:
        wide = torch.full(t.shape[:3] + (t.shape[3] + pad,), 7.0, device=DEV)
        wide[..., :t.shape[3]] = t
        return wide[..., :t.shape[3]]
    return t


def _unfused(low: torch.Tensor, size, crop, excl, st, k, align=True):
    """The product path the fused call replaces: pp_bilinear_fwd (NCHW out) -> crop view -> pp_acq_score_topk."""
    pred = E.bilinear(E.Tape(False), E.Var(low), size, align, 0.0, out_nchw=True).t
    if crop is not None:
        pred = pred[:, :, :crop[0], :crop[1]]
    return acq.score_topk(pred, excl, st, k, return_map=True)


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
@pytest.mark.parametrize("pad", [0, 5])
def test_golden(g, si, st, pad):
    low = _nhwc(g[f"s{si}_low"], pad)
    size, crop, excl = tuple(g[f"s{si}_size"]), tuple(g[f"s{si}_crop"]), g[f"s{si}_exclude"]
    _, _, m = acq.score_topk_lowres(low, size, None, st, 0, crop=crop)
    np.testing.assert_allclose(m.cpu().numpy(), g[f"s{si}_map_{st}"], rtol=RTOL, atol=ATOL)
    idx, val, omap = acq.score_topk_lowres(low, size, torch.from_numpy(excl), st, 20, crop=crop, return_map=True)
    idx, val, omap = idx.cpu().numpy(), val.cpu().numpy(), omap.cpu().numpy()
    for b in range(low.shape[0]):
        assert sorted(idx[b].tolist()) == g[f"s{si}_sel_{st}"][b].tolist()
        assert idx[b].tolist() == g[f"s{si}_order_{st}"][b].tolist()
        np.testing.assert_array_equal(val[b], omap[b].reshape(-1)[idx[b]])
    fill = 1.0 if st == "margin_sampling" else 0.0
    assert (omap[excl.astype(bool)] == fill).all()


def test_interpolated_logits_match_golden(g):
    """The lerp itself (shared with pp_bilinear_fwd) against the reference's F.interpolate output."""
    low = _nhwc(g["s1_low"])
    pred = E.bilinear(E.Tape(False), E.Var(low), tuple(g["s1_size"]), True, 0.0, out_nchw=True).t
    np.testing.assert_allclose(pred.cpu().numpy(), g["s1_pred"], rtol=1e-5, atol=2e-6)


CASES = [
    # B, C, (h,w), (H,W), crop, align, k
    (8, 19, (64, 128), (256, 512), None, True, 20),         # Cityscapes, 8-row tiles
    (1, 19, (64, 128), (256, 512), None, True, 20),         # one image: 4-row tiles
    (2, 11, (90, 120), (360, 480), None, True, 20),         # CamVid
    (3, 21, (80, 80), (320, 320), (317, 301), True, 20),    # VOC: padded to x8, cropped back (query.py:171-174,190)
    (2, 19, (23, 31), (67, 101), None, True, 7),            # ragged everything
    (2, 19, (16, 32), (16, 32), None, True, 20),            # identity size
    (2, 7, (40, 60), (20, 30), None, True, 5),              # down-sampling (patch wider than the tile)
    (2, 19, (32, 64), (128, 256), None, False, 20),         # align_corners=False arithmetic
    (1, 19, (64, 128), (256, 512), None, True, 6553),       # top_n_percent mode: radix select
    (3, 21, (40, 40), (160, 160), (157, 150), True, 1280),  # top_n_percent mode, cropped VOC shape: the scorer launch fills the selection's histogram
    (2, 11, (45, 60), (90, 120), None, False, 540),         # the same with align_corners=False arithmetic (x2)
    (2, 40, (12, 20), (48, 80), None, True, 20),            # generic C <= 64 bucket
    (2, 26, (12, 20), (48, 80), (48, 77), True, 48),        # generic C <= 32 bucket, largest fused k
    (1, 19, (200, 300), (25, 40), None, True, 20),          # 8x down-sampling: patch exceeds LDS -> global-read variant
    (2, 19, (256, 512), (1024, 2048), None, True, 20),      # Cityscapes full resolution (BASELINE configs[4])
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"B{c[0]}C{c[1]}_{c[2][0]}x{c[2][1]}to{c[3][0]}x{c[3][1]}_k{c[6]}")
@pytest.mark.parametrize("st", STRATS)
def test_fused_equals_two_launch_path_bit_for_bit(case, st):
    B, C, lo, size, crop, align, k = case
    rng = np.random.RandomState(hash((B, C, lo, size)) % 2**31)
    low = _nhwc((rng.randn(B, C, *lo) * 3).astype(np.float32), pad=3 if C == 19 else 0)
    hc, wc = size if crop is None else crop
    excl = rng.rand(B, hc, wc) < 0.05
    excl_t = torch.from_numpy(excl)
    i1, v1, m1 = acq.score_topk_lowres(low, size, excl_t, st, k, crop=crop, align_corners=align, return_map=True)
    i0, v0, m0 = _unfused(low, size, crop, excl_t, st, k, align)
    assert torch.equal(m1, m0)
    assert torch.equal(i1, i0)
    assert torch.equal(v1, v0)
    # without the map (the production call) the indices are the same
    i2, v2, _ = acq.score_topk_lowres(low, size, excl_t, st, k, crop=crop, align_corners=align)
    assert torch.equal(i2, i0) and torch.equal(v2, v0)


@pytest.mark.parametrize("st", STRATS)
@pytest.mark.parametrize("shape", [(2, 19, 16, 32, 64, 128), (2, 11, 12, 15, 45, 60), (1, 21, 10, 10, 40, 40)])
def test_vs_oracle(st, shape):
    B, C, h, w, H, W = shape
    rng = np.random.RandomState(5 + C)
    low = (rng.randn(B, C, h, w) * 3).astype(np.float32)
    excl = (rng.rand(B, H, W) < 0.05).astype(np.uint8)
    oi, ov, om = orc.lowres_score_topk(low, (H, W), excl, st, 20, want_map=True)
    idx, val, m = acq.score_topk_lowres(_nhwc(low), (H, W), torch.from_numpy(excl), st, 20, return_map=True)
    np.testing.assert_allclose(m.cpu().numpy(), om, rtol=RTOL, atol=ATOL)
    # index equality wherever the oracle's own k-th/(k+1)-th gap is not within rounding noise
    for b in range(B):
        srt = np.sort(om[b].reshape(-1))
        srt = srt[::-1] if st != "margin_sampling" else srt
        if np.min(np.abs(np.diff(srt[:22]))) > 1e-4 * max(1e-3, abs(float(srt[20]))):
            assert idx[b].cpu().numpy().tolist() == oi[b].tolist()


@pytest.mark.parametrize("st", STRATS)
def test_score_at_picked_pixels(st):
    rng = np.random.RandomState(11)
    B, C, size, crop = 3, 19, (128, 256), (125, 250)
    low = _nhwc((rng.randn(B, C, 32, 64) * 3).astype(np.float32), pad=1)
    _, _, m = acq.score_topk_lowres(low, size, None, st, 0, crop=crop)
    n = 500
    img = rng.randint(0, B, n)
    pix = rng.randint(0, crop[0] * crop[1], n)
    out = acq.score_at_lowres(low, size, img, pix, st, crop=crop)
    ref = m.reshape(B, -1)[torch.from_numpy(img).to(DEV), torch.from_numpy(pix).to(DEV)]
    assert torch.equal(out, ref)
    assert acq.score_at_lowres(low, size, [], [], st, crop=crop).numel() == 0


def test_reference_order_scorer_agrees():
    rng = np.random.RandomState(3)
    low = _nhwc((rng.randn(2, 19, 16, 32) * 3).astype(np.float32))
    try:
        _lib.lib().pp_debug_set_exact_formula(1)
        i1, _, m1 = acq.score_topk_lowres(low, (64, 128), None, "entropy", 20, return_map=True)
    finally:
        _lib.lib().pp_debug_set_exact_formula(0)
    i0, _, m0 = acq.score_topk_lowres(low, (64, 128), None, "entropy", 20, return_map=True)
    np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=RTOL, atol=ATOL)


def test_full_size_properties():
    """BASELINE shape (B=16 x 19 x 64x128 -> 256x512): indices in range, distinct, never excluded, value-sorted, and the
    values are the map's k largest."""
    torch.manual_seed(0)
    B, k = 16, 20
    low = torch.randn(B, 64, 128, 19, device=DEV) * 3
    excl = torch.rand(B, 256, 512, device=DEV) < 0.05
    idx, val, m = acq.score_topk_lowres(low, (256, 512), excl.cpu(), "entropy", k, return_map=True)
    flat = m.reshape(B, -1)
    assert int(idx.min()) >= 0 and int(idx.max()) < 256 * 512
    for b in range(B):
        ib = idx[b].long()
        assert ib.unique().numel() == k
        assert not excl[b].reshape(-1)[ib].any()
        assert torch.equal(val[b], flat[b][ib])
        assert (val[b][:-1] >= val[b][1:]).all()
        assert torch.equal(torch.sort(val[b], descending=True).values, torch.topk(flat[b], k).values)


def test_errors():
    low = torch.randn(1, 8, 8, 19, device=DEV)
    with pytest.raises(ValueError):                      # crop larger than the interpolated size
        acq.score_topk_lowres(low, (32, 32), None, "entropy", 5, crop=(33, 32))
    with pytest.raises(ValueError):                      # k > crop_h * crop_w
        acq.score_topk_lowres(low, (4, 4), None, "entropy", 17)
    with pytest.raises(ValueError):                      # not channels-last dense
        acq.score_topk_lowres(low.permute(0, 3, 1, 2), (32, 32), None, "entropy", 5)
    with pytest.raises(_lib.PixelPickHipError):          # no CPU fallback
        acq.score_topk_lowres(low.cpu(), (32, 32), None, "entropy", 5)
