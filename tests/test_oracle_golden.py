"""Pins the CPU oracle (oracle/acq_oracle.c) against golden vectors generated from the imported
reference (tools/gen_golden_acq.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import acq as orc

STRATS = ["entropy", "least_confidence", "margin_sampling"]
FILL = {"entropy": 0.0, "least_confidence": 0.0, "margin_sampling": 1.0}


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "acq_scores_topk.npz"))


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_score_maps_match_reference(g, si, st):
    logits = g[f"s{si}_logits"]
    ref = g[f"s{si}_map_{st}"]
    got = orc.score_map(logits, st)
    # float32 exp/log implementations differ by ulps between libm and torch's vectorised kernels
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_nhwc_strides_give_same_map(g, si, st):
    logits = g[f"s{si}_logits"]
    nhwc = np.ascontiguousarray(logits.transpose(0, 2, 3, 1)).transpose(0, 3, 1, 2)  # NCHW view of NHWC storage
    assert not nhwc.flags["C_CONTIGUOUS"]
    np.testing.assert_array_equal(orc.score_map(nhwc, st), orc.score_map(logits, st))


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_topk_index_sets_and_order_bit_exact(g, si, st):
    logits, excl = g[f"s{si}_logits"], g[f"s{si}_exclude"]
    idx, val = orc.score_topk(logits, excl, st, 20)
    for b in range(logits.shape[0]):
        assert sorted(idx[b].tolist()) == g[f"s{si}_sel_{st}"][b].tolist()
        # gap-guarded fixtures: the value-sorted order is pinned as well
        assert idx[b].tolist() == g[f"s{si}_order_{st}"][b].tolist()


@pytest.mark.parametrize("st", STRATS)
def test_select_modes_top_percent_order(golden_dir, st):
    m = np.load(os.path.join(golden_dir, "acq_select_modes.npz"))
    uc = m[f"{st}_uc"]
    k = int(uc.size * 0.05)
    idx, _ = orc.topk(uc, k, largest=st != "margin_sampling")
    assert idx.tolist() == m[f"{st}_top5_order"].tolist()
    # subsample exactly like query.py:63-64
    np.random.seed(int(m["np_seed_top5"]))
    sub = np.random.choice(idx.astype(np.int64), 10, False)
    assert sorted(sub.tolist()) == m[f"{st}_top5_sel"].tolist()


def test_edges_nan_and_few(golden_dir):
    e = np.load(os.path.join(golden_dir, "acq_edges.npz"))
    ent = orc.score_map(e["nan_logits"], "entropy")[0]
    ref = e["nan_entropy_map"]
    assert np.array_equal(np.isnan(ent), np.isnan(ref))
    np.testing.assert_allclose(ent[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(orc.score_map(e["nan_logits"], "least_confidence")[0], e["nan_lc_map"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(orc.score_map(e["nan_logits"], "margin_sampling")[0], e["nan_margin_map"], rtol=2e-5, atol=2e-6)
    idx, val = orc.topk(ent, 4, True)
    # NaN first (both NaN pixels, lower index first under our tiebreak); then the two largest finite
    assert np.isnan(val[:2]).all() and not np.isnan(val[2:]).any()
    assert set(idx[:2].tolist()) == set(e["nan_top4_idx"][:2].tolist())
    assert idx[2:].tolist() == e["nan_top4_idx"][2:].tolist()
    assert idx[0] < idx[1]
    # k == #free and k > #free
    uc = e["few_uc"]
    idx5, _ = orc.topk(uc, 5, True)
    assert sorted(idx5.tolist()) == e["few_sel_k5"].tolist()
    idx8, _ = orc.topk(uc, 8, True)
    assert len(set(idx8.tolist())) == 8 == int(e["few_sel_k8_count"])
    assert set(e["few_sel_k5"].tolist()) <= set(idx8.tolist())
    # excluded extras: lowest flat index first (fixed tiebreak)
    extras = [i for i in idx8.tolist() if i not in set(e["few_sel_k5"].tolist())]
    excl_idx = np.flatnonzero(e["few_exclude"].reshape(-1))
    assert extras == excl_idx[:3].tolist()


def test_tiebreak_and_signed_zero():
    s = np.array([1.0, 2.0, 2.0, -0.0, 0.0, 2.0, np.nan], dtype=np.float32)
    idx, _ = orc.topk(s, 7, True)
    assert idx.tolist() == [6, 1, 2, 5, 0, 3, 4]
    idx, _ = orc.topk(s, 7, False)
    assert idx.tolist() == [3, 4, 0, 1, 2, 5, 6]


def test_oracle_reproduces_voc_pad_and_human_label_branches(golden_dir):
    """query.py:171-174,190 and :145-146,196-197 — the oracle (score + exclude + top-k) on logits made the way the
    reference makes them (reflect pad -> net -> crop) lands on the reference QuerySelector's own coordinates."""
    import torch
    import torch.nn.functional as F
    gb = np.load(os.path.join(golden_dir, "acq_branches.npz"))
    W, b = torch.from_numpy(gb["voc_W"]), torch.from_numpy(gb["voc_b"])
    for i in range(2):
        x = torch.from_numpy(gb["voc_xs"][i:i + 1])
        xp = F.pad(x, (0, 56 - 53, 0, 40 - 37), mode="reflect")
        logits = F.conv2d(xp, W, b, padding=1)[:, :, :37, :53].contiguous().numpy()
        excl = (gb["voc_prev"][i] | (gb["voc_ys"][i] == 255)).astype(np.uint8)[None]
        idx, _ = orc.score_topk(logits, excl, "margin_sampling", 10)
        q = np.zeros(37 * 53, bool)
        q[idx[0]] = True
        ys, xs = np.where(q.reshape(37, 53))
        np.testing.assert_array_equal(xs, gb[f"voc_x_{i}"])
        np.testing.assert_array_equal(ys, gb[f"voc_y_{i}"])
    W, b = torch.from_numpy(gb["hl_W"]), torch.from_numpy(gb["hl_b"])
    for i in range(2):
        logits = F.conv2d(torch.from_numpy(gb["hl_xs"][i:i + 1]), W, b).numpy()
        excl = (gb["hl_labelled"][i] != 11).astype(np.uint8)[None]
        idx, _ = orc.score_topk(logits, excl, "least_confidence", 12)
        q = np.zeros(24 * 40, bool)
        q[idx[0]] = True
        ys, xs = np.where(q.reshape(24, 40))
        np.testing.assert_array_equal(xs, gb[f"hl_x_{i}"])
        np.testing.assert_array_equal(ys, gb[f"hl_y_{i}"])


# ---- low-resolution logits -> interpolate -> crop -> score -> top-k ---------------------------
@pytest.fixture(scope="module")
def glow(golden_dir):
    return np.load(os.path.join(golden_dir, "acq_lowres.npz"))


def test_bilinear_restatement_matches_torch_and_golden(glow):
    import torch
    import torch.nn.functional as F
    np.testing.assert_allclose(orc.bilinear_resize(glow["s1_low"], tuple(glow["s1_size"])), glow["s1_pred"], rtol=1e-5, atol=2e-6)
    x = np.random.RandomState(0).randn(2, 5, 7, 9).astype(np.float32)
    for ac in (True, False):
        for size in [(28, 36), (13, 20), (7, 9), (3, 4), (1, 1), (40, 5)]:
            ref = F.interpolate(torch.from_numpy(x), size=size, mode="bilinear", align_corners=ac).numpy()
            np.testing.assert_allclose(orc.bilinear_resize(x, size, ac), ref, rtol=1e-5, atol=2e-6, err_msg=f"{ac} {size}")


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_lowres_path_matches_reference(glow, si, st):
    low, excl = glow[f"s{si}_low"], glow[f"s{si}_exclude"]
    size, crop = tuple(glow[f"s{si}_size"]), tuple(glow[f"s{si}_crop"])
    idx, val, m = orc.lowres_score_topk(low, size, excl, st, 20, crop=crop, want_map=True)
    _, _, raw = orc.lowres_score_topk(low, size, None, st, 1, crop=crop, want_map=True)
    np.testing.assert_allclose(raw, glow[f"s{si}_map_{st}"], rtol=2e-5, atol=4e-6)
    for b in range(low.shape[0]):
        assert sorted(idx[b].tolist()) == glow[f"s{si}_sel_{st}"][b].tolist()
        assert idx[b].tolist() == glow[f"s{si}_order_{st}"][b].tolist()


# ---- cv2.GaussianBlur (datasets/base_dataset.py:192-208): closes "parity unpinned" the moment a cv2-written fixture exists ------
def _cv2_blur_fixture(golden_dir):
    p = os.path.join(golden_dir, "aug_blur_cv2.npz")
    if not os.path.exists(p):
        pytest.skip("no cv2-written fixture (tools/gen_golden_blur_cv2.py needs a box with opencv; this image has none): the blur's "
                    "parity with the real library stays unpinned")
    return np.load(p)


def test_blur_oracle_equals_cv2_fixture(golden_dir):
    """oracle/augment.py:gaussian_blur (OpenCV's 8-bit fixed-point path restated) against cv2.GaussianBlur's own outputs: every
    pixel of every (ksize, sigma) case.  Skips only while the fixture file does not exist; a mismatch is a failure."""
    import sys
    g = _cv2_blur_fixture(golden_dir)
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "tools"))
    from gen_golden_blur_cv2 import image
    from oracle import augment as aug
    for ks, sg, seed, ref in zip(g["ksize"], g["sigma"], g["seed"], g["blurred"]):
        got = aug.gaussian_blur(image(int(seed)), int(ks), float(sg))
        assert np.array_equal(got, ref), (int(ks), float(sg), str(g["cv2_version"]), int(np.abs(got.astype(int) - ref).max()))
