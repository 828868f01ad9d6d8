"""GPU parity tests of the STREAMED scorers (csrc/acq.hip acq_stream_kernel / acq_lowres_stream_kernel): class counts beyond the
register-resident kernels' 64.  The reference softmaxes whatever width the model emits (query.py:190) and its samplers reduce over
dim 1 of any size (query.py:229-239), so no class count may be rejected.
  * forced onto C <= 64 inputs (pp_debug_set_acq_tuning(10, 0)) the streamed kernels are BIT-EQUAL to the register kernels:
    maps, picks and values, every layout, default and reference-order scorer, from-probability entry, low-resolution entry;
  * at C = 100 / 150 they are checked against the CPU oracle (oracle/acq.py: reference operation order, host libm): maps within the
    score tolerance, picks identical on gap-guarded inputs."""
import numpy as np
import pytest
import torch

from oracle import acq as orc
from pixelpick_amd import _lib
from pixelpick_amd import acquisition as acq

from test_acq_gpu import _guarded_full_size_case, _layouts
from test_acq_lowres_gpu import _nhwc

pytestmark = pytest.mark.gpu
STRATS = ["entropy", "least_confidence", "margin_sampling"]
DEV = "cuda:0"
RTOL, ATOL = 2e-5, 4e-6


class _streamed:
    def __enter__(self):
        _lib.lib().pp_debug_set_acq_tuning(10, 0)

    def __exit__(self, *exc):
        _lib.lib().pp_debug_set_acq_tuning(0, 0)


@pytest.mark.parametrize("exact", [0, 1], ids=["default-scorer", "reference-order-scorer"])
@pytest.mark.parametrize("C", [19, 21, 40])
@pytest.mark.parametrize("st", STRATS)
def test_streamed_scorer_is_bit_equal_to_the_register_scorer(st, C, exact):
    rng = np.random.RandomState(C * 7 + exact)
    B, H, W = 3, 36, 52
    logits = (rng.randn(B, C, H, W) * 3).astype(np.float32)
    logits[0, :, 3, 5] = 0.0
    logits[0, 0, 3, 5] = 120.0                                  # smallest probability underflows: the 0 * log 0 = NaN branch
    excl = torch.from_numpy(rng.rand(B, H, W) < 0.05)
    _lib.lib().pp_debug_set_exact_formula(exact)
    try:
        for name, t in _layouts(logits):
            for k in (20, 300):                                  # fused per-block extraction / map + radix select
                i0, v0, m0 = acq.score_topk(t, excl, st, k, return_map=True)
                with _streamed():
                    i1, v1, m1 = acq.score_topk(t, excl, st, k, return_map=True)
                    m2 = acq.score_map(t, excl, st)
                assert torch.equal(m1.view(torch.int32), m0.view(torch.int32)), (name, k)
                assert torch.equal(m2.view(torch.int32), m0.view(torch.int32)), (name, k)
                assert torch.equal(i1, i0) and torch.equal(v1.view(torch.int32), v0.view(torch.int32)), (name, k)
    finally:
        _lib.lib().pp_debug_set_exact_formula(0)


@pytest.mark.parametrize("st", STRATS)
def test_streamed_from_probability_entry_is_bit_equal(st):
    torch.manual_seed(4)
    prob = torch.softmax(torch.randn(2, 19, 24, 40, device=DEV) * 3, dim=1)
    for t in (prob, prob.contiguous(memory_format=torch.channels_last)):
        m0 = acq.uncertainty_from_prob(t, st)
        with _streamed():
            m1 = acq.uncertainty_from_prob(t, st)
        assert torch.equal(m1.view(torch.int32), m0.view(torch.int32))


@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("st", STRATS)
def test_streamed_lowres_entry_is_bit_equal(st, align, exact):
    rng = np.random.RandomState(17)
    B, C, lo, size, crop = 2, 19, (16, 32), (64, 128) if align else (32, 64), None
    low = _nhwc((rng.randn(B, C, *lo) * 3).astype(np.float32), pad=3)
    excl = torch.from_numpy(rng.rand(B, *size) < 0.05)
    _lib.lib().pp_debug_set_exact_formula(exact)
    try:
        i0, v0, m0 = acq.score_topk_lowres(low, size, excl, st, 20, crop=crop, align_corners=align, return_map=True)
        with _streamed():
            i1, v1, m1 = acq.score_topk_lowres(low, size, excl, st, 20, crop=crop, align_corners=align, return_map=True)
        assert torch.equal(m1.view(torch.int32), m0.view(torch.int32))
        assert torch.equal(i1, i0) and torch.equal(v1, v0)
        if not exact:
            img = rng.randint(0, B, 200)
            pix = rng.randint(0, size[0] * size[1], 200)
            a0 = acq.score_at_lowres(low, size, img, pix, st, align_corners=align)
            with _streamed():
                a1 = acq.score_at_lowres(low, size, img, pix, st, align_corners=align)
            assert torch.equal(a1.view(torch.int32), a0.view(torch.int32))
    finally:
        _lib.lib().pp_debug_set_exact_formula(0)


@pytest.mark.parametrize("C", [65, 100, 150])
@pytest.mark.parametrize("st", STRATS)
def test_wide_heads_match_the_oracle(st, C):
    """query.py:190,229-239 at class counts the register kernels do not take: maps against the oracle, every layout."""
    rng = np.random.RandomState(C)
    B, H, W = 2, 20, 28
    logits = (rng.randn(B, C, H, W) * 3).astype(np.float32)
    excl = (rng.rand(B, H, W) < 0.05).astype(np.uint8)
    ref = orc.apply_exclude(orc.score_map(logits, st), excl, st)
    for exact in (0, 1):
        _lib.lib().pp_debug_set_exact_formula(exact)
        try:
            for name, t in _layouts(logits):
                m = acq.score_map(t, torch.from_numpy(excl), st).cpu().numpy()
                np.testing.assert_allclose(m, ref, rtol=RTOL, atol=ATOL, err_msg=f"{name} exact={exact}")
                idx, val, omap = acq.score_topk(t, torch.from_numpy(excl), st, 20, return_map=True)
                np.testing.assert_array_equal(omap.cpu().numpy(), m)
                for b in range(B):                                      # picks = the value-sorted top-k of the device's own map
                    key = omap[b].reshape(-1).cpu().numpy().astype(np.float64)
                    order = np.lexsort((np.arange(key.size), -key if st != "margin_sampling" else key))[:20]
                    assert idx[b].cpu().numpy().tolist() == order.tolist(), name
        finally:
            _lib.lib().pp_debug_set_exact_formula(0)
    prob = torch.softmax(torch.from_numpy(logits).to(DEV), dim=1)
    np.testing.assert_allclose(acq.uncertainty_from_prob(prob, st).cpu().numpy(), orc.score_map(logits, st), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("exact", [0, 1], ids=["default-scorer", "reference-order-scorer"])
@pytest.mark.parametrize("C,H,W,st", [(150, 128, 256, "entropy"), (150, 128, 256, "least_confidence"), (100, 96, 160, "margin_sampling")])
def test_wide_head_picks_equal_the_oracles_on_gap_guarded_input(C, H, W, st, exact):
    # planted class vectors (a_j, 0, ..., 0): at C = 150 the entropy / least-confidence steps of the C = 19 cases fall below the 1e-3 guard
    steps = (3.0, 0.15, 0.05) if st != "margin_sampling" else None
    logits, excl, o_idx, o_val = _guarded_full_size_case(C, H, W, st, seed=H + C + len(st), steps=steps)
    _lib.lib().pp_debug_set_exact_formula(exact)
    try:
        for name, t in (("nchw", torch.from_numpy(logits).to(DEV)),
                        ("nhwc", torch.from_numpy(logits).to(DEV).contiguous(memory_format=torch.channels_last))):
            idx, val, _ = acq.score_topk(t, torch.from_numpy(excl), st, 20)
            assert idx[0].cpu().numpy().tolist() == o_idx[0].tolist(), name
            np.testing.assert_allclose(val[0].cpu().numpy(), o_val[0], rtol=RTOL, atol=ATOL)
    finally:
        _lib.lib().pp_debug_set_exact_formula(0)


@pytest.mark.parametrize("st", STRATS)
def test_wide_head_lowres_entry_matches_the_oracle(st):
    B, C, h, w, H, W = 2, 100, 12, 20, 48, 80
    rng = np.random.RandomState(31)
    low = (rng.randn(B, C, h, w) * 3).astype(np.float32)
    excl = (rng.rand(B, H, W) < 0.05).astype(np.uint8)
    oi, ov, om = orc.lowres_score_topk(low, (H, W), excl, st, 20, want_map=True)
    idx, val, m = acq.score_topk_lowres(_nhwc(low), (H, W), torch.from_numpy(excl), st, 20, return_map=True)
    np.testing.assert_allclose(m.cpu().numpy(), om, rtol=RTOL, atol=ATOL)
    for b in range(B):
        srt = np.sort(om[b].reshape(-1))
        srt = srt[::-1] if st != "margin_sampling" else srt
        if np.min(np.abs(np.diff(srt[:22]))) > 1e-4 * max(1e-3, abs(float(srt[20]))):
            assert idx[b].cpu().numpy().tolist() == oi[b].tolist()
    pix = idx.reshape(-1).cpu().numpy()
    img = np.repeat(np.arange(B), 20)
    at = acq.score_at_lowres(_nhwc(low), (H, W), img, pix, st)
    raw = orc.lowres_score_topk(low, (H, W), None, st, 20, want_map=True)[2]
    np.testing.assert_allclose(at.cpu().numpy(), raw.reshape(B, -1)[img, pix], rtol=RTOL, atol=ATOL)


def test_mc_accumulate_wide_head():
    """pp_acq_softmax_sum beyond 64 classes: per-class sums in chunks of 64 (query.py:181-187)."""
    torch.manual_seed(2)
    T, C, H, W = 4, 150, 9, 13
    logits = torch.randn(T, C, H, W, device=DEV) * 3
    prob = torch.softmax(logits, dim=1)
    for st, ref_uc in (("entropy", (-prob * prob.log()).sum(1)), ("least_confidence", 1 - prob.max(1)[0]),
                       ("margin_sampling", (prob.topk(2, dim=1).values[:, 0] - prob.topk(2, dim=1).values[:, 1]).abs())):
        p_out = torch.empty(C, H, W, device=DEV)
        u_out = torch.empty(H, W, device=DEV)
        acq.mc_accumulate_(logits[:1], p_out, u_out, st, 1.0 / T, accumulate=False)
        acq.mc_accumulate_(logits[1:], p_out, u_out, st, 1.0 / T, accumulate=True)
        torch.testing.assert_close(p_out, prob.mean(0), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(u_out, ref_uc.mean(0), rtol=1e-5, atol=1e-6)
