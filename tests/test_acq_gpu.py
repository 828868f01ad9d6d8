"""GPU parity tests: the HIP acquisition path (through the C ABI) vs the CPU oracle and the
reference-generated golden vectors.  Index work must be bit-exact; scores within 2e-5 rel / 2e-6 abs
(float32 exp/log ulp differences between device libm and host libm)."""
import os
import tempfile
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import acq as orc
from pixelpick_amd import _lib
from pixelpick_amd import acquisition as acq
from pixelpick_amd import query as ppq

pytestmark = pytest.mark.gpu
STRATS = ["entropy", "least_confidence", "margin_sampling"]
DEV = "cuda:0"
RTOL, ATOL = 2e-5, 2e-6


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "acq_scores_topk.npz"))


@pytest.fixture(params=[0, 1, 2], ids=["prefilter-dpp", "prefilter-shfl", "plain-loop"])
def reduce_mode(request):
    _lib.lib().pp_debug_set_reduce_mode(request.param)
    yield request.param
    _lib.lib().pp_debug_set_reduce_mode(0)


def _layouts(logits_np):
    t = torch.from_numpy(logits_np).to(DEV)
    yield "nchw", t
    yield "nhwc", t.contiguous(memory_format=torch.channels_last)
    B, C, H, W = t.shape
    big = torch.zeros(B, C, H + 3, W + 5, device=DEV)
    big[:, :, :H, :W] = t
    yield "cropped", big[:, :, :H, :W]


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_golden_maps_and_topk(g, si, st, reduce_mode):
    logits, excl = g[f"s{si}_logits"], g[f"s{si}_exclude"]
    ref_map = g[f"s{si}_map_{st}"]
    for name, t in _layouts(logits):
        m = acq.score_map(t, None, st).cpu().numpy()
        np.testing.assert_allclose(m, ref_map, rtol=RTOL, atol=ATOL, err_msg=name)
        idx, val, omap = acq.score_topk(t, torch.from_numpy(excl), st, 20, return_map=True)
        idx, val, omap = idx.cpu().numpy(), val.cpu().numpy(), omap.cpu().numpy()
        for b in range(logits.shape[0]):
            assert sorted(idx[b].tolist()) == g[f"s{si}_sel_{st}"][b].tolist(), name
            assert idx[b].tolist() == g[f"s{si}_order_{st}"][b].tolist(), name
            np.testing.assert_array_equal(val[b], omap[b].reshape(-1)[idx[b]])
        fill = 1.0 if st == "margin_sampling" else 0.0
        assert (omap[excl.astype(bool)] == fill).all()


@pytest.mark.parametrize("st", STRATS)
@pytest.mark.parametrize("shape", [(3, 19, 64, 128), (2, 11, 45, 60), (2, 21, 33, 47), (1, 7, 16, 40), (1, 40, 24, 36)])
def test_vs_oracle_random(st, shape, reduce_mode):
    rng = np.random.RandomState(hash((st, shape)) % 2**31)
    B, C, H, W = shape
    logits = (rng.randn(*shape) * 3).astype(np.float32)
    excl = (rng.rand(B, H, W) < 0.07).astype(np.uint8)
    k = 20
    o_idx, o_val, o_map = orc.score_topk(logits, excl, st, k, want_map=True)
    idx, val, omap = acq.score_topk(torch.from_numpy(logits).to(DEV), torch.from_numpy(excl), st, k, return_map=True)
    np.testing.assert_allclose(omap.cpu().numpy(), o_map, rtol=RTOL, atol=ATOL)
    # index parity is checked self-consistently against the oracle's top-k of the DEVICE map (ulps in
    # exp/log may legitimately swap near-ties on unguarded random data) ...
    dmap = omap.cpu().numpy()
    for b in range(B):
        e_idx, e_val = orc.topk(dmap[b], k, st != "margin_sampling")
        assert idx[b].cpu().numpy().tolist() == e_idx.tolist()
        np.testing.assert_array_equal(val[b].cpu().numpy(), e_val)
    # ... and the flip rate against the oracle's own selection stays tiny
    flips = sum(len(set(idx[b].cpu().numpy().tolist()) ^ set(o_idx[b].tolist())) for b in range(B))
    assert flips <= 2 * B


def test_void_regions_force_the_tie_fallback(reduce_mode):
    """Whole waves of excluded pixels (Cityscapes ego-vehicle / border void areas) make every key in
    the wave equal: the prefilter must fall back to the exact loop and still honour the tiebreak."""
    rng = np.random.RandomState(11)
    logits = (rng.randn(2, 19, 64, 512) * 3).astype(np.float32)
    excl = np.zeros((2, 64, 512), dtype=np.uint8)
    excl[0, :40] = 1            # 20480 consecutive excluded pixels
    excl[1] = 1
    excl[1, 63, 500:] = 0       # only 12 free pixels < k
    for st in STRATS:
        o_idx, _ = orc.score_topk(logits, excl, st, 20)
        idx, val, omap = acq.score_topk(torch.from_numpy(logits).to(DEV), torch.from_numpy(excl), st, 20, return_map=True)
        dmap = omap.cpu().numpy()
        for b in range(2):
            e_idx, _ = orc.topk(dmap[b], 20, st != "margin_sampling")
            assert idx[b].cpu().numpy().tolist() == e_idx.tolist()
        assert sorted(idx[1].cpu().numpy().tolist())[:8] == list(range(8))  # excluded extras: lowest index first


@pytest.mark.parametrize("largest", [True, False])
@pytest.mark.parametrize("B,N,k", [(1, 1000, 1), (3, 4096, 20), (2, 70001, 64), (2, 5000, 65), (1, 131072, 6553),
                                   (2, 9000, 9000), (1, 40000, 20000), (1, 300, 300)])
def test_topk_select_vs_oracle(B, N, k, largest, reduce_mode):
    rng = np.random.RandomState(N + k)
    s = rng.randn(B, N).astype(np.float32)
    s[:, ::7] = 0.5            # many ties
    s[0, 3] = np.nan
    s[0, 11] = -0.0
    idx, val = acq.topk_select(torch.from_numpy(s).to(DEV), k, largest)
    for b in range(B):
        e_idx, e_val = orc.topk(s[b], k, largest)
        assert idx[b].cpu().numpy().tolist() == e_idx.tolist()
        np.testing.assert_array_equal(val[b].cpu().numpy(), e_val)


@pytest.mark.parametrize("largest", [True, False])
@pytest.mark.parametrize("B,N,k,levels", [(3, 131072, 6553, 10), (2, 50000, 2500, 3), (5, 20000, 100, 1), (260, 8192, 409, 40)])
def test_large_k_select_with_ties_at_the_threshold(B, N, k, levels, largest):
    """The multi-block radix select (three histogram passes + one sorting block per image) on maps quantised to a few
    levels, so that the k-th value sits inside a big group of equal keys: the lowest-index members of that group must be
    taken, exactly as the oracle's stable sort does (void regions produce exactly this in the top-5 % mode)."""
    rng = np.random.RandomState(B * 7 + levels)
    s = (np.floor(rng.rand(B, N) * levels) / levels).astype(np.float32)
    s[0, :50] = np.nan if B > 1 else 0.0
    idx, val = acq.topk_select(torch.from_numpy(s).to(DEV), k, largest)
    for b in range(min(B, 6)):
        e_idx, e_val = orc.topk(s[b], k, largest)
        assert idx[b].cpu().numpy().tolist() == e_idx.tolist()
        np.testing.assert_array_equal(val[b].cpu().numpy(), e_val)
    # the one-block-per-image kernel gives the same answer (A/B switch)
    from pixelpick_amd import _lib
    _lib.lib().pp_debug_set_reduce_mode(256)
    try:
        idx2, val2 = acq.topk_select(torch.from_numpy(s).to(DEV), k, largest)
    finally:
        _lib.lib().pp_debug_set_reduce_mode(0)
    assert torch.equal(idx, idx2) and torch.equal(val.view(torch.int32), val2.view(torch.int32))


@pytest.mark.parametrize("largest", [True, False])
def test_generic_select_over_the_images_own_range_equals_the_radix_select(largest):
    """pp_topk_select on maps whose value range nobody states (the random strategy's torch.rand maps, query.py:216-221; MC-dropout mean
    scores, query.py:181-187): one min / max pass, then the quantised histogram select over every image's own finite range - three
    passes over the map instead of the radix select's five.  Same picks and values as the radix select (pp_debug_set_reduce_mode bit 22)
    and as the oracle's stable sort: negative and huge ranges, infinities, NaN, a constant image, a two-valued image, a range of a few ulps."""
    from pixelpick_amd import _lib
    L = _lib.lib()
    rng = np.random.RandomState(5)
    N, k = 131072, 6553
    s = np.empty((8, N), dtype=np.float32)
    s[0] = rng.rand(N)                                     # the random strategy
    s[1] = rng.randn(N) * 1e6 - 3e6                        # large negative range
    s[2] = rng.randn(N) * 1e-30                            # tiny magnitudes
    s[3] = rng.rand(N); s[3, ::1000] = np.inf; s[3, 5::1000] = -np.inf; s[3, 7::5000] = np.nan
    s[4] = 0.25                                            # constant -> one bin -> exact fallback
    s[5] = (rng.rand(N) < 0.5).astype(np.float32)          # two values
    s[6] = np.float32(1.0) + rng.randint(0, 4, N).astype(np.float32) * np.float32(2.0 ** -23)      # four neighbouring floats
    s[7] = np.log(19.0) * rng.beta(0.5, 3.0, N)            # an entropy-like map
    t = torch.from_numpy(s).to(DEV)
    for kk in (k, 100, 7000):
        try:
            L.pp_debug_set_reduce_mode(1 << 22)
            ref = acq.topk_select(t, kk, largest)
        finally:
            L.pp_debug_set_reduce_mode(0)
        got = acq.topk_select(t, kk, largest)
        assert torch.equal(ref[0], got[0]), kk
        assert torch.equal(torch.nan_to_num(ref[1], nan=-7.0), torch.nan_to_num(got[1], nan=-7.0)), kk
    for b in range(8):
        e_idx, _ = orc.topk(s[b], k, largest)
        assert acq.topk_select(t[b:b + 1], k, largest)[0][0].cpu().numpy().tolist() == e_idx.tolist(), b


@pytest.mark.parametrize("st", STRATS)
def test_select_modes_golden(golden_dir, st):
    m = np.load(os.path.join(golden_dir, "acq_select_modes.npz"))
    uc = torch.from_numpy(m[f"{st}_uc"])
    args = _args(query_strategy=st, top_n_percent=0.05, n_pixels_by_us=10)
    qs = ppq.QuerySelector(args, None, device=torch.device(DEV))
    np.random.seed(int(m["np_seed_top5"]))
    q = qs._select_queries(uc)
    assert np.flatnonzero(q.reshape(-1)).tolist() == m[f"{st}_top5_sel"].tolist()
    args = _args(query_strategy=st, top_n_percent=0.05, n_pixels_by_us=10, reverse_order=True)
    qs = ppq.QuerySelector(args, None, device=torch.device(DEV))
    np.random.seed(int(m["np_seed_rev"]))
    q = qs._select_queries(uc)
    assert np.flatnonzero(q.reshape(-1)).tolist() == m[f"{st}_rev_sel"].tolist()
    k = int(uc.numel() * 0.05)
    idx, _ = acq.topk_select(uc.reshape(1, -1).to(DEV), k, st != "margin_sampling")
    assert idx[0].cpu().numpy().tolist() == m[f"{st}_top5_order"].tolist()


@pytest.fixture(params=[0, 1], ids=["default-scorer", "reference-order-scorer"])
def exact_formula(request):
    _lib.lib().pp_debug_set_exact_formula(request.param)
    yield request.param
    _lib.lib().pp_debug_set_exact_formula(0)


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("st", STRATS)
def test_both_scorers_match_golden(g, si, st, exact_formula):
    logits, excl = g[f"s{si}_logits"], g[f"s{si}_exclude"]
    t = torch.from_numpy(logits).to(DEV)
    m = acq.score_map(t, None, st).cpu().numpy()
    np.testing.assert_allclose(m, g[f"s{si}_map_{st}"], rtol=RTOL, atol=ATOL)
    idx, _, _ = acq.score_topk(t, torch.from_numpy(excl), st, 20)
    for b in range(logits.shape[0]):
        assert idx[b].cpu().numpy().tolist() == g[f"s{si}_order_{st}"][b].tolist()


def test_scorers_agree_on_full_size_and_nan_semantics(exact_formula):
    gen = torch.Generator(device=DEV).manual_seed(3)
    logits = torch.randn((2, 19, 256, 512), device=DEV, generator=gen) * 3
    logits[0, :, 5, 7] = torch.tensor([300.0] + [0.0] * 18, device=DEV)      # p underflows -> NaN (query.py:230)
    logits[1, :, 9, 1] = torch.tensor([95.0] + [0.0] * 18, device=DEV)       # denormal p: finite in the reference
    m = acq.score_map(logits, None, "entropy")
    assert torch.isnan(m[0, 5, 7]) and torch.isfinite(m[1, 9, 1])
    assert int(torch.isnan(m).sum()) == 1
    ref = orc.score_map(logits[:, :, :16, :64].contiguous().cpu().numpy(), "entropy")
    got = m[:, :16, :64].cpu().numpy()
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=RTOL, atol=ATOL)


def test_edges(golden_dir):
    e = np.load(os.path.join(golden_dir, "acq_edges.npz"))
    t = torch.from_numpy(e["nan_logits"]).to(DEV)
    ent = acq.score_map(t, None, "entropy")[0].cpu().numpy()
    ref = e["nan_entropy_map"]
    assert np.array_equal(np.isnan(ent), np.isnan(ref))
    np.testing.assert_allclose(ent[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(acq.score_map(t, None, "least_confidence")[0].cpu().numpy(), e["nan_lc_map"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(acq.score_map(t, None, "margin_sampling")[0].cpu().numpy(), e["nan_margin_map"], rtol=RTOL, atol=ATOL)
    idx, val, _ = acq.score_topk(t, None, "entropy", 4)
    idx, val = idx[0].cpu().numpy(), val[0].cpu().numpy()
    assert np.isnan(val[:2]).all() and not np.isnan(val[2:]).any() and idx[0] < idx[1]
    assert set(idx[:2].tolist()) == set(e["nan_top4_idx"][:2].tolist())
    assert idx[2:].tolist() == e["nan_top4_idx"][2:].tolist()
    uc = torch.from_numpy(e["few_uc"]).reshape(1, -1).to(DEV)
    i5, _ = acq.topk_select(uc, 5, True)
    assert sorted(i5[0].cpu().numpy().tolist()) == e["few_sel_k5"].tolist()
    i8, _ = acq.topk_select(uc, 8, True)
    i8 = i8[0].cpu().numpy().tolist()
    extras = [i for i in i8 if i not in set(e["few_sel_k5"].tolist())]
    assert extras == np.flatnonzero(e["few_exclude"].reshape(-1))[:3].tolist()


def test_errors():
    t = torch.zeros(1, 19, 8, 8, device=DEV)
    with pytest.raises(ValueError):
        acq.score_topk(t, None, "entropy", 0)
    with pytest.raises(ValueError):
        acq.score_topk(t, None, "entropy", 65)   # k > H*W
    # (no class count is rejected: beyond 64 the streamed scorers take over - tests/test_acq_wide_gpu.py)
    i65, _, _ = acq.score_topk(torch.zeros(1, 65, 8, 8, device=DEV), None, "entropy", 2)
    assert i65[0].cpu().numpy().tolist() == [0, 1]
    # k == H*W is legal
    idx, _, _ = acq.score_topk(t, None, "entropy", 64)
    assert sorted(idx[0].cpu().numpy().tolist()) == list(range(64))


def test_uncertainty_sampler_from_prob(g):
    logits = torch.from_numpy(g["s0_logits"]).to(DEV)
    prob = torch.softmax(logits, dim=1)
    for st in STRATS:
        got = ppq.UncertaintySampler(st)(prob).cpu().numpy()
        np.testing.assert_allclose(got, g[f"s0_map_{st}"], rtol=RTOL, atol=ATOL)
        assert got.shape == (1, 32, 64)
    assert getattr(ppq.UncertaintySampler, "_entropy")(prob).shape == (1, 32, 64)   # train.py:82 lookup style


# ---------------------------------------------------------------- end to end vs the reference's QuerySelector
def _args(**kw):
    base = dict(dataset_name="cs", debug=False, dir_root="/tmp", experim_name="golden", ignore_index=19,
                mc_n_steps=20, n_classes=19, n_pixels_by_us=20, network_name="deeplab", query_strategy="entropy",
                reverse_order=False, stride_total=8, top_n_percent=0.0, use_mc_dropout=False, vote_type="hard")
    base.update(kw)
    return Namespace(**base)


class _DS:
    def __init__(self, xs, ys, queries, names):
        self.xs, self.ys, self.queries, self.names, self.labelled = xs, ys, queries, names, None

    def label_queries(self, d, nth):
        self.labelled = (d, nth)


class _DL:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        for i in range(len(self.dataset.xs)):
            yield {"x": self.dataset.xs[i][None], "y": self.dataset.ys[i][None], "p_img": [self.dataset.names[i]]}


class _OneConv(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b = w, b

    def forward(self, x):
        return {"pred": torch.nn.functional.conv2d(x, self.w, self.b)}


@pytest.mark.parametrize("st", STRATS)
def test_query_selector_end_to_end_matches_reference(golden_dir, st):
    g4 = np.load(os.path.join(golden_dir, "acq_end_to_end.npz"))
    names = [str(n) for n in g4["names"]]
    ds = _DS(torch.from_numpy(g4[f"{st}_xs"]), torch.from_numpy(g4[f"{st}_ys"]), list(g4[f"{st}_prev"]), names)
    model = _OneConv(torch.from_numpy(g4[f"{st}_W"]).to(DEV), torch.from_numpy(g4[f"{st}_b"]).to(DEV))
    with tempfile.TemporaryDirectory() as td:
        qs = ppq.QuerySelector(_args(query_strategy=st, dir_root=td), _DL(ds), device=torch.device(DEV))
        dq = qs(nth_query=1, model=model)
        import pickle
        stats = pickle.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
    assert list(dq.keys()) == names
    for i, n in enumerate(names):
        assert dq[n]["height"] == 40 and dq[n]["width"] == 56
        np.testing.assert_array_equal(dq[n]["x_coords"], g4[f"{st}_x_{i}"])
        np.testing.assert_array_equal(dq[n]["y_coords"], g4[f"{st}_y_{i}"])
    np.testing.assert_array_equal(np.array([stats["label_distribution"][l] for l in range(19)]), g4[f"{st}_stats_label_cnt"])
    assert abs(stats["avg_entropy"] - float(g4[f"{st}_stats_avg_entropy"])) < 1e-5
    assert abs(stats["avg_n_unique_labels"] - float(g4[f"{st}_stats_avg_n_unique"])) < 1e-9
    assert abs(stats["avg_spatial_coverage"] - float(g4[f"{st}_stats_avg_cov"])) < 1e-9
    assert ds.labelled is not None and ds.labelled[1] == 1


@pytest.mark.parametrize("bs", [1, 3])
@pytest.mark.parametrize("mode,kw", [("k20", dict(n_pixels_by_us=20)), ("top5", dict(n_pixels_by_us=10, top_n_percent=0.05)),
                                     ("rev", dict(n_pixels_by_us=10, top_n_percent=0.05, reverse_order=True))])
def test_query_selector_random_strategy_matches_reference(golden_dir, mode, kw, bs):
    """args.py:27 `random` through QuerySelector.__call__ (query.py:190-204,242-247): host torch.rand map per image, fill
    1.0, k smallest; fixture from the imported reference with the torch / numpy seeds it was drawn under."""
    gr = np.load(os.path.join(golden_dir, "acq_random.npz"))
    seed = int(gr[f"{mode}_seed"])
    names = [f"/data/rnd_{i:03d}.png" for i in range(3)]
    ds = _DS(torch.from_numpy(gr[f"{mode}_xs"]), torch.from_numpy(gr[f"{mode}_ys"]), list(gr[f"{mode}_prev"]), names)
    model = _OneConv(torch.from_numpy(gr[f"{mode}_W"]).to(DEV), torch.from_numpy(gr[f"{mode}_b"]).to(DEV))
    with tempfile.TemporaryDirectory() as td:
        qs = ppq.QuerySelector(_args(query_strategy="random", dir_root=td, query_batch_size=bs, **kw), _DL(ds),
                               device=torch.device(DEV))
        torch.manual_seed(seed + 1000)
        np.random.seed(seed + 2000)
        dq = qs(nth_query=1, model=model)
        import pickle
        stats = pickle.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
    for i, n in enumerate(names):
        np.testing.assert_array_equal(dq[n]["x_coords"], gr[f"{mode}_x_{i}"])
        np.testing.assert_array_equal(dq[n]["y_coords"], gr[f"{mode}_y_{i}"])
        picked = np.zeros((40, 56), bool)
        picked[dq[n]["y_coords"], dq[n]["x_coords"]] = True
        assert not (picked & (gr[f"{mode}_prev"][i] | (gr[f"{mode}_ys"][i] == 19))).any()
    np.testing.assert_array_equal(np.array([stats["label_distribution"][l] for l in range(19)]), gr[f"{mode}_stats_label_cnt"])
    assert abs(stats["avg_entropy"] - float(gr[f"{mode}_stats_avg_entropy"])) < 1e-5
    assert abs(stats["avg_spatial_coverage"] - float(gr[f"{mode}_stats_avg_cov"])) < 1e-9
    assert ds.labelled is not None


def test_random_strategy_has_no_score_kernel():
    with pytest.raises(ValueError):
        acq.score_topk(torch.zeros(1, 19, 8, 8, device=DEV), None, "random", 4)


# ---------------------------------------------------------------- full-size, size-independent properties
@pytest.mark.parametrize("st,shape", [("entropy", (8, 19, 256, 512)), ("margin_sampling", (4, 21, 320, 320)),
                                       ("least_confidence", (1, 19, 1024, 2048))])
def test_full_size_properties(st, shape):
    B, C, H, W = shape
    gen = torch.Generator(device=DEV).manual_seed(0)
    logits = torch.randn(shape, device=DEV, generator=gen) * 3
    excl = torch.rand((B, H, W), device=DEV, generator=gen) < 0.05
    k = 20
    idx, val, omap = acq.score_topk(logits, excl, st, k, return_map=True)
    largest = st != "margin_sampling"
    # (1) value-sorted, unique, never an excluded pixel
    v = val if largest else -val
    assert (v[:, :-1] >= v[:, 1:]).all()
    flat = omap.reshape(B, -1)
    assert (torch.gather(flat, 1, idx.long()) == val).all()
    assert not torch.gather(excl.reshape(B, -1), 1, idx.long()).any()
    for b in range(B):
        assert len(set(idx[b].tolist())) == k
    # (2) k-th value is the true k-th order statistic of the map; every other pixel is no better
    kth = val[:, -1:]
    better = (flat > kth) if largest else (flat < kth)
    assert (better.sum(dim=1) <= k - 1).all()
    # (3) idempotence / determinism: same call twice is bit-identical; layout does not matter
    idx2, val2, _ = acq.score_topk(logits.contiguous(memory_format=torch.channels_last), excl, st, k)
    assert torch.equal(idx, idx2) and torch.equal(val, val2)
    # (4) batch independence: image 0 alone gives the same answer
    idx0, _, _ = acq.score_topk(logits[:1], excl[:1], st, k)
    assert torch.equal(idx0[0], idx[0])
    # (5) a sample of the map agrees with the oracle
    sub = logits[0, :, :8, :64].contiguous().cpu().numpy()[None]
    np.testing.assert_allclose(acq.score_map(logits[:1, :, :8, :64], None, st).cpu().numpy(), orc.score_map(sub, st),
                               rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("st", STRATS)
def test_quantised_select_equals_the_radix_select(st):
    """k > 48 with a known score range (query.py:36 top_n_percent: k = 5 % of the pixels): one linear-bin histogram pass +
    compaction + register / shuffle sort.  Must return exactly what the four-pass radix select + LDS sort returns
    (pp_debug_set_reduce_mode bit 9 switches it off) - on random maps, with exclusions, with heavy ties (overflow -> the exact
    fallback launch), with NaN scores, and in images whose every pixel is excluded - and what the oracle's stable sort gives."""
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(7)
    cases = []
    B, C, H, W = 3, 19, 256, 512
    logits = torch.randn((B, C, H, W), device=DEV, generator=gen) * 3
    excl = torch.rand((B, H, W), device=DEV, generator=gen) < 0.05
    cases.append(("random", logits, excl, 6553))
    cases.append(("random-small-k", logits[:, :, :64, :96].contiguous(), excl[:, :64, :96].contiguous(), 307))
    tied = torch.round(torch.randn((2, C, 128, 256), device=DEV, generator=gen))           # few distinct class vectors: heavy ties
    cases.append(("ties", tied, None, 1638))
    const = torch.zeros((2, C, 128, 256), device=DEV)
    const[1] = logits[0, :, :128, :256]
    ex2 = torch.zeros((2, 128, 256), dtype=torch.bool, device=DEV)
    ex2[1, :100] = True                                                                    # 78 % of image 1 excluded: one huge bin
    cases.append(("constant+mostly-excluded", const, ex2, 1638))
    if st == "entropy":
        nanl = logits[:2, :, :64, :128].clone()
        nanl[0, 0, 3, 5:40] = 200.0                                                         # p -> 0 for the others: 0 * log 0 = NaN (query.py:230)
        cases.append(("nan", nanl, None, 409))
    for Cx in (11, 21):                                                                      # the other two dataset class counts (their own scorer instantiations)
        lx = torch.randn((2, Cx, 96, 160), device=DEV, generator=gen) * 3
        cases.append((f"random-C{Cx}", lx, torch.rand((2, 96, 160), device=DEV, generator=gen) < 0.05, 768))
    for name, lg, ex, k in cases:
        try:
            L.pp_debug_set_reduce_mode(512)
            ref = acq.score_topk(lg, ex, st, k, return_map=True)
            L.pp_debug_set_reduce_mode(1024)       # the histogram in its own pass over the map (select_qhist_kernel), as before round 5
            sep = acq.score_topk(lg, ex, st, k, return_map=True)
            L.pp_debug_set_reduce_mode(0)          # default: the scorer launch counts its scores into the histogram (acq_kernel<..., HIST>)
            got = acq.score_topk(lg, ex, st, k, return_map=True)
        finally:
            L.pp_debug_set_reduce_mode(0)
        for a, b, c in zip(ref, got, sep):
            for other in (b, c):
                assert torch.equal(a, other) or (name == "nan" and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(other, nan=-7.0))), (name, st, k)
        # oracle: stable sort of the device's own map (ties -> lower index first, NaN first for largest)
        dmap = got[2].cpu().numpy()
        for b in range(lg.shape[0]):
            e_idx, _ = orc.topk(dmap[b], k, st != "margin_sampling")
            assert got[0][b].cpu().numpy().tolist() == e_idx.tolist(), (name, st, b)


@pytest.mark.parametrize("st", STRATS)
def test_list_select_equals_the_map_select(st):
    """Large k without a caller's map (query.py:36 top_n_percent as QuerySelector calls it): a sampled threshold key per image, the
    scorer writes only the (key, index) words beyond it, topk_lsel_kernel selects from the lists; whatever the sample says, the picks
    and values must be those of the map-writing scorer + topk_qsel_kernel (pp_debug_set_reduce_mode bit 11) - with exclusions, ties
    and constant maps (flagged -> the exact fallback), NaN scores, ragged sizes, both pixel tiles, the XCD block order, and with the
    sample's aim forced far too high (every image falls back) and far too low (a fifth of the pixels pass)."""
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(11)
    C = 19
    cases = []
    logits = torch.randn((8, C, 256, 512), device=DEV, generator=gen) * 3
    excl = torch.rand((8, 256, 512), device=DEV, generator=gen) < 0.05
    cases.append(("random-8px-tile", logits, excl, 6553))
    cases.append(("random-4px-tile", logits[:3], excl[:3], 6553))
    cases.append(("no-exclusion", logits[:2], None, 4000))
    cases.append(("ragged", logits[:2, :, :100, :172].contiguous(), excl[:2, :100, :172].contiguous(), 860))
    tied = torch.round(torch.randn((2, C, 128, 256), device=DEV, generator=gen))
    cases.append(("ties", tied, None, 1638))
    const = torch.zeros((2, C, 128, 256), device=DEV)
    const[1] = logits[0, :, :128, :256]
    ex2 = torch.zeros((2, 128, 256), dtype=torch.bool, device=DEV)
    ex2[1, :100] = True
    cases.append(("constant+mostly-excluded", const, ex2, 1638))
    smooth = torch.nn.functional.interpolate(torch.randn((2, C, 16, 32), device=DEV, generator=gen) * 4, size=(256, 512), mode="bilinear")
    cases.append(("smooth", smooth.contiguous(), None, 6553))                                 # spatially correlated scores: the sample's locations are not independent
    if st == "entropy":
        nanl = logits[:2, :, :128, :128].clone()
        nanl[0, 0, 3, 5:40] = 200.0
        cases.append(("nan", nanl.contiguous(), None, 819))
    for Cx in (11, 21):
        lx = torch.randn((2, Cx, 128, 160), device=DEV, generator=gen) * 3
        cases.append((f"random-C{Cx}", lx, torch.rand((2, 128, 160), device=DEV, generator=gen) < 0.05, 1024))
    cases.append(("xcd-order", torch.randn((1, C, 1024, 1024), device=DEV, generator=gen) * 3, None, 5000))
    for name, lg, ex, k in cases:
        try:
            L.pp_debug_set_reduce_mode(2048)
            ref = acq.score_topk(lg, ex, st, k)
            outs = []
            for mode in (0, 1 << 12, 63 << 12, 1 << 18, (2 << 18) | (2 << 20)):    # default; aim of k / 16 and of 3.9 k passing pixels; 64 x 16- and 256 x 4-pixel samples
                L.pp_debug_set_reduce_mode(mode)
                outs.append(acq.score_topk(lg, ex, st, k))
        finally:
            L.pp_debug_set_reduce_mode(0)
        for got in outs:
            assert torch.equal(ref[0], got[0]), (name, st, k)
            assert torch.equal(torch.nan_to_num(ref[1], nan=-7.0), torch.nan_to_num(got[1], nan=-7.0)), (name, st, k)


@pytest.mark.parametrize("C", [11, 19, 21])
def test_block_order_and_tile_variants_are_bit_identical(C):
    """The XCD-contiguous block order (default for class planes >= 4 MB, forced here on small ragged ones), 4 / 8 pixels per
    thread and 2 / 3 / 4 waves per SIMD are schedules of the SAME arithmetic: picks, scores and the map must not move a bit."""
    L = _lib.lib()
    gen = torch.Generator(device=DEV).manual_seed(C)
    for B, H, W, k in ((3, 36, 52, 20), (2, 200, 328, 7), (9, 64, 128, 48)):
        logits = torch.randn((B, C, H, W), device=DEV, generator=gen) * 3
        excl = torch.rand((B, H, W), device=DEV, generator=gen) < 0.1
        for st in STRATS:
            L.pp_debug_set_acq_tuning(0, 0)
            ref = acq.score_topk(logits, excl, st, k, return_map=True)
            try:
                for occ in (2, 3, 4):
                    for ppt in (4, 8):
                        for xcd in (1, 2):
                            L.pp_debug_set_acq_tuning(occ | (xcd << 8), ppt)
                            got = acq.score_topk(logits, excl, st, k, return_map=True)
                            for a, b in zip(ref, got):
                                assert torch.equal(a, b), (C, st, occ, ppt, xcd)
            finally:
                L.pp_debug_set_acq_tuning(0, 0)


def test_query_selector_batched_forward_gives_identical_queries(golden_dir):
    """query_batch_size > 1 (several equal-sized images per forward) must not change a single coordinate."""
    g4 = np.load(os.path.join(golden_dir, "acq_end_to_end.npz"))
    st = "entropy"
    names = [str(n) for n in g4["names"]]
    model = _OneConv(torch.from_numpy(g4[f"{st}_W"]).to(DEV), torch.from_numpy(g4[f"{st}_b"]).to(DEV))
    outs = []
    for bs in (1, 2, 3):
        ds = _DS(torch.from_numpy(g4[f"{st}_xs"]), torch.from_numpy(g4[f"{st}_ys"]), list(g4[f"{st}_prev"]), names)
        with tempfile.TemporaryDirectory() as td:
            qs = ppq.QuerySelector(_args(query_strategy=st, dir_root=td, query_batch_size=bs), _DL(ds), device=torch.device(DEV))
            outs.append(qs(nth_query=1, model=model))
    for n in names:
        for o in outs[1:]:
            np.testing.assert_array_equal(o[n]["x_coords"], outs[0][n]["x_coords"])
            np.testing.assert_array_equal(o[n]["y_coords"], outs[0][n]["y_coords"])
        np.testing.assert_array_equal(outs[0][n]["x_coords"], g4[f"{st}_x_{names.index(n)}"])


# ---------------------------------------------------------------- the other __call__ branches, pinned on the reference
class _Conv3(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b = w, b

    def forward(self, x):
        return {"pred": torch.nn.functional.conv2d(x, self.w, self.b, padding=1)}


class _DLNoY(_DL):
    def __iter__(self):
        for i in range(len(self.dataset.xs)):
            yield {"x": self.dataset.xs[i][None], "p_img": [self.dataset.names[i]]}


@pytest.mark.parametrize("bs", [1, 2])
def test_query_selector_voc_reflect_pad_branch(golden_dir, bs):
    """query.py:171-174,190: voc images are reflect-padded to a multiple of stride_total, logits cropped back."""
    gb = np.load(os.path.join(golden_dir, "acq_branches.npz"))
    names = [f"/voc/img_{i}.jpg" for i in range(2)]
    ds = _DS(torch.from_numpy(gb["voc_xs"]), torch.from_numpy(gb["voc_ys"]), list(gb["voc_prev"]), names)
    model = _Conv3(torch.from_numpy(gb["voc_W"]).to(DEV), torch.from_numpy(gb["voc_b"]).to(DEV))
    with tempfile.TemporaryDirectory() as td:
        a = _args(query_strategy="margin_sampling", dir_root=td, dataset_name="voc", n_classes=21, ignore_index=255,
                  n_pixels_by_us=10, query_batch_size=bs)
        dq = ppq.QuerySelector(a, _DL(ds), device=torch.device(DEV))(nth_query=1, model=model)
    for i, n in enumerate(names):
        assert dq[n]["height"] == 37 and dq[n]["width"] == 53
        np.testing.assert_array_equal(dq[n]["x_coords"], gb[f"voc_x_{i}"])
        np.testing.assert_array_equal(dq[n]["y_coords"], gb[f"voc_y_{i}"])


def test_query_selector_human_labels_branch(golden_dir):
    """query.py:145-146,196-197,215: previous labels are int64 maps (labelled where != ignore_index); no 'y', no
    stats, and dataset.label_queries is NOT called."""
    gb = np.load(os.path.join(golden_dir, "acq_branches.npz"))
    names = [f"/cv/img_{i}.png" for i in range(2)]
    ds = _DS(torch.from_numpy(gb["hl_xs"]), None, None, names)
    ds.list_labelled_queries = list(gb["hl_labelled"])
    model = _OneConv(torch.from_numpy(gb["hl_W"]).to(DEV), torch.from_numpy(gb["hl_b"]).to(DEV))
    with tempfile.TemporaryDirectory() as td:
        a = _args(query_strategy="least_confidence", dir_root=td, dataset_name="cv", n_classes=11, ignore_index=11,
                  n_pixels_by_us=12)
        dq = ppq.QuerySelector(a, _DLNoY(ds), device=torch.device(DEV))(nth_query=2, model=model, human_labels=True)
        assert not os.path.exists(f"{td}/checkpoints/golden/2_query/query_stats.pkl")
    for i, n in enumerate(names):
        np.testing.assert_array_equal(dq[n]["x_coords"], gb[f"hl_x_{i}"])
        np.testing.assert_array_equal(dq[n]["y_coords"], gb[f"hl_y_{i}"])
        picked = np.zeros((24, 40), bool)
        picked[dq[n]["y_coords"], dq[n]["x_coords"]] = True
        assert not (picked & (gb["hl_labelled"][i] != 11)).any()
    assert ds.labelled is None


class _DropModel(torch.nn.Module):
    """1x1 classifier with an element dropout in front; turn_on_dropout as networks/deeplab.py:41-46."""
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b, self.drop = w, b, torch.nn.Dropout(0.3)

    def turn_on_dropout(self):
        self.drop.train()

    def forward(self, x):
        return {"pred": torch.nn.functional.conv2d(self.drop(x), self.w, self.b)}


@pytest.mark.parametrize("chunk", [32, 3])
def test_query_selector_mc_dropout_mean_of_maps(chunk):
    """MC-dropout branch (query.py:176-188 as intended — upstream's version crashes): the selector must
    pick top-k of the MEAN uncertainty map over mc_n_steps stochastic passes.  The passes of one image run as ONE forward
    over copies of it (`mc_chunk` at a time); re-derive that with the same torch RNG."""
    torch.manual_seed(5)
    C, h, w, steps, k = 7, 24, 40, 4, 9
    W, b = (torch.randn(C, 3, 1, 1) * 2).to(DEV), (torch.randn(C) * .5).to(DEV)
    xs, ys = torch.randn(2, 3, h, w), torch.randint(0, C, (2, h, w))
    prev = [np.zeros((h, w), bool) for _ in range(2)]
    prev[0][3, 4] = True
    names = ["/a.png", "/b.png"]
    model = _DropModel(W, b)
    with tempfile.TemporaryDirectory() as td:
        a = _args(query_strategy="entropy", dir_root=td, n_classes=C, ignore_index=C, n_pixels_by_us=k,
                  use_mc_dropout=True, mc_n_steps=steps, mc_chunk=chunk)
        torch.manual_seed(11); torch.cuda.manual_seed(11)
        dq = ppq.QuerySelector(a, _DL(_DS(xs, ys, prev, names)), device=torch.device(DEV))(nth_query=1, model=model)
    torch.manual_seed(11); torch.cuda.manual_seed(11)
    model.eval(); model.turn_on_dropout()
    for i, n in enumerate(names):
        uc = torch.zeros(h, w, device=DEV)
        with torch.no_grad():
            left = steps
            while left > 0:
                t = min(left, chunk)
                lg = model(xs[i:i + 1].to(DEV).expand(t, -1, -1, -1).contiguous())["pred"]
                uc += acq.score_map(lg, None, "entropy").sum(dim=0)
                left -= t
        uc /= steps
        uc[torch.from_numpy(prev[i]).to(DEV)] = 0.0
        want = uc.flatten().topk(k).indices.cpu().numpy()
        got = dq[n]["y_coords"].astype(np.int64) * w + dq[n]["x_coords"]
        assert set(got.tolist()) == set(want.tolist())


# ---------------------------------------------------------------- SURVEY.md §8f rank 1 through the real network
@pytest.mark.parametrize("dataset,st,top_n,backbone", [("cs", "entropy", 0.0, "mobilenet"), ("voc", "margin_sampling", 0.0, "mobilenet"),
                                                       ("cs", "least_confidence", 0.05, "mobilenet"), ("cs", "entropy", 0.0, "resnet")])
def test_query_selector_fused_lowres_gives_identical_queries_and_stats(monkeypatch, dataset, st, top_n, backbone):
    """DeepLab exposes forward_lowres: the selector then scores straight from the 1/4-resolution classifier output
    (pp_acq_lowres_score_topk).  Coordinates and QueryStats must equal the full-resolution-logits path exactly —
    including the VOC reflect-pad / crop branch (query.py:171-174,190) and the top-n-percent mode (query.py:36,62-64)."""
    from pixelpick_amd.networks.deeplab import DeepLab
    C = 21 if dataset == "voc" else 19
    h, w = (77, 90) if dataset == "voc" else (64, 96)
    torch.manual_seed(3)
    net_args = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, use_aspp=True, use_softmax=False, use_img_inp=False)
    # backbone "resnet": the assembled DeepLabv3+-ResNet50 (SURVEY.md 0.1 extra) takes the same acquisition path
    model = (DeepLab(net_args) if backbone == "mobilenet" else DeepLab(net_args, backbone="resnet", output_stride=8)).to(DEV)
    n = 5
    xs, ys = torch.randn(n, 3, h, w), torch.randint(0, C + 1, (n, h, w))
    ys[ys == C] = 255 if dataset == "voc" else C
    rng = np.random.RandomState(0)
    prev = [rng.rand(h, w) < 0.01 for _ in range(n)]
    names = [f"/img{i}.png" for i in range(n)]
    outs, stats = [], []
    for fused in (True, False):
        monkeypatch.setattr(ppq, "FUSED_LOWRES", fused)
        called = []
        orig = model.forward_lowres
        monkeypatch.setattr(model, "forward_lowres", lambda x, _o=orig: (called.append(1), _o(x))[1], raising=False)
        with tempfile.TemporaryDirectory() as td:
            a = _args(query_strategy=st, dir_root=td, dataset_name=dataset, n_classes=C, ignore_index=255 if dataset == "voc" else C,
                      top_n_percent=top_n, query_batch_size=2)
            np.random.seed(4)
            qs = ppq.QuerySelector(a, _DL(_DS(xs, ys, prev, names)), device=torch.device(DEV))
            outs.append(qs(nth_query=1, model=model))
            stats.append(qs.query_stats)
        assert bool(called) == fused
        monkeypatch.undo()
    for nme in names:
        np.testing.assert_array_equal(outs[0][nme]["x_coords"], outs[1][nme]["x_coords"])
        np.testing.assert_array_equal(outs[0][nme]["y_coords"], outs[1][nme]["y_coords"])
        assert outs[0][nme]["height"] == h and outs[0][nme]["width"] == w
    assert len(stats[0].list_entropy) >= n and stats[0].list_entropy == stats[1].list_entropy
    assert stats[0].list_n_unique_labels == stats[1].list_n_unique_labels
    assert stats[0].list_spatial_coverage == stats[1].list_spatial_coverage
    assert stats[0].dict_label_cnt == stats[1].dict_label_cnt


def test_query_selector_pipelined_round_equals_the_strict_order(monkeypatch):
    """The default configuration only ENQUEUES a batch in flush() and finishes it (read-back, masks, statistics, codec) after
    the next batch has been launched; coordinates, statistics and the dataset side effect must equal the strict per-batch order."""
    from pixelpick_amd.networks.deeplab import DeepLab
    C, h, w, n = 19, 64, 96, 7
    torch.manual_seed(9)
    model = DeepLab(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, use_aspp=True, use_softmax=False, use_img_inp=False)).to(DEV)
    xs, ys = torch.randn(n, 3, h, w), torch.randint(0, C + 1, (n, h, w))
    rng = np.random.RandomState(1)
    prev = [rng.rand(h, w) < 0.02 for _ in range(n)]
    names = [f"/img{i}.png" for i in range(n)]
    outs = []
    for pipe in (True, False):
        monkeypatch.setattr(ppq, "QUERY_PIPELINE", pipe)
        ds = _DS(xs, ys, prev, names)
        with tempfile.TemporaryDirectory() as td:
            qs = ppq.QuerySelector(_args(query_strategy="entropy", dir_root=td, query_batch_size=3), _DL(ds), device=torch.device(DEV))
            dq = qs(nth_query=1, model=model)
        st = qs.query_stats
        outs.append((dq, st.list_entropy, st.list_n_unique_labels, st.list_spatial_coverage, st.dict_label_cnt, ds.labelled))
    a, b = outs
    assert list(a[0].keys()) == list(b[0].keys()) == names
    for nme in names:
        np.testing.assert_array_equal(a[0][nme]["x_coords"], b[0][nme]["x_coords"])
        np.testing.assert_array_equal(a[0][nme]["y_coords"], b[0][nme]["y_coords"])
    assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    assert a[5][1] == b[5][1] == 1 and list(a[5][0].keys()) == names


# ------------------------------------------------------------------------ full size, oracle's OWN picks (gap-guarded by construction)
def _guarded_full_size_case(C, H, W, st, seed, k=20, steps=None):
    """Random logits at a BASELINE size whose k + 1 leading scores are separated by construction, so that the picks of the
    oracle (reference operation order, host libm) are the only right answer for any evaluation within the score tolerance:
    every random pixel that could compete is made confident (+6 on its arg-max logit), then k + 6 planted pixels get the class
    vector (a_j, 0, ..., 0): a_j = 1 + 0.04 j for entropy (2.86 .. 2.64 at C = 19) and least confidence (0.87 .. 0.70), falling with
    j; a_j = 0.02 + 0.004 j for the margin (e^a - 1) / (e^a + C - 1) = 0.001 .. 0.0065, rising with j - each step >= 1e-3 relative.
    The guard is asserted on the oracle's map, not assumed."""
    rng = np.random.RandomState(seed)
    logits = (rng.randn(1, C, H, W) * 3).astype(np.float32)
    excl = (rng.rand(1, H, W) < 0.05).astype(np.uint8)
    largest = st != "margin_sampling"
    a0, da, slack = steps or ((1.0, 0.04, 0.05) if largest else (0.02, 0.004, 0.01))      # (wide heads need larger steps: test_acq_wide_gpu.py)
    m = orc.score_map(logits, st)[0]
    a_max = a0 + da * (k + 6)
    probe = np.zeros((1, C, 1, 1), dtype=np.float32)
    probe[0, 0, 0, 0] = a_max
    edge = float(orc.score_map(probe, st)[0, 0, 0])                      # the weakest planted score
    rivals = (m > edge - slack) if largest else (m < edge + slack)
    ys, xs = np.nonzero(rivals)
    top = logits[0, :, ys, xs].argmax(axis=1)
    logits[0, top, ys, xs] += 6.0
    free = np.flatnonzero((excl[0] == 0).reshape(-1) & ~rivals.reshape(-1))
    spots = rng.choice(free, k + 6, replace=False)
    for j, p in enumerate(spots):
        logits[0, :, p // W, p % W] = 0.0
        logits[0, 0, p // W, p % W] = a0 + da * j
    o_idx, o_val, o_map = orc.score_topk(logits, excl, st, k, want_map=True)
    srt = np.sort(orc.apply_exclude(o_map, excl, st).reshape(-1).astype(np.float64))
    lead = srt[::-1][:k + 1] if largest else srt[:k + 1]
    gaps = np.abs(np.diff(lead)) / np.maximum(np.abs(lead[1:]), np.abs(lead[:-1]))
    assert gaps.min() >= 1e-3, gaps.min()
    assert o_idx[0].tolist() == spots[:k].tolist()
    return logits, excl, o_idx, o_val


@pytest.mark.parametrize("exact", [0, 1], ids=["default-scorer", "reference-order-scorer"])
@pytest.mark.parametrize("C,H,W,st", [(19, 256, 512, "entropy"), (19, 256, 512, "least_confidence"), (19, 256, 512, "margin_sampling"),
                                       (21, 320, 320, "margin_sampling"), (19, 1024, 2048, "least_confidence"), (19, 1024, 2048, "entropy")])
def test_full_size_picks_equal_the_oracles_on_gap_guarded_input(C, H, W, st, exact):
    """query.py:229-239,57-61 at the BASELINE sizes against the ORACLE'S picks (not the device's own map): value-sorted indices
    identical, values within the score tolerance - for the default scorer and for the reference operation order."""
    logits, excl, o_idx, o_val = _guarded_full_size_case(C, H, W, st, seed=H + C + len(st))
    _lib.lib().pp_debug_set_exact_formula(exact)
    try:
        for name, t in (("nchw", torch.from_numpy(logits).to(DEV)),
                        ("nhwc", torch.from_numpy(logits).to(DEV).contiguous(memory_format=torch.channels_last))):
            idx, val, _ = acq.score_topk(t, torch.from_numpy(excl), st, 20)
            assert idx[0].cpu().numpy().tolist() == o_idx[0].tolist(), name
            np.testing.assert_allclose(val[0].cpu().numpy(), o_val[0], rtol=RTOL, atol=ATOL)
    finally:
        _lib.lib().pp_debug_set_exact_formula(0)
