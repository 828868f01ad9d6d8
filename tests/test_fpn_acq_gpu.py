"""The acquisition tail of the reference's ResNet50 model (BASELINE configs[2-4]) on the HIP path:
FPNSeg forward -> softmax -> score -> exclusion -> top-k (networks/model.py:6-14, decoders.py:57-77,90-101, query.py:144-221).

`FPNSeg.forward_lowres` stops in front of the decoder's last x2 interpolation (classifier on the half-resolution branch
sum) and `pp_acq_lowres_score_topk(align_corners=0)` interpolates, scores and selects per tile - the four full-resolution
128-channel branch maps, "emb" and the full-size logits are never written.

* the whole round against `tests/golden/acq_fpn_round.npz` (tools/gen_golden_acq.py --fpn: the IMPORTED reference FPNSeg
  under the reference's QuerySelector on formula weights / images): same picks, same QueryStats - fused and unfused;
* fused == unfused picks at 256x512 and 1024x2048, peak memory and time of both."""
import os
import pickle
import tempfile
import time
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch

import formula_init as fi
from oracle import acq as orc
from pixelpick_amd import acquisition as acq
from pixelpick_amd import engine as E
from pixelpick_amd import query as ppq
from pixelpick_amd.utils.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fpn(C=19):
    a = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="FPN", weight_type="random",
                  use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(a)
    m.load_state_dict(fi.formula_state_dict(m.state_dict()))
    return m.to(DEV).eval()


def _args(**kw):
    base = dict(dataset_name="cs", debug=False, dir_root="/tmp", experim_name="golden", ignore_index=19,
                mc_n_steps=20, n_classes=19, n_pixels_by_us=20, network_name="FPN", query_strategy="entropy",
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


CASES = [("cs_entropy", "cs", "entropy"), ("cs_least_confidence", "cs", "least_confidence"), ("voc_margin", "voc", "margin_sampling")]


@pytest.mark.parametrize("fused", [True, False], ids=["half-resolution-tail", "reference-order"])
@pytest.mark.parametrize("tag,ds_name,st", CASES, ids=[c[0] for c in CASES])
def test_fpn_acquisition_round_matches_reference(golden_dir, monkeypatch, tag, ds_name, st, fused):
    g = np.load(os.path.join(golden_dir, "acq_fpn_round.npz"))
    C, ign = [int(v) for v in g[f"{tag}_meta"]]
    sizes = [tuple(int(v) for v in s) for s in g[f"{tag}_sizes"]]
    names = [str(n) for n in g[f"{tag}_names"]]
    keys = [str(k) for k in g[f"{tag}_keys"]]
    xs = [fi.formula_input(1, h, w, key=k)[0] for k, (h, w) in zip(keys, sizes)]
    ys = [torch.from_numpy(g[f"{tag}_y_{i}"].astype(np.int64)) for i in range(len(sizes))]
    prev = [np.unpackbits(g[f"{tag}_prev_{i}"])[:h * w].reshape(h, w).astype(bool) for i, (h, w) in enumerate(sizes)]
    monkeypatch.setattr(ppq, "FUSED_LOWRES", fused)
    model = _fpn(C)
    ds = _DS(xs, ys, prev, names)
    with tempfile.TemporaryDirectory() as td:
        qs = ppq.QuerySelector(_args(query_strategy=st, dir_root=td, dataset_name=ds_name, n_classes=C, ignore_index=ign),
                               _DL(ds), device=torch.device(DEV))
        dq = qs(nth_query=1, model=model)
        stats = pickle.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
    assert list(dq.keys()) == names
    for i, n in enumerate(names):
        assert (dq[n]["height"], dq[n]["width"]) == sizes[i]
        np.testing.assert_array_equal(dq[n]["x_coords"], g[f"{tag}_xc_{i}"])
        np.testing.assert_array_equal(dq[n]["y_coords"], g[f"{tag}_yc_{i}"])
    np.testing.assert_array_equal(np.array([stats["label_distribution"][l] for l in range(C)]), g[f"{tag}_stats_label_cnt"])
    assert abs(stats["avg_entropy"] - float(g[f"{tag}_stats_avg_entropy"])) < 2e-4
    assert abs(stats["avg_n_unique_labels"] - float(g[f"{tag}_stats_avg_n_unique"])) < 1e-9
    assert abs(stats["avg_spatial_coverage"] - float(g[f"{tag}_stats_avg_cov"])) < 1e-9
    assert ds.labelled is not None and ds.labelled[1] == 1


def test_fpn_forward_lowres_is_the_classifier_in_front_of_the_last_interpolation():
    """up2(forward_lowres) == forward()["pred"] to fp32 rounding (linear ops commute; decoders.py:75-77,101), for an even and for
    a ragged (voc-padded) size; the fused scorer's map equals the scorer on those logits to the same rounding."""
    m = _fpn(21)
    for (H, W) in [(64, 96), (56, 40)]:
        x = fi.formula_input(2, H, W, key=f"fpnlow{H}").to(DEV)
        with torch.no_grad():
            low, size = m.forward_lowres(x)
            pred = m(x)["pred"]
        assert size == (H, W) and tuple(low.shape) == (2, H // 2, W // 2, 21)
        up = torch.nn.functional.interpolate(low.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear")
        assert (up - pred).abs().max().item() <= 2e-5 * pred.abs().max().item()
        for st in ["entropy", "least_confidence", "margin_sampling"]:
            _, _, mf = acq.score_topk_lowres(low, size, None, st, 0, align_corners=False)
            mt = acq.score_map(pred, None, st)
            np.testing.assert_allclose(mf.cpu().numpy(), mt.cpu().numpy(), rtol=1e-4, atol=2e-5)
            # and bit-exactly the scorer on pp_bilinear_fwd's interpolation of the same low-resolution logits
            mu = acq.score_map(E.bilinear(E.Tape(False), E.Var(low), size, False, 2.0, out_nchw=True).t, None, st)
            assert torch.equal(mf, mu), st


@pytest.mark.parametrize("H,W,B,st", [(256, 512, 4, "entropy"), (1024, 2048, 1, "least_confidence")],
                         ids=["configs2-256x512", "configs4-1024x2048"])
def test_fpn_fused_tail_at_the_baseline_sizes(H, W, B, st):
    """Picks of the half-resolution tail == picks of the reference-order path wherever the device's own full-size map has a
    k-th / (k+1)-th gap above the rounding difference; always the exact top-k of the tail's own map (oracle's stable sort);
    peak memory and time of both paths."""
    C, k = 19, 20
    m = _fpn(C)
    x = fi.formula_input(B, H, W, key=f"fpnfull{H}").to(DEV)
    excl = torch.zeros((B, H, W), dtype=torch.uint8, device=DEV)
    excl[:, ::7, ::5] = 1

    def fused():
        low, size = m.forward_lowres(x)
        return acq.score_topk_lowres(low, size, excl, st, k, align_corners=False)

    def unfused():
        return acq.score_topk(m(x)["pred"], excl, st, k)

    res = {}
    for name, fn in (("fused", fused), ("unfused", unfused)):
        with torch.no_grad():
            fn()                                            # warm: workspaces, plans
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            t0 = time.perf_counter()
            idx, val, _ = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[name] = (idx.cpu().numpy(), val.cpu().numpy(), dt, (torch.cuda.max_memory_allocated() - base) / 2 ** 30)
    print(f"\n[FPNSeg {H}x{W} B={B} {st}] forward + top-{k}: fused {res['fused'][2] * 1e3:.1f} ms, peak +{res['fused'][3]:.2f} GiB; "
          f"unfused {res['unfused'][2] * 1e3:.1f} ms, peak +{res['unfused'][3]:.2f} GiB")
    assert res["fused"][3] < 0.5 * res["unfused"][3]
    with torch.no_grad():
        low, size = m.forward_lowres(x)
        _, _, fmap = acq.score_topk_lowres(low, size, excl, st, 0, align_corners=False)
    fmap = fmap.cpu().numpy()
    for b in range(B):
        e_idx, _ = orc.topk(fmap[b], k, True)
        assert res["fused"][0][b].tolist() == e_idx.tolist()
        assert not excl[b].reshape(-1)[torch.from_numpy(e_idx).to(DEV).long()].any()
        srt = np.sort(fmap[b].reshape(-1))[::-1][:k + 1].astype(np.float64)
        if (srt[k - 1] - srt[k]) > 1e-4 * srt[k - 1]:      # guarded boundary: the two orders agree on the set
            assert set(res["fused"][0][b].tolist()) == set(res["unfused"][0][b].tolist())
