#!/usr/bin/env python3
"""Generate acquisition golden vectors by IMPORTING the reference (authoring container only).

Runs only where /root/reference exists.  Writes small .npz fixtures into tests/golden/.
The fixtures are data (inputs + the reference's outputs); no reference source is copied.

Reference call sites exercised:
  query.py:229-239   UncertaintySampler._entropy/_least_confidence/_margin_sampling   (G1)
  query.py:33-69     QuerySelector._select_queries (k=n_pixels_by_us and top_n_percent modes) (G2)
  query.py:230       0*log(0)=NaN behaviour, k > #unmasked behaviour                  (G3)
  query.py:144-221   QuerySelector.__call__ with a fake dataloader + 1x1-conv model   (G4)
  query.py:71-142    encode_query / decode_queries                                    (G5)
  query.py:320-351   merge_previous_query_files                                       (G5)
  deeplab.py:55-56 + query.py:190   low-res logits -> interpolate -> crop -> score  (--lowres, §8f-1)
  networks/model.py:6-14 + decoders.py:57-77 + query.py:144-221   FPNSeg-ResNet50 acquisition round (--fpn)
"""
import os
import sys
import pickle as pkl
import tempfile
from argparse import Namespace

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
sys.path.insert(0, REF)
import query as refq  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

STRATS = ["entropy", "least_confidence", "margin_sampling"]
FILL = {"entropy": 0.0, "least_confidence": 0.0, "margin_sampling": 1.0}


def mk_args(strategy, n_classes, k=20, top_n_percent=0.0, reverse_order=False, dataset_name="cs",
            ignore_index=None, dir_root="/tmp"):
    return Namespace(dataset_name=dataset_name, debug=False, dir_root=dir_root, experim_name="golden",
                     ignore_index=n_classes if ignore_index is None else ignore_index, mc_n_steps=20,
                     n_classes=n_classes, n_pixels_by_us=k, network_name="deeplab",
                     query_strategy=strategy, reverse_order=reverse_order, stride_total=8,
                     top_n_percent=top_n_percent, use_mc_dropout=False, vote_type="hard")


def gap_ok(vals_sorted_desc_or_asc, k, rel=1e-4):
    """k-th vs (k+1)-th gap guard."""
    a, b = float(vals_sorted_desc_or_asc[k - 1]), float(vals_sorted_desc_or_asc[k])
    return abs(a - b) >= rel * max(abs(a), abs(b), 1e-30)


def all_gaps_ok(vals, rel=1e-4):
    v = np.asarray(vals, dtype=np.float64)
    d = np.abs(np.diff(v))
    return bool(np.all(d >= rel * np.maximum(np.abs(v[1:]), np.abs(v[:-1]))))


# ----------------------------------------------------------------------------- G1 + G2 (k=20)
def gen_scores_and_topk():
    shapes = [(1, 19, 32, 64), (1, 11, 45, 60), (2, 21, 37, 53)]
    out = {}
    for si, (b, c, h, w) in enumerate(shapes):
        seed = 100 + si
        while True:
            torch.manual_seed(seed)
            logits = torch.randn(b, c, h, w) * 3
            rng = np.random.RandomState(seed)
            excl = np.zeros((b, h, w), dtype=np.uint8)
            for i in range(b):
                idx = rng.choice(h * w, 40, replace=False)
                excl[i].reshape(-1)[idx] = 1
                excl[i][rng.rand(h, w) < 0.05] = 1
            prob = F.softmax(logits, dim=1)
            ok = True
            rec = {}
            for st in STRATS:
                sampler = refq.UncertaintySampler(st)
                uc = sampler(prob)  # b,h,w
                rec[f"map_{st}"] = uc.numpy().copy()
                largest = st in ["entropy", "least_confidence"]
                sets, orders = [], []
                for i in range(b):
                    args = mk_args(st, c, k=20)
                    qs = refq.QuerySelector(args, dataloader=None, device=torch.device("cpu"))
                    m = uc[i].clone()
                    m[torch.from_numpy(excl[i].astype(bool))] = FILL[st]
                    # gap guard on this image
                    srt = torch.sort(m.flatten(), descending=largest).values.numpy()
                    if not gap_ok(srt, 20) or not all_gaps_ok(srt[:21]):
                        ok = False
                    qmask = qs._select_queries(m.clone())
                    sets.append(np.flatnonzero(qmask.reshape(-1)).astype(np.int64))
                    orders.append(m.flatten().topk(20, largest=largest).indices.numpy().astype(np.int64))
                rec[f"sel_{st}"] = np.stack(sets)
                rec[f"order_{st}"] = np.stack(orders)
            if ok:
                break
            seed += 1000
        out[f"s{si}_logits"] = logits.numpy()
        out[f"s{si}_exclude"] = excl
        out[f"s{si}_seed"] = np.int64(seed)
        for kk, v in rec.items():
            out[f"s{si}_{kk}"] = v
    np.savez_compressed(os.path.join(OUT, "acq_scores_topk.npz"), **out)
    print("G1/G2 written", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


# ----------------------------------------------------------------------------- G2 top-p% + subsample, reverse_order
def gen_select_modes():
    """_select_queries on a GIVEN uc_map (query.py:33-69): top_n_percent mode depends on the
    value-sorted order of topk + numpy RNG; reverse_order mode on numpy RNG only."""
    out = {}
    h, w = 48, 80
    for st in STRATS:
        largest = st in ["entropy", "least_confidence"]
        torch.manual_seed(7)
        # distinct values everywhere (a random permutation scaled) -> order is implementation independent
        vals = torch.randperm(h * w).float() / (h * w)
        uc = vals.reshape(h, w).clone()
        out[f"{st}_uc"] = uc.numpy().copy()
        # top 5% then subsample 10 with seed
        args = mk_args(st, 19, k=10, top_n_percent=0.05)
        qs = refq.QuerySelector(args, None, device=torch.device("cpu"))
        np.random.seed(1234)
        q = qs._select_queries(uc.clone())
        out[f"{st}_top5_sel"] = np.flatnonzero(q.reshape(-1)).astype(np.int64)
        kk = int(h * w * 0.05)
        out[f"{st}_top5_order"] = uc.flatten().topk(kk, largest=largest).indices.numpy().astype(np.int64)
        # reverse order: random 5% candidates then top-n
        args = mk_args(st, 19, k=10, top_n_percent=0.05, reverse_order=True)
        qs = refq.QuerySelector(args, None, device=torch.device("cpu"))
        np.random.seed(4321)
        q = qs._select_queries(uc.clone())
        out[f"{st}_rev_sel"] = np.flatnonzero(q.reshape(-1)).astype(np.int64)
    out["np_seed_top5"] = np.int64(1234)
    out["np_seed_rev"] = np.int64(4321)
    np.savez_compressed(os.path.join(OUT, "acq_select_modes.npz"), **out)
    print("G2 modes written")


# ----------------------------------------------------------------------------- G3 edge cases
def gen_edges():
    out = {}
    # (a) underflow -> NaN entropy pixel (query.py:230): logit gap > ~104 makes p underflow to 0
    torch.manual_seed(3)
    logits = torch.randn(1, 5, 8, 8) * 2
    logits[0, :, 2, 3] = torch.tensor([200.0, 0.0, -5.0, 1.0, 2.0])
    logits[0, :, 6, 1] = torch.tensor([0.0, 150.0, -5.0, 1.0, 2.0])
    prob = F.softmax(logits, dim=1)
    ent = refq.UncertaintySampler("entropy")(prob)[0]
    out["nan_logits"] = logits.numpy()
    out["nan_entropy_map"] = ent.numpy().copy()
    assert torch.isnan(ent[2, 3]) and torch.isnan(ent[6, 1])
    top = ent.flatten().topk(4, largest=True)
    out["nan_top4_idx"] = top.indices.numpy().astype(np.int64)   # NaNs first (torch semantic)
    out["nan_top4_isnan"] = torch.isnan(top.values).numpy()
    # LC and margin on the same logits (finite)
    out["nan_lc_map"] = refq.UncertaintySampler("least_confidence")(prob)[0].numpy().copy()
    out["nan_margin_map"] = refq.UncertaintySampler("margin_sampling")(prob)[0].numpy().copy()

    # (b) k == unmasked count / k > unmasked count: reference returns excluded pixels too.
    torch.manual_seed(5)
    h, w = 6, 7
    uc = torch.rand(h, w) + 0.5       # all > 0 so that excluded (0.0) sort last for entropy
    excl = np.ones((h, w), dtype=bool)
    free = [3, 11, 17, 25, 40]
    excl.reshape(-1)[free] = False
    m = uc.clone()
    m[torch.from_numpy(excl)] = 0.0
    out["few_uc"] = m.numpy().copy()
    out["few_exclude"] = excl.astype(np.uint8)
    args = mk_args("entropy", 19, k=5)
    qs = refq.QuerySelector(args, None, device=torch.device("cpu"))
    out["few_sel_k5"] = np.flatnonzero(qs._select_queries(m.clone()).reshape(-1)).astype(np.int64)
    # k=8 > 5 free: the 3 extra come from the excluded (all 0.0, tie order implementation-defined in
    # torch) -> only the count and the superset property are pinned.
    args = mk_args("entropy", 19, k=8)
    qs = refq.QuerySelector(args, None, device=torch.device("cpu"))
    q8 = qs._select_queries(m.clone())
    out["few_sel_k8_count"] = np.int64(q8.sum())
    out["few_sel_k8_contains_free"] = np.bool_(all(q8.reshape(-1)[free]))
    np.savez_compressed(os.path.join(OUT, "acq_edges.npz"), **out)
    print("G3 written")


# ----------------------------------------------------------------------------- G4 end to end
class FakeDataset:
    def __init__(self, xs, ys, queries, names):
        self.xs, self.ys, self.queries, self.names = xs, ys, queries, names
        self.labelled = None

    def label_queries(self, dict_queries, nth):
        self.labelled = (dict_queries, nth)


class FakeLoader:
    def __init__(self, ds):
        self.dataset = ds

    def __iter__(self):
        for i in range(len(self.dataset.xs)):
            yield {"x": self.dataset.xs[i][None], "y": self.dataset.ys[i][None], "p_img": [self.dataset.names[i]]}

    def __len__(self):
        return len(self.dataset.xs)


class OneConv(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b = w, b

    def forward(self, x):
        return {"pred": F.conv2d(x, self.w, self.b)}


def gen_end_to_end():
    out = {}
    C, h, w, n_img = 19, 40, 56, 3
    for st in STRATS:
        seed = 50
        while True:
            torch.manual_seed(seed)
            rng = np.random.RandomState(seed)
            W = torch.randn(C, 3, 1, 1) * 2.0
            Bv = torch.randn(C) * 0.5
            xs = [torch.randn(3, h, w) * 1.5 for _ in range(n_img)]
            ys = [torch.from_numpy(rng.randint(0, C + 1, size=(h, w)).astype(np.int64)) for _ in range(n_img)]
            for y in ys:   # ~6% void
                y[torch.from_numpy(rng.rand(h, w) < 0.06)] = C
            prev = []
            for _ in range(n_img):
                q = np.zeros((h, w), dtype=bool)
                q.reshape(-1)[rng.choice(h * w, 30, replace=False)] = True
                prev.append(q)
            names = [f"/data/img_{i:03d}.png" for i in range(n_img)]
            model = OneConv(W, Bv)
            # gap guard
            largest = st in ["entropy", "least_confidence"]
            ok = True
            with torch.no_grad():
                for i in range(n_img):
                    prob = F.softmax(model(xs[i][None])["pred"], dim=1)
                    uc = refq.UncertaintySampler(st)(prob)[0]
                    uc[torch.from_numpy(prev[i])] = FILL[st]
                    uc[ys[i] == C] = FILL[st]
                    srt = torch.sort(uc.flatten(), descending=largest).values.numpy()
                    if not all_gaps_ok(srt[:21]):
                        ok = False
            if ok:
                break
            seed += 1
        ds = FakeDataset(xs, ys, prev, names)
        with tempfile.TemporaryDirectory() as td:
            args = mk_args(st, C, k=20, dir_root=td)
            qs = refq.QuerySelector(args, FakeLoader(ds), device=torch.device("cpu"))
            dq = qs(nth_query=1, model=model)
            stats = pkl.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
        out[f"{st}_W"] = W.numpy()
        out[f"{st}_b"] = Bv.numpy()
        out[f"{st}_xs"] = torch.stack(xs).numpy()
        out[f"{st}_ys"] = torch.stack(ys).numpy()
        out[f"{st}_prev"] = np.stack(prev)
        for i, nme in enumerate(names):
            out[f"{st}_x_{i}"] = np.asarray(dq[nme]["x_coords"], dtype=np.int64)
            out[f"{st}_y_{i}"] = np.asarray(dq[nme]["y_coords"], dtype=np.int64)
        out[f"{st}_stats_label_cnt"] = np.array([stats["label_distribution"][l] for l in range(C)], dtype=np.int64)
        out[f"{st}_stats_avg_entropy"] = np.float64(stats["avg_entropy"])
        out[f"{st}_stats_avg_n_unique"] = np.float64(stats["avg_n_unique_labels"])
        out[f"{st}_stats_avg_cov"] = np.float64(stats["avg_spatial_coverage"])
        assert ds.labelled is not None and ds.labelled[1] == 1
    out["names"] = np.array([f"/data/img_{i:03d}.png" for i in range(n_img)])
    np.savez_compressed(os.path.join(OUT, "acq_end_to_end.npz"), **out)
    print("G4 written")


# ----------------------------------------------------------------------------- G5 codec
def gen_codec():
    out = {}
    rng = np.random.RandomState(9)
    h, w = 12, 17
    q = rng.rand(h, w) < 0.1
    enc = refq.QuerySelector.encode_query("a/b.png", (h, w), q)
    out["mask"] = q
    out["enc_x"] = enc["a/b.png"]["x_coords"].astype(np.int64)
    out["enc_y"] = enc["a/b.png"]["y_coords"].astype(np.int64)
    dec = refq.QuerySelector.decode_queries(enc)
    assert len(dec) == 1 and (dec[0] == q).all()
    out["dec_single"] = dec[0]
    # with category ids -> int64 label map with ignore_index fill
    enc2 = {"z.png": dict(enc["a/b.png"]), "a.png": dict(enc["a/b.png"])}
    cat = rng.randint(0, 11, size=int(q.sum()))
    enc2["z.png"]["category_id"] = cat.tolist()
    d2 = refq.QuerySelector.decode_queries(enc2, ignore_index=11, return_as_dict=True)
    out["cat"] = cat.astype(np.int64)
    out["dec_cat_z"] = d2["z.png"]
    out["dec_plain_a"] = d2["a.png"]
    l2 = refq.QuerySelector.decode_queries(enc2, ignore_index=11)
    out["dec_list_order_first_is_a"] = np.bool_(l2[0].dtype == np.bool_)   # sorted by key: a.png first
    # merge_previous_query_files
    with tempfile.TemporaryDirectory() as td:
        files = []
        maps = []
        for r in range(3):
            os.makedirs(f"{td}/{r}_query")
            qq = rng.rand(h, w) < 0.05
            yy, xx = np.where(qq)
            e = {"img0.png": {"height": h, "width": w, "x_coords": xx, "y_coords": yy,
                              "category_id": rng.randint(0, 11, size=len(xx)).tolist()}}
            if r != 1:
                qq2 = rng.rand(h, w) < 0.05
                yy2, xx2 = np.where(qq2)
                e["img1.png"] = {"height": h, "width": w, "x_coords": xx2, "y_coords": yy2,
                                 "category_id": rng.randint(0, 11, size=len(xx2)).tolist()}
            pkl.dump(e, open(f"{td}/{r}_query/queries.pkl", "wb"))
            files.append(f"{td}/{r}_query/queries.pkl")
            maps.append(e)
        found = sorted(refq.gather_previous_query_files(td))
        assert found == sorted(files)
        merged = refq.merge_previous_query_files(files, ignore_index=11, verbose=False)
        out["merge_img0"] = merged["img0.png"]
        out["merge_img1"] = merged["img1.png"]
        for r, e in enumerate(maps):
            for nme, info in e.items():
                tag = nme.split(".")[0]
                out[f"merge_in_{r}_{tag}_x"] = np.asarray(info["x_coords"], dtype=np.int64)
                out[f"merge_in_{r}_{tag}_y"] = np.asarray(info["y_coords"], dtype=np.int64)
                out[f"merge_in_{r}_{tag}_c"] = np.asarray(info["category_id"], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "acq_codec.npz"), **out)
    print("G5 written")


class Conv3(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.w, self.b = w, b

    def forward(self, x):
        return {"pred": F.conv2d(x, self.w, self.b, padding=1)}


class FakeLoaderNoY(FakeLoader):
    def __iter__(self):
        for i in range(len(self.dataset.xs)):
            yield {"x": self.dataset.xs[i][None], "p_img": [self.dataset.names[i]]}


def gen_branches():
    """query.py:171-174,190 (voc: reflect-pad to a multiple of stride_total, crop the logits) and
    query.py:145-146,196-197 (human_labels: previous labels are int64 maps, excluded where != ignore_index, no 'y')."""
    out = {}
    C, h, w, n_img = 21, 37, 53, 2
    seed = 77
    while True:
        torch.manual_seed(seed)
        rng = np.random.RandomState(seed)
        W3 = torch.randn(C, 3, 3, 3) * 0.8
        b3 = torch.randn(C) * 0.3
        xs = [torch.randn(3, h, w) * 1.5 for _ in range(n_img)]
        ys = [torch.from_numpy(rng.randint(0, C, size=(h, w)).astype(np.int64)) for _ in range(n_img)]
        for y in ys:
            y[torch.from_numpy(rng.rand(h, w) < 0.05)] = 255
        prev = []
        for _ in range(n_img):
            q = np.zeros((h, w), dtype=bool)
            q.reshape(-1)[rng.choice(h * w, 25, replace=False)] = True
            prev.append(q)
        names = [f"/voc/img_{i}.jpg" for i in range(n_img)]
        model = Conv3(W3, b3)
        ok = True
        with torch.no_grad():
            for i in range(n_img):
                xp = F.pad(xs[i][None], (0, 56 - w, 0, 40 - h), mode="reflect")
                prob = F.softmax(model(xp)["pred"][:, :, :h, :w], dim=1)
                uc = refq.UncertaintySampler("margin_sampling")(prob)[0]
                uc[torch.from_numpy(prev[i])] = 1.0
                uc[ys[i] == 255] = 1.0
                if not all_gaps_ok(torch.sort(uc.flatten()).values.numpy()[:11]):
                    ok = False
        if ok:
            break
        seed += 1
    ds = FakeDataset(xs, ys, prev, names)
    with tempfile.TemporaryDirectory() as td:
        args = mk_args("margin_sampling", C, k=10, dataset_name="voc", ignore_index=255, dir_root=td)
        dq = refq.QuerySelector(args, FakeLoader(ds), device=torch.device("cpu"))(nth_query=1, model=model)
    out["voc_W"], out["voc_b"] = W3.numpy(), b3.numpy()
    out["voc_xs"], out["voc_ys"], out["voc_prev"] = torch.stack(xs).numpy(), torch.stack(ys).numpy(), np.stack(prev)
    for i, nme in enumerate(names):
        out[f"voc_x_{i}"], out[f"voc_y_{i}"] = np.asarray(dq[nme]["x_coords"]), np.asarray(dq[nme]["y_coords"])
    # human labels
    C, h, w = 11, 24, 40
    seed = 91
    while True:
        torch.manual_seed(seed)
        rng = np.random.RandomState(seed)
        W1 = torch.randn(C, 3, 1, 1) * 2.0
        b1 = torch.randn(C) * 0.5
        xs = [torch.randn(3, h, w) * 1.5 for _ in range(n_img)]
        labelled = []
        for _ in range(n_img):
            m = np.full((h, w), 11, dtype=np.int64)
            idx = rng.choice(h * w, 30, replace=False)
            m.reshape(-1)[idx] = rng.randint(0, C, size=30)
            labelled.append(m)
        ok = True
        with torch.no_grad():
            for i in range(n_img):
                prob = F.softmax(OneConv(W1, b1)(xs[i][None])["pred"], dim=1)
                uc = refq.UncertaintySampler("least_confidence")(prob)[0]
                uc[torch.from_numpy(labelled[i] != 11)] = 0.0
                if not all_gaps_ok(torch.sort(uc.flatten(), descending=True).values.numpy()[:13]):
                    ok = False
        if ok:
            break
        seed += 1
    names = [f"/cv/img_{i}.png" for i in range(n_img)]
    ds = FakeDataset(xs, [None] * n_img, None, names)
    ds.list_labelled_queries = labelled
    with tempfile.TemporaryDirectory() as td:
        args = mk_args("least_confidence", C, k=12, dataset_name="cv", ignore_index=11, dir_root=td)
        dq = refq.QuerySelector(args, FakeLoaderNoY(ds), device=torch.device("cpu"))(nth_query=2, model=OneConv(W1, b1), human_labels=True)
    out["hl_W"], out["hl_b"] = W1.numpy(), b1.numpy()
    out["hl_xs"], out["hl_labelled"] = torch.stack(xs).numpy(), np.stack(labelled)
    for i, nme in enumerate(names):
        out[f"hl_x_{i}"], out[f"hl_y_{i}"] = np.asarray(dq[nme]["x_coords"]), np.asarray(dq[nme]["y_coords"])
    assert ds.labelled is None        # human_labels: label_queries is NOT called (query.py:215)
    np.savez_compressed(os.path.join(OUT, "acq_branches.npz"), **out)
    print("branches written")


def gen_random():
    """args.py:27 `random` strategy through QuerySelector.__call__ (query.py:190-204,242-247): the score map is a CPU
    torch.rand((1,h,w)) per image, excluded pixels are filled with 1.0 and the k smallest win.  Three modes: k = 20,
    top-5 % + numpy sub-sample, reverse order.  torch / numpy seeds are part of the fixture."""
    out = {}
    C, h, w, n_img = 19, 40, 56, 3
    for mode, kw in (("k20", dict(k=20)), ("top5", dict(k=10, top_n_percent=0.05)),
                     ("rev", dict(k=10, top_n_percent=0.05, reverse_order=True))):
        seed = 300
        while True:
            torch.manual_seed(seed)
            rng = np.random.RandomState(seed)
            W = torch.randn(C, 3, 1, 1) * 2.0
            Bv = torch.randn(C) * 0.5
            xs = [torch.randn(3, h, w) * 1.5 for _ in range(n_img)]
            ys = [torch.from_numpy(rng.randint(0, C + 1, size=(h, w)).astype(np.int64)) for _ in range(n_img)]
            prev = []
            for _ in range(n_img):
                q = np.zeros((h, w), dtype=bool)
                q.reshape(-1)[rng.choice(h * w, 30, replace=False)] = True
                prev.append(q)
            # gap guard on the maps the selector is going to draw (no other consumer of the torch CPU RNG in between)
            torch.manual_seed(seed + 1000)
            np.random.seed(seed + 2000)
            ok = True
            for i in range(n_img):
                uc = torch.rand((1, h, w))[0]
                uc[torch.from_numpy(prev[i])] = 1.0
                uc[ys[i] == C] = 1.0
                uc = uc.flatten()
                if mode == "rev":
                    k5 = int(h * w * 0.05)
                    cand = np.random.choice(range(h * w), k5, False)
                    sm = np.zeros(h * w, dtype=bool)
                    sm[cand] = True
                    uc[torch.from_numpy(~sm)] = 1.0
                    n_chk = kw["k"] + 1
                else:
                    n_chk = (int(h * w * 0.05) if mode == "top5" else kw["k"]) + 1
                srt = torch.sort(uc).values.numpy()
                if not all_gaps_ok(srt[:n_chk]) or srt[n_chk - 1] >= 1.0:
                    ok = False
            if ok:
                break
            seed += 1
        names = [f"/data/rnd_{i:03d}.png" for i in range(n_img)]
        ds = FakeDataset(xs, ys, prev, names)
        with tempfile.TemporaryDirectory() as td:
            args = mk_args("random", C, dir_root=td, **kw)
            qs = refq.QuerySelector(args, FakeLoader(ds), device=torch.device("cpu"))
            torch.manual_seed(seed + 1000)
            np.random.seed(seed + 2000)
            dq = qs(nth_query=1, model=OneConv(W, Bv))
            stats = pkl.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
        out[f"{mode}_seed"] = np.int64(seed)
        out[f"{mode}_W"], out[f"{mode}_b"] = W.numpy(), Bv.numpy()
        out[f"{mode}_xs"], out[f"{mode}_ys"], out[f"{mode}_prev"] = torch.stack(xs).numpy(), torch.stack(ys).numpy(), np.stack(prev)
        for i, nme in enumerate(names):
            out[f"{mode}_x_{i}"] = np.asarray(dq[nme]["x_coords"], dtype=np.int64)
            out[f"{mode}_y_{i}"] = np.asarray(dq[nme]["y_coords"], dtype=np.int64)
            assert len(dq[nme]["x_coords"]) == kw["k"]
        out[f"{mode}_stats_label_cnt"] = np.array([stats["label_distribution"][l] for l in range(C)], dtype=np.int64)
        out[f"{mode}_stats_avg_entropy"] = np.float64(stats["avg_entropy"])
        out[f"{mode}_stats_avg_cov"] = np.float64(stats["avg_spatial_coverage"])
        print("random", mode, "seed", seed)
    np.savez_compressed(os.path.join(OUT, "acq_random.npz"), **out)
    print("random written")


def gen_lowres():
    """SURVEY.md §8f-1 fixture: low-resolution classifier logits -> deeplab.py:55-56 F.interpolate(bilinear,
    align_corners=True) -> [:h,:w] crop (query.py:190) -> reference sampler + _select_queries."""
    cases = [((2, 19, 8, 16), (32, 64), None), ((1, 11, 12, 15), (45, 60), None), ((2, 21, 5, 7), (40, 56), (37, 53))]
    out = {}
    for si, ((b, c, hl, wl), size, crop) in enumerate(cases):
        seed = 700 + si
        hc, wc = size if crop is None else crop
        while True:
            torch.manual_seed(seed)
            low = torch.randn(b, c, hl, wl) * 3
            rng = np.random.RandomState(seed)
            excl = np.zeros((b, hc, wc), dtype=np.uint8)
            for i in range(b):
                excl[i].reshape(-1)[rng.choice(hc * wc, 40, replace=False)] = 1
                excl[i][rng.rand(hc, wc) < 0.05] = 1
            pred = F.interpolate(low, size=size, mode="bilinear", align_corners=True)[:, :, :hc, :wc]
            prob = F.softmax(pred, dim=1)
            ok, rec = True, {}
            for st in STRATS:
                uc = refq.UncertaintySampler(st)(prob)
                rec[f"map_{st}"] = uc.numpy().copy()
                largest = st in ["entropy", "least_confidence"]
                sets, orders = [], []
                for i in range(b):
                    qs = refq.QuerySelector(mk_args(st, c, k=20), dataloader=None, device=torch.device("cpu"))
                    m = uc[i].clone()
                    m[torch.from_numpy(excl[i].astype(bool))] = FILL[st]
                    srt = torch.sort(m.flatten(), descending=largest).values.numpy()
                    if not gap_ok(srt, 20) or not all_gaps_ok(srt[:21]):
                        ok = False
                    sets.append(np.flatnonzero(qs._select_queries(m.clone()).reshape(-1)).astype(np.int64))
                    orders.append(m.flatten().topk(20, largest=largest).indices.numpy().astype(np.int64))
                rec[f"sel_{st}"] = np.stack(sets)
                rec[f"order_{st}"] = np.stack(orders)
            if ok:
                break
            seed += 1000
        out[f"s{si}_low"] = low.numpy()
        out[f"s{si}_size"] = np.array(size, dtype=np.int64)
        out[f"s{si}_crop"] = np.array([hc, wc], dtype=np.int64)
        out[f"s{si}_exclude"] = excl
        if si == 1:          # the interpolated logits themselves, one case (pins the bilinear restatement)
            out[f"s{si}_pred"] = pred.numpy()
        for kk, v in rec.items():
            out[f"s{si}_{kk}"] = v
    np.savez_compressed(os.path.join(OUT, "acq_lowres.npz"), **out)
    print("lowres written", os.path.getsize(os.path.join(OUT, "acq_lowres.npz")))


def gen_fpn_round():
    """BASELINE configs[2-4] network leg: the reference's FPNSeg-ResNet50 (networks/model.py:6-14, decoders.py:57-77: four
    UpsampleBlock outputs summed at full resolution, 128 channels, then the 1x1 classifier) driven by the reference's
    QuerySelector.__call__ (query.py:144-221) on a fake loader.  Weights and images are formula-initialised
    (tests/formula_init.py), so the fixture holds only labels, previous queries and what the reference picked.  `dict_queries`
    holds the picks in row-major order (np.where), so only the SET is pinned: keys are searched until the 20th and 21st
    score of every image are >= 2e-3 (entropy / least confidence) or 2e-2 (margin: scores near 0) apart, relative - the build's
    half-resolution order (classifier in front of the last x2 interpolation) differs from the reference's by fp32 rounding
    only."""
    import contextlib, io
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import formula_init as fi
    from utils.utils import get_model
    out = {}
    cases = [("cs_entropy", "cs", 19, 19, "entropy", [(64, 96)] * 3),
             ("cs_least_confidence", "cs", 19, 19, "least_confidence", [(64, 96)] * 2),
             ("voc_margin", "voc", 21, 255, "margin_sampling", [(43, 61), (50, 37)])]
    for tag, ds_name, C, ign, st, sizes in cases:
        margs = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name="FPN", weight_type="random",
                          use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)
        with contextlib.redirect_stdout(io.StringIO()):
            model = get_model(margs)
        model.load_state_dict(fi.formula_state_dict(model.state_dict()))
        model.eval()
        largest = st in ["entropy", "least_confidence"]
        guard = 2e-3 if largest else 2e-2
        attempt = 0
        while True:
            rng = np.random.RandomState(900 + attempt)
            keys = [f"fpnacq_{tag}_{attempt}_{i}" for i in range(len(sizes))]
            xs = [fi.formula_input(1, h, w, key=k)[0] for k, (h, w) in zip(keys, sizes)]
            ys, prev = [], []
            for (h, w) in sizes:
                y = rng.randint(0, C, size=(h, w)).astype(np.int64)
                y[rng.rand(h, w) < 0.06] = ign
                ys.append(torch.from_numpy(y))
                q = np.zeros((h, w), dtype=bool)
                q.reshape(-1)[rng.choice(h * w, 30, replace=False)] = True
                prev.append(q)
            ok, gaps = True, []
            with torch.no_grad():
                for i, (h, w) in enumerate(sizes):
                    x = xs[i][None]
                    if ds_name == "voc":
                        x = F.pad(x, pad=(0, -w % 8, 0, -h % 8), mode="reflect")
                    prob = F.softmax(model(x)["pred"][:, :, :h, :w], dim=1)
                    uc = refq.UncertaintySampler(st)(prob)[0]
                    uc[torch.from_numpy(prev[i])] = FILL[st]
                    uc[ys[i] == ign] = FILL[st]
                    srt = torch.sort(uc.flatten(), descending=largest).values.numpy()
                    a, b = float(srt[19]), float(srt[20])
                    gaps.append(abs(a - b) / max(abs(a), abs(b), 1e-30))
                    ok = ok and gaps[-1] >= guard
            print(tag, "attempt", attempt, "min rel gap at k", min(gaps), "ok" if ok else "retry", flush=True)
            if ok:
                break
            attempt += 1
        names = [f"/data/{tag}_{i:03d}.png" for i in range(len(sizes))]
        ds = FakeDataset(xs, ys, prev, names)
        with tempfile.TemporaryDirectory() as td:
            args = mk_args(st, C, k=20, dir_root=td, dataset_name=ds_name, ignore_index=ign)
            args.network_name = "FPN"
            qs = refq.QuerySelector(args, FakeLoader(ds), device=torch.device("cpu"))
            dq = qs(nth_query=1, model=model)
            stats = pkl.load(open(f"{td}/checkpoints/golden/1_query/query_stats.pkl", "rb"))
        out[f"{tag}_keys"] = np.array(keys)
        out[f"{tag}_names"] = np.array(names)
        out[f"{tag}_sizes"] = np.array(sizes, dtype=np.int64)
        out[f"{tag}_meta"] = np.array([C, ign], dtype=np.int64)
        out[f"{tag}_min_rel_gap"] = np.float64(min(gaps))
        for i, nme in enumerate(names):
            out[f"{tag}_y_{i}"] = ys[i].numpy().astype(np.uint8 if ign < 256 else np.int64)
            out[f"{tag}_prev_{i}"] = np.packbits(prev[i].reshape(-1))
            out[f"{tag}_xc_{i}"] = np.asarray(dq[nme]["x_coords"], dtype=np.int64)
            out[f"{tag}_yc_{i}"] = np.asarray(dq[nme]["y_coords"], dtype=np.int64)
        out[f"{tag}_stats_label_cnt"] = np.array([stats["label_distribution"][l] for l in range(C)], dtype=np.int64)
        out[f"{tag}_stats_avg_entropy"] = np.float64(stats["avg_entropy"])
        out[f"{tag}_stats_avg_n_unique"] = np.float64(stats["avg_n_unique_labels"])
        out[f"{tag}_stats_avg_cov"] = np.float64(stats["avg_spatial_coverage"])
    np.savez_compressed(os.path.join(OUT, "acq_fpn_round.npz"), **out)
    print("fpn round written", os.path.getsize(os.path.join(OUT, "acq_fpn_round.npz")))


if __name__ == "__main__":
    if "--fpn" in sys.argv:
        gen_fpn_round()
        sys.exit(0)
    if "--lowres" in sys.argv:
        gen_lowres()
        sys.exit(0)
    if "--random" in sys.argv:
        gen_random()
        sys.exit(0)
    if "--branches" in sys.argv:
        gen_branches()
        sys.exit(0)
    gen_scores_and_topk()
    gen_select_modes()
    gen_edges()
    gen_end_to_end()
    gen_codec()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden dir bytes:", tot)
