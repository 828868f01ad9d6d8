"""GPU: device-side metrics vs the numpy restatement of RunningScore, and the active-learning driver end to end
on a synthetic dataset (model.py:53-86 control flow: train -> query -> label, twice)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from pixelpick_amd.model import Model
from pixelpick_amd.synthetic import SyntheticDataset
from pixelpick_amd.utils.metrics import RunningScore

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_confusion_matrix_matches_numpy_bincount():
    torch.manual_seed(0)
    B, C, H, W = 3, 19, 37, 53
    logits = torch.randn(B, C, H, W, device=DEV)
    y = torch.randint(0, C + 1, (B, H, W), device=DEV)        # C == ignore_index
    y[0, :5] = 255
    rs = RunningScore(C)
    rs.update_from_logits(y, logits)
    rs.update_from_logits(y, logits)
    got = rs.get_scores()
    ref = RunningScore(C)
    pred = logits.argmax(dim=1).cpu().numpy()
    for _ in range(2):
        ref.update(y.cpu().numpy(), pred)
    np.testing.assert_array_equal(rs.confusion_matrix, ref.confusion_matrix)
    exp = ref.get_scores()
    for k in exp[0]:
        assert np.isclose(got[0][k], exp[0][k], equal_nan=True)


def _args(td, **kw):
    base = dict(dataset_name="cs", debug=False, dir_root=td, experim_name="synthetic", ignore_index=5, mc_n_steps=20,
                n_classes=5, n_pixels_by_us=10, network_name="deeplab", query_strategy="margin_sampling", reverse_order=False,
                stride_total=16, top_n_percent=0.0, use_mc_dropout=False, vote_type="hard", mc_dropout_p=0.2,
                n_init_pixels=10, max_budget=20, n_epochs=2, lr_scheduler_type="Poly",
                optimizer_params={"lr": 5e-4, "betas": (0.9, 0.999), "weight_decay": 2e-4, "eps": 1e-7})
    base.update(kw)
    return Namespace(**base)


def test_active_learning_rounds_on_synthetic_data(tmp_path):
    import warnings
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = SyntheticDataset(8, 64, 96, 5, 5, n_init_pixels=10, seed=1)
    ds_val = SyntheticDataset(4, 64, 96, 5, 5, seed=2)
    mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
    args = _args(str(tmp_path))
    m = Model(args, mk(ds, 4, True), mk(ds, 1, False), mk(ds_val, 1, False), device=torch.device(DEV))
    before = [q.copy() for q in ds.queries]
    m()
    # 3 stages (1 initial + max_budget/n_pixels_by_us = 2): each adds exactly 10 new, previously unlabelled, non-void pixels per image
    assert ds.labelled_rounds == [0, 1, 1, 2, 2, 3][:len(ds.labelled_rounds)] or len(ds.labelled_rounds) >= 3
    for i in range(len(ds)):
        assert ds.queries[i].sum() == 10 + 3 * 10
        assert (ds.queries[i] & before[i]).sum() == 10
        assert not ((ds.queries[i] & ~before[i]) & (ds.ys[i].numpy() == 5)).any()      # new picks never hit void
    for nth in range(3):
        d = tmp_path / "checkpoints" / "synthetic" / f"{nth}_query"
        assert (d / "log_train.txt").exists() and (d / "best_miou_model.pt").exists() and (d / "query_stats.pkl").exists()
    losses = [h[5] for h in m.history if h[0] == "train"]
    assert all(np.isfinite(l) for l in losses)
    sd = torch.load(tmp_path / "checkpoints" / "synthetic" / "2_query" / "best_miou_model.pt")["model"]
    assert len(sd) == 668


def test_validation_batched_equals_per_image(tmp_path):
    """Model._val forwards equal-sized images val_batch_size at a time; eval-mode results per image do not depend on
    the batch, so the confusion matrix (and mIoU) match the reference's one-image-per-forward loop."""
    import warnings
    from pixelpick_amd.utils.utils import get_model
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    ds = SyntheticDataset(4, 64, 96, 5, 5, n_init_pixels=10, seed=1)
    ds_val = SyntheticDataset(11, 64, 96, 5, 5, seed=2)
    mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
    res = []
    model = None
    for vbs in (1, 4, 8):
        args = _args(str(tmp_path / f"v{vbs}"), val_batch_size=vbs)
        m = Model(args, mk(ds, 4, True), mk(ds, 1, False), mk(ds_val, 1, False), device=torch.device(DEV))
        m.nth_query = 0
        os.makedirs(f"{m.dir_checkpoints}/0_query", exist_ok=True)
        m._open_logs(f"{m.dir_checkpoints}/0_query")
        if model is None:
            model = get_model(args).to(DEV)
        cm = []
        orig = m.running_score.get_scores

        def spy():
            out = orig()                                   # pulls the device-side matrix to the host
            cm.append(np.array(m.running_score.confusion_matrix, copy=True))
            return out
        m.running_score.get_scores = spy
        m._val(1, model)
        res.append((cm[0], m.history[-1][3], m.history[-1][4]))
    for cmx, miou, acc in res[1:]:
        assert cmx.sum() == res[0][0].sum() == 11 * 64 * 96 - int(sum((y == 5).sum() for y in ds_val.ys))
        assert np.abs(cmx - res[0][0]).sum() <= 4          # a near-tie argmax may flip with the summation order
        assert abs(miou - res[0][1]) < 1e-3 and abs(acc - res[0][2]) < 1e-3


def test_active_learning_round_with_replayed_train_steps(tmp_path, monkeypatch):
    """args.replay_train_step: the driver records the first train step of every round's trainer and re-issues its launch list
    afterwards (FlatTrainer.enable_replay).  Same control flow and artefacts as the eager loop; a ragged last batch falls back
    to eager steps."""
    import warnings
    from pixelpick_amd.trainer import FlatTrainer
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = SyntheticDataset(10, 64, 96, 5, 5, n_init_pixels=10, seed=1)          # 10 images, batch 4: two full batches + a ragged one
    ds_val = SyntheticDataset(4, 64, 96, 5, 5, seed=2)
    mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
    args = _args(str(tmp_path), replay_train_step=True, max_budget=10)
    seen = {"replayed": 0, "recorded": 0, "dropped": 0}
    orig_enable, orig_disable = FlatTrainer.enable_replay, FlatTrainer.disable_replay

    def enable(self, *a, **k):
        seen["recorded"] += 1
        return orig_enable(self, *a, **k)

    def disable(self):
        seen["dropped"] += int(self._plan is not None)
        return orig_disable(self)

    monkeypatch.setattr(FlatTrainer, "enable_replay", enable)
    monkeypatch.setattr(FlatTrainer, "disable_replay", disable)
    m = Model(args, mk(ds, 4, True), mk(ds, 1, False), mk(ds_val, 1, False), device=torch.device(DEV))
    m()
    assert seen["recorded"] >= 2 and seen["dropped"] >= 1       # re-recorded after every ragged batch, in both rounds
    losses = [h[5] for h in m.history if h[0] == "train"]
    assert len(losses) == 4 and all(np.isfinite(l) for l in losses)
    for i in range(len(ds)):
        assert ds.queries[i].sum() == 10 + 2 * 10
    assert (tmp_path / "checkpoints" / "synthetic" / "1_query" / "best_miou_model.pt").exists()


def test_replayed_rounds_do_not_grow_device_memory(tmp_path):
    """Every active-learning round builds a new trainer with its own private memory pool for the recorded step; _train() drops the
    plan (FlatTrainer.disable_replay in a finally) so that the pool is released with the trainer: reserved device memory after
    round r + 1 is not above round r's (it grew by a full step of activations per round before the teardown existed)."""
    import gc
    import warnings
    warnings.simplefilter("ignore")
    torch.manual_seed(0)
    np.random.seed(0)
    ds = SyntheticDataset(8, 64, 96, 5, 5, n_init_pixels=10, seed=1)
    ds_val = SyntheticDataset(4, 64, 96, 5, 5, seed=2)
    mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
    args = _args(str(tmp_path), replay_train_step=True, max_budget=40)          # four rounds of 10 pixels per image
    m = Model(args, mk(ds, 4, True), mk(ds, 1, False), mk(ds_val, 1, False), device=torch.device(DEV))
    reserved = []
    orig = m._train

    def train_and_measure(*a, **k):
        out = orig(*a, **k)
        gc.collect()
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
        return out

    m._train = train_and_measure
    m()
    assert len(reserved) >= 3
    assert max(reserved[2:]) <= reserved[1] + (8 << 20), [r >> 20 for r in reserved]


def test_active_learning_round_on_the_device_data_path(tmp_path):
    """SURVEY.md 8f-4 wired end to end: the train loader yields RAW uint8 batches (RawSyntheticDataset), Model._train_epoch
    augments them on the GPU (DeviceAugmenter: random scale / pad / crop / flip of image + label map + query mask, colour
    jitter / grayscale / blur, to_tensor + normalize) and trains on the result; a full active-learning round runs on it.
    Every train batch is replayed on the host with the primitives the reference calls (oracle/augment.py: PIL resize / expand /
    crop / transpose, torch nearest for the query tensor - datasets/base_dataset.py:55-118) from the SAME draws: the sparse
    label map the step trained on (labels at the warped query pixels, ignore_index elsewhere - model.py:108-110) must be
    bit-identical, i.e. the warped query masks land on exactly the labelled pixels of the host pipeline; the image matches
    bit for bit when no blur was drawn (blur: cv2 restatement, +-1.5 grey levels)."""
    import warnings
    from oracle import augment as orc
    from pixelpick_amd.augment import DeviceAugmenter
    from pixelpick_amd.synthetic import RawSyntheticDataset
    warnings.simplefilter("ignore")
    import random
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    crop = (64, 96)
    ds = RawSyntheticDataset(8, 72, 104, 5, 5, MEAN, STD, n_init_pixels=12, seed=1, train=True)
    ds_q = ds.view(train=False)
    ds_val = RawSyntheticDataset(4, 64, 96, 5, 5, MEAN, STD, seed=2, train=False)
    mk = lambda d, b, sh: torch.utils.data.DataLoader(d, batch_size=b, shuffle=sh)
    aug = DeviceAugmenter(crop, MEAN, STD, ignore_index=5, device=DEV)
    args = _args(str(tmp_path), max_budget=10, n_init_pixels=12, n_epochs=2)
    m = Model(args, mk(ds, 4, True), mk(ds_q, 1, False), mk(ds_val, 1, False), device=torch.device(DEV), augmenter=aug)
    seen = []
    m.on_train_batch = lambda d, x, y, mask, params: seen.append(
        (list(d['p_img']), d['queries'].numpy().copy(), x.cpu(), y.cpu(), mask.cpu(), params))
    m()
    assert len(seen) == 2 * 2 * 2                                  # 2 stages x 2 epochs x 2 batches of 4
    name_to_i = {n: i for i, n in enumerate(ds.names)}
    n_lab = n_blur = n_exact = n_scaled = 0
    for names, q_in, x, y, mask, params in seen:
        assert x.shape == (4, 3) + crop and y.shape == (4,) + crop and y.dtype == torch.int64
        for b, (name, p) in enumerate(zip(names, params)):
            i = name_to_i[name]
            img, y_ref, q_ref = orc.geometric(ds.imgs[i], ds.labels[i], q_in[b].astype(np.uint8), p, crop, aug.mean_val, 5)
            assert np.array_equal(mask[b].numpy(), q_ref), (name, p)
            y_sparse = np.where(q_ref != 0, y_ref, 5)
            assert np.array_equal(y[b].numpy(), y_sparse), (name, p)      # bit-exact: same labelled pixels, same labels
            n_lab += int((y_sparse != 5).sum())
            n_scaled += (p["h_rs"], p["w_rs"]) != (72, 104)
            for op, f in p["ops"]:
                img = orc.jitter(img, op, f)
            arr = np.asarray(img)
            if p["blur"] is not None:
                arr = orc.gaussian_blur(arr, *p["blur"])
                n_blur += 1
            x_ref = orc.to_tensor_normalize(arr, MEAN, STD)
            assert (x[b] - x_ref).abs().max().item() <= 1e-7, (name, p)      # incl. the blur: integer arithmetic on both sides
            n_exact += p["blur"] is None
    assert n_lab > 0 and n_scaled > 0 and n_blur > 0 and n_exact > 0
    # the round itself: labels grew by 10 px per image per stage on both dataset views, artefacts written
    for i in range(len(ds)):
        assert ds.queries[i].sum() == 12 + 2 * 10 and ds_q.queries[i].sum() == 12 + 2 * 10
    for nth in range(2):
        d = tmp_path / "checkpoints" / "synthetic" / f"{nth}_query"
        assert (d / "best_miou_model.pt").exists() and (d / "query_stats.pkl").exists()
    assert all(np.isfinite(h[5]) for h in m.history if h[0] == "train")
    # warm loop uploads no tables: every (rule, in, out) table was built once and is cached on the device
    n_up = aug.n_table_uploads
    aug(torch.from_numpy(np.stack(ds.imgs[:2])), torch.from_numpy(np.stack(ds.labels[:2])), torch.from_numpy(np.stack(ds.queries[:2])),
        params=seen[0][5][:2])
    assert aug.n_table_uploads == n_up
