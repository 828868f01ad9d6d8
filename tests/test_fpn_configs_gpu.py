"""BASELINE configs[2..4] on the HIP path at their real sizes (round-1 tests ran the ResNet50 model at 64x96 / 40x56 only):
the FPN-ResNet50 train step at 256x512 per-GPU batch 4 (configs[2]), and the network leg of configs[4]: FPNSeg eval forward
at 1024x2048 followed by least-confidence acquisition on the logits it produced (networks/model.py:6-14,
resnet_models.py:115-121, decoders.py:57-77, query.py:190-204).  Size-independent properties: finiteness, determinism,
batch independence in eval mode, picks == exact top-k of the device's own score map (checked by the CPU oracle)."""
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch

import formula_init as fi
from oracle import acq as orc
from pixelpick_amd import acquisition as acq
from pixelpick_amd.trainer import FlatTrainer
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
    return m.to(DEV)


def test_fpn_train_step_at_the_baseline_shape_is_finite_and_bit_reproducible():
    B, H, W, C = 4, 256, 512, 19
    x = fi.formula_input(B, H, W, key="fpnfull").to(DEV)
    y = fi.formula_labels(B, H, W, C, C, 20, key="fpnfull").to(DEV)
    runs = []
    for rep in range(2):
        tr = FlatTrainer(_fpn().train(), ignore_index=C, slow_module_names=("encoder",))
        losses = [tr.train_step(x, y).item() for _ in range(3)]
        runs.append((losses, tr.flat_p.clone(), tr.flat_g.clone()))
    assert all(np.isfinite(l) for l in runs[0][0]) and runs[0][0][2] < runs[0][0][0]          # the step learns the batch
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert torch.isfinite(runs[0][2]).all() and (runs[0][2] != 0).float().mean().item() > 0.99   # every weight gets a gradient


def test_fpn_full_resolution_forward_and_least_confidence_acquisition():
    """configs[4]: 1024x2048, least-confidence, B=1 per forward (the reference's query loop, query.py:159)."""
    C, H, W, k = 19, 1024, 2048, 20
    m = _fpn().eval()
    x = fi.formula_input(1, H, W, key="fpn1024").to(DEV)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        out = m(x)
        pred = out["pred"]
    assert pred.shape == (1, C, H, W) and torch.isfinite(pred).all()
    excl = torch.zeros((1, H, W), dtype=torch.uint8, device=DEV)
    excl[0, ::7, ::5] = 1
    idx, val, omap = acq.score_topk(pred, excl, "least_confidence", k, return_map=True)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"\\n[configs[4] network leg] FPNSeg eval forward + LC top-{k} at {H}x{W}: peak device memory {peak:.2f} GiB")
    assert peak < 24.0
    dmap = omap[0].cpu().numpy()
    e_idx, _ = orc.topk(dmap, k, True)
    assert idx[0].cpu().numpy().tolist() == e_idx.tolist()                     # exact top-k of the device's own map
    assert not excl.reshape(-1)[idx[0].long()].any()
    # the map itself against the oracle on a crop of the produced logits
    crop = pred[:, :, 500:516, 1000:1064].contiguous()
    np.testing.assert_allclose(acq.score_map(crop, None, "least_confidence").cpu().numpy(),
                               orc.score_map(crop.cpu().numpy(), "least_confidence"), rtol=2e-5, atol=2e-6)
    # eval-mode batch independence at a size where two images fit comfortably: image 0 alone == image 0 in a pair
    xs = fi.formula_input(2, 256, 512, key="fpnpair").to(DEV)
    with torch.no_grad():
        p2 = m(xs)["pred"]
        p1 = m(xs[:1])["pred"]
    assert (p2[:1] - p1).abs().max().item() <= 1e-4 * p1.abs().max().item()
    emb = out["emb"]                                                          # lazy [1,128,H,W] like decoders.py:75-77
    assert emb.shape == (1, 128, H, W)
