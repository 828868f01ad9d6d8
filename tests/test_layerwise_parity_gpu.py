"""Composed forward/backward parity at full chain depth, tight and without a noise term (tests/layerwise.py explains
the method): every convolution output, every BatchNorm/GroupNorm(+residual+activation) output, the gradient arriving at
each of them and every parameter gradient of DeepLabv3+-MobileNetV2 (182) and FPN-ResNet50 (213) against the
plain-PyTorch oracle ON THE ORACLE'S OWN INPUTS for that layer (reference: model.py:113-121 forward, cross_entropy,
backward).  The oracle is pinned to the imported reference by tests/test_oracle_net_golden.py and
tests/test_oracle_tight_golden.py (CPU suite).

Bars (rel-L2 per tensor, NO noise allowance): forward sites 1e-4, arriving gradients and parameter gradients 2e-4
(observed: ~1e-6; see the printed summaries)."""
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from layerwise import LayerwiseParity, OracleTrace
from oracle.net import OracleDeepLab, OracleFPN
from pixelpick_amd.networks.layers import Dropout
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_FWD, TOL_GRAD = 1e-4, 2e-4


def _models(network, C, salt=""):
    a = Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=network, weight_type="random",
                  use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(a)
    sd = fi.formula_state_dict(m.state_dict()) if not salt else fi.formula_state_dict(m.state_dict(), salt=salt)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, Dropout):
            mod.p = 0.0
    o = OracleDeepLab(C, 0.0, 0.0, 0.0) if network == "deeplab" else OracleFPN(C)
    o.load_state_dict(sd)
    return m.to(DEV).train(), o.train()


def _oracle_step(o, x, y, ign):
    tr = OracleTrace(o)
    logits = o(x)
    loss = F.cross_entropy(logits, y, ignore_index=ign)
    loss.backward()
    tr.close()
    return tr, loss.item()


def _run(network, C, ign, B, H, W, n_lab, force=True, key="lw"):
    m, o = _models(network, C)
    x = fi.formula_input(B, H, W, key=f"x{key}")
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{key}")
    trace, o_loss = _oracle_step(o, x, y, ign)
    tr = FlatTrainer(m, ignore_index=ign)
    with LayerwiseParity(m, trace, force=force) as lp:
        loss = tr.forward_backward(x.to(DEV), y.to(DEV))
        lp.compare_param_grads({n: tr._grad_view[id(p)] for n, p in m.named_parameters()})
    return lp, loss.item(), o_loss, m, o


def _assert_tight(lp, n_params):
    print("\n" + lp.summary())
    assert len([r for r in lp.rec if r[0] == "param_grad"]) == n_params
    for kind, tol in (("conv", TOL_FWD), ("norm", TOL_FWD), ("dy", TOL_GRAD), ("param_grad", TOL_GRAD)):
        bad = [(n, e) for k, n, e, _ in lp.rec if k == kind and not e <= tol]
        assert not bad, f"{kind}: {len(bad)} tensors above {tol:g}: {bad[:8]}"


@pytest.mark.parametrize("shape", [(4, 64, 96, 20), (3, 72, 88, 10)])
def test_deeplab_every_layer_forward_and_backward_matches_oracle(shape):
    B, H, W, n_lab = shape
    lp, loss, o_loss, m, o = _run("deeplab", 19, 19, B, H, W, n_lab)
    assert abs(loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss))
    _assert_tight(lp, 182)
    # forced run: no unit may take another branch than the oracle's (each layer saw the oracle's input)
    assert sum(f for f, _ in lp.flips.values()) <= 2, lp.flips
    # BatchNorm running statistics after the step (momentum update from the forced inputs)
    osd = o.state_dict()
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            ref = osd[k]
            assert (v.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), k


def test_fpn_every_layer_forward_and_backward_matches_oracle():
    lp, loss, o_loss, m, o = _run("FPN", 19, 19, 2, 64, 96, 20)
    assert abs(loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss))
    _assert_tight(lp, 213)
    assert sum(f for f, _ in lp.flips.values()) <= 2, lp.flips


def test_deeplab_every_layer_at_the_baseline_shape():
    """BASELINE configs[1]: 256x512, per-GPU batch 4, 20 labelled pixels per image (the shape bench.py times)."""
    lp, loss, o_loss, m, o = _run("deeplab", 19, 19, 4, 256, 512, 20, key="base")
    assert abs(loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss))
    _assert_tight(lp, 182)


@pytest.mark.parametrize("network,n_params,B,H,W", [("deeplab", 182, 4, 128, 192), ("FPN", 213, 2, 64, 96)])
def test_free_running_noise_is_reported_and_bounded(network, n_params, B, H, W):
    """Same bookkeeping WITHOUT forcing: how far fp32 rounding compounds through the train-mode network and how many
    ReLU/ReLU6 units end up on the other branch than in the oracle's run.  Bounds here are those of the phenomenon (the
    oracle's own fp32-vs-fp64 deviation has the same size, tools/act_deviation.py), not kernel bars: flipped units
    <= 2e-5 of all units, median parameter-gradient rel-L2 <= 5e-3."""
    lp, loss, o_loss, m, o = _run(network, 19, 19, B, H, W, 20, force=False, key="free")
    print("\n[free-running] " + lp.summary().replace("\n", "\n[free-running] "))
    assert abs(loss - o_loss) <= 1e-3 * max(1.0, abs(o_loss))
    nfl = sum(f for f, _ in lp.flips.values())
    nun = sum(u for _, u in lp.flips.values())
    assert nfl <= max(4, 2e-5 * nun), (nfl, nun)
    errs = np.array([e for k, _, e, _ in lp.rec if k == "param_grad"])
    assert len(errs) == n_params and np.median(errs) <= 5e-3
