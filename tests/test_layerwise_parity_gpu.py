"""Composed forward/backward parity at full chain depth, tight and without a noise term (tests/layerwise.py explains
the method): every convolution output, every BatchNorm/GroupNorm(+residual+activation) output, the gradient arriving at
each of them and every parameter gradient of DeepLabv3+-MobileNetV2 (182) and FPN-ResNet50 (213) against the
plain-PyTorch oracle ON THE ORACLE'S OWN INPUTS for that layer (reference: model.py:113-121 forward, cross_entropy,
backward).  The oracle is pinned to the imported reference module by module - outputs, arriving gradients, parameter
gradients, bit for bit - by tests/test_oracle_trace_golden.py (CPU suite).

Three modes (tests/layerwise.py):
  forced         every layer starts from the oracle's activation / arriving gradient: errors are those of ONE layer.
                 Bars (rel-L2 per tensor, NO noise allowance): forward sites 1e-4, arriving and parameter gradients 2e-4;
                 measured <= 2e-6 and <= 6e-6.
  branch-forced  the network runs FREE (nothing overwritten, rounding compounds through all layers, forward and backward)
                 except that the few ReLU/ReLU6 units whose branch differs from the oracle's run (tens out of 1e7..1e8)
                 are put on the oracle's side: every one of the 182 / 213 parameter gradients within 5e-4, no noise term.
  free-running   reported: how many units flip and what that does to a sparse-label gradient (1e-2, every tensor alike)."""
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from layerwise import LayerwiseParity, OracleTrace
from oracle.net import OracleDeepLab, OracleDeepLabR50, OracleFPN
from pixelpick_amd.networks.layers import Dropout
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_FWD, TOL_GRAD = 1e-4, 2e-4
# branch-aligned free-running network: measured worst parameter gradient 5.4e-5 (DeepLab 128x192), 4.7e-5 (DeepLab 256x512),
# 1.0e-4 (FPN 64x96) against the fp32 oracle; 3.2e-5 / 6.5e-5 / 6.0e-5 against the oracle evaluated in fp64
# (profiles/r02_layerwise_parity.txt).  The bar is half the north_star's 1e-3 and carries no noise term.
TOL_BRANCH = 5e-4


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
    o = {"deeplab": lambda: OracleDeepLab(C, 0.0, 0.0, 0.0), "FPN": lambda: OracleFPN(C),
         "deeplab_r50": lambda: OracleDeepLabR50(C, 0.0, 0.0, 0.0)}[network]()
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
    import pixelpick_amd.trainer as T
    # module-by-module comparison needs the reference's operation order: FPNSeg's low-resolution training tail (classifier in front
    # of the last x2 interpolation: same function, other tensors) is switched off here; test_networks_gpu.py holds it to the dense path
    keep, T.SPARSE_LOWRES_CE = T.SPARSE_LOWRES_CE, (T.SPARSE_LOWRES_CE and network != "FPN")
    try:
        with LayerwiseParity(m, trace, force=force) as lp:
            loss = tr.forward_backward(x.to(DEV), y.to(DEV))
            lp.compare_param_grads({n: tr._grad_view[id(p)] for n, p in m.named_parameters()})
    finally:
        T.SPARSE_LOWRES_CE = keep
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


def test_deeplab_r50_every_layer_forward_and_backward_matches_oracle():
    """The assembled DeepLabv3+-ResNet50 (SURVEY.md 0.1 extra): dilated ResNet50 + ASPP at output stride 8 (rates 12/24/36) +
    SegmentHead, every layer against the oracle of the same assembly (itself pinned to the reference's parts)."""
    lp, loss, o_loss, m, o = _run("deeplab_r50", 19, 19, 4, 64, 96, 20)      # (the image-pooling BatchNorm sees only B rows: ill-conditioned below 4)
    assert abs(loss - o_loss) <= 1e-5 * max(1.0, abs(o_loss))
    _assert_tight(lp, len(list(m.parameters())))
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
    <= 2e-5 of all units.  The gradients are dominated by a handful of flipped units at the segmentation head: with 20
    labelled pixels per image the gradient entering the last ReLU lives on 80 pixels x 256 channels = 20 K units, so ONE
    flipped unit there is 1/sqrt(20 K) = 0.7 % of every downstream gradient tensor (1.3e-2 .. 1.5e-2 measured, every
    tensor alike - see the branch-forced test below for the same run without that discontinuity)."""
    lp, loss, o_loss, m, o = _run(network, 19, 19, B, H, W, 20, force=False, key="free")
    print("\n[free-running] " + lp.summary().replace("\n", "\n[free-running] "))
    assert abs(loss - o_loss) <= 1e-3 * max(1.0, abs(o_loss))
    nfl = sum(f for f, _ in lp.flips.values())
    nun = sum(u for _, u in lp.flips.values())
    assert nfl <= max(4, 2e-5 * nun), (nfl, nun)
    errs = np.array([e for k, _, e, _ in lp.rec if k == "param_grad"])
    assert len(errs) == n_params and np.median(errs) <= 5e-2


@pytest.mark.parametrize("network,n_params,B,H,W", [("deeplab", 182, 4, 128, 192), ("deeplab", 182, 4, 256, 512), ("FPN", 213, 2, 64, 96),
                                                    ("deeplab_r50", 188, 4, 64, 96)])
def test_branch_forced_free_running_gradients_are_tight(network, n_params, B, H, W):
    """The whole network FREE-RUNNING (no tensor is overwritten, rounding compounds through all 60 layers forward and
    backward) with one intervention: the few ReLU/ReLU6 units whose branch differs from the oracle's run are put on the
    oracle's side, so both sides differentiate the same piecewise-linear function.  What is left is compounded fp32
    rounding: every parameter gradient within TOL_BRANCH rel-L2 of the oracle's, no noise term."""
    lp, loss, o_loss, m, o = _run(network, 19, 19, B, H, W, 20, force="branch", key="free")
    print("\n[branch-forced] " + lp.summary().replace("\n", "\n[branch-forced] "))
    errs = sorted(((e, n) for k, n, e, _ in lp.rec if k == "param_grad"), reverse=True)
    print("[branch-forced] worst parameter gradients:", [(n, f"{e:.2e}") for e, n in errs[:5]])
    assert len(errs) == n_params
    assert errs[0][0] <= TOL_BRANCH, errs[:8]


def _oracle_param_grads(network, C, ign, x, y, dtype):
    """Parameter gradients of the oracle evaluated in `dtype` on the formula weights (a second, independent evaluation of the
    SAME function: what two correct implementations of this train step differ by)."""
    _, o = _models(network, C)
    o = o.to(dtype)
    F.cross_entropy(o(x.to(dtype)), y, ignore_index=ign).backward()
    return {n: p.grad.float() for n, p in o.named_parameters()}


@pytest.mark.parametrize("network,n_params,B,H,W", [("deeplab", 182, 4, 128, 192), ("FPN", 213, 2, 64, 96)])
def test_free_running_dense_label_gradients_within_the_oracles_own_fp32_band(network, n_params, B, H, W):
    """Free-running, UNINTERVENED run (nothing overwritten, no ReLU unit aligned) on DENSE labels (~63 % of the pixels:
    formula_labels draws H*W positions with replacement), model.py:116-121.

    Measured (profiles/r03_dense_free_running.txt): forward sites agree to <= 3e-5, the gradient arriving at the classifier and
    at the last BatchNorm to 2e-5 - and the gradient LEAVING that BatchNorm is 4e-3 off, on every label density from 20 px/img
    to all pixels.  The ~40 ReLU units (of 1.9e7) that sit within the fp32 forward noise (2.5e-5 of a standard deviation) of
    their threshold take the other branch, and at the head a dense gradient lives on B*h*w*256/2 = 8e5 active units: ONE
    flipped unit is 1/sqrt(8e5) = 1.1e-3 of the tensor, not 1e-5, so label density does not buy the literal 1e-3.  The same
    holds between any two correct fp32 evaluations of this function; the yardstick used here is therefore the ORACLE AGAINST
    ITSELF: oracle/net.py evaluated in fp64 vs in fp32 on the same weights and inputs.  Bar: the HIP run's deviation from the
    fp32 oracle stays within 2.5x of that band in the median and in the maximum over all parameter gradients (measured ~1x),
    flipped units are reported, not aligned.  The no-discontinuity bar (branch-aligned, <= 5e-4, no noise term) is the test
    below this one."""
    lp, loss, o_loss, m, o = _run(network, 19, 19, B, H, W, H * W, force=False, key="dense")
    print("\n[dense free-running] " + lp.summary().replace("\n", "\n[dense free-running] "))
    assert abs(loss - o_loss) <= 1e-4 * max(1.0, abs(o_loss))
    x = fi.formula_input(B, H, W, key="xdense")
    y = fi.formula_labels(B, H, W, 19, 19, H * W, key="ydense")
    g32 = {n: p.grad for n, p in o.named_parameters()}
    g64 = _oracle_param_grads(network, 19, 19, x, y, torch.float64)
    from layerwise import rel_l2
    band = np.array([rel_l2(g32[n], g64[n]) for n in g32])
    errs = np.array([e for k, _, e, _ in lp.rec if k == "param_grad"])
    assert len(errs) == n_params == len(band)
    nfl = sum(f for f, _ in lp.flips.values())
    print(f"[dense free-running] flipped units (reported, not aligned): {nfl} of {sum(u for _, u in lp.flips.values())}")
    print(f"[dense free-running] HIP vs fp32 oracle: median {np.median(errs):.2e} max {errs.max():.2e} | "
          f"fp32 oracle vs fp64 oracle: median {np.median(band):.2e} max {band.max():.2e}")
    assert np.median(errs) <= 2.5 * np.median(band) and errs.max() <= 2.5 * band.max(), (np.median(errs), errs.max(), np.median(band), band.max())
    assert errs.max() <= 5e-2
