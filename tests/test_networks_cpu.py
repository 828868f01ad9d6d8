"""CPU-only: module surface / state_dict compatibility of the network mirrors (no compute)."""
import os
import zlib
from argparse import Namespace

import numpy as np
import pytest
import torch

import formula_init as fi
from pixelpick_amd.utils.utils import get_model, get_optimizer, get_lr_scheduler


def _args(n_classes=19):
    return Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=n_classes, network_name="deeplab",
                     weight_type="random", use_dilated_resnet=True, n_layers=50, width_multiplier=1.0,
                     dataset_name="cs", optimizer_params={"lr": 5e-4, "betas": (0.9, 0.999), "weight_decay": 2e-4, "eps": 1e-7},
                     lr_scheduler_type="Poly", n_epochs=50)


@pytest.fixture(scope="module")
def model():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return get_model(_args())


def test_state_dict_keys_and_shapes_match_reference(model, golden_dir):
    g = np.load(os.path.join(golden_dir, "net_deeplab_cs128x192.npz"))
    sd = model.state_dict()
    assert len(sd) == int(g["n_state_keys"]) == 668          # aliased backbone slices
    crc = zlib.crc32("\n".join(f"{k}:{tuple(v.shape)}" for k, v in sd.items()).encode())
    assert crc == int(g["state_keys_crc"])
    assert [k for k, _ in model.named_parameters()] == [str(k) for k in g["grad_names"]]


def _canonical(sd):
    """backbone.features.N.* and backbone.{low,high}_level_features.N.* alias the same tensors
    (mobilenet_v2.py:125-126); load_state_dict visits the aliases last, so their values win."""
    sd = dict(sd)
    for k in list(sd):
        if k.startswith("backbone.features."):
            n = int(k.split(".")[2])
            alias = k.replace("backbone.features.", "backbone.low_level_features." if n < 4 else "backbone.high_level_features.", 1)
            sd[k] = sd[alias]
    return sd


def test_state_dict_roundtrip_in_reference_layout(model):
    sd = _canonical(fi.formula_state_dict(model.state_dict()))
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    sd2 = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(sd2[k], v), k
    w = model.seg_head.segment_head[0].weight            # kernel layout HWIO, reference layout OIHW
    assert tuple(w.shape) == (3, 3, 304, 256)
    assert torch.equal(w.detach().permute(3, 2, 0, 1), sd["seg_head.segment_head.0.weight"])
    dw = model.backbone.features[1].conv[0]
    assert dw.depthwise and tuple(dw.weight.shape) == (3, 3, 32)
    assert tuple(sd2["backbone.features.1.conv.0.weight"].shape) == (32, 1, 3, 3)


def test_parameter_counts_and_optimizer_groups(model):
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(model) == 5815539                            # DeepLab-MNv2 at C=19
    assert (n(model.backbone), n(model.aspp), n(model.low_level_conv), n(model.seg_head)) == (1811712, 2706432, 1248, 1296147)
    args = _args()
    opt = get_optimizer(args, model)
    assert [g["lr"] for g in opt.param_groups] == [5e-5, 5e-4, 5e-4, 5e-4]
    # the reference hands only lr and weight_decay to Adam (utils/utils.py:141): eps/betas are torch's defaults although
    # args.optimizer_params carries "eps": 1e-7
    assert all(g["eps"] == 1e-8 and g["betas"] == (0.9, 0.999) and g["weight_decay"] == 2e-4 for g in opt.param_groups)
    from pixelpick_amd.utils.utils import optimizer_spec
    voc = _args(); voc.dataset_name = "voc"
    assert optimizer_spec(voc) == ("sgd", 1e-3, 1e-2, 5e-4, 0.9)
    sgd = get_optimizer(voc, model)
    assert type(sgd).__name__ == "SGD" and [g["lr"] for g in sgd.param_groups] == [1e-3, 1e-2, 1e-2, 1e-2]
    assert all(g["momentum"] == 0.9 and g["weight_decay"] == 5e-4 for g in sgd.param_groups)
    cv = _args(); cv.dataset_name = "cv"; cv.optimizer_type = "SGD"
    assert optimizer_spec(cv)[0] == "sgd" and optimizer_spec(_args())[0] == "adam"
    assert [len(g["params"]) for g in opt.param_groups] == [153, 18, 3, 8]
    sched = get_lr_scheduler(args, opt, iters_per_epoch=10)
    lrs = []
    for _ in range(3):
        opt.step()
        sched.step(epoch=0)
        lrs.append(opt.param_groups[1]["lr"])
    expect = [5e-4 * (1 - t / 500) ** 0.9 for t in (1, 2, 3)]
    assert np.allclose(lrs, expect, rtol=1e-12)


def test_dropout_toggles_and_cpu_input_is_rejected(model):
    from pixelpick_amd.networks.layers import Dropout
    model.eval()
    assert all(not m.training for m in model.modules() if isinstance(m, Dropout))
    model.turn_on_dropout()
    assert all(m.training for m in model.modules() if isinstance(m, Dropout))
    assert not model.aspp.bn1.training
    model.turn_off_dropout()
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 32, 32))


def test_mc_dropout_variant_builds_with_the_reference_layout():
    """use_mc_dropout=True (mobilenet_v2.py:114-115,127): an nn.Dropout2d closes `features` (index 18, no parameters) and a
    second one sits on the low-level branch; state_dict keys / shapes are those of the plain model, and neither Dropout2d
    is touched by turn_on_dropout / turn_off_dropout (deeplab.py:33-41 toggles nn.Dropout only)."""
    import warnings
    from argparse import Namespace
    from pixelpick_amd.networks.layers import Dropout, Dropout2d
    from pixelpick_amd.utils.utils import get_model
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = get_model(Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=19, network_name="deeplab"))
        mc = get_model(Namespace(use_mc_dropout=True, mc_dropout_p=0.2, n_classes=19, network_name="deeplab"))
    assert [(k, tuple(v.shape)) for k, v in mc.state_dict().items()] == [(k, tuple(v.shape)) for k, v in plain.state_dict().items()]
    assert isinstance(mc.backbone.features[18], Dropout2d) and isinstance(mc.backbone.high_level_features[-1], Dropout2d)
    assert isinstance(mc.backbone.dropout, Dropout2d) and mc.backbone.mc_dropout
    assert not any(isinstance(m, Dropout2d) for m in plain.backbone.features)
    mc.eval()
    mc.turn_on_dropout()
    assert all(m.training for m in mc.modules() if isinstance(m, Dropout))
    assert not any(m.training for m in mc.modules() if isinstance(m, Dropout2d))


def test_deeplab_r50_extra_has_the_surface_of_the_assembled_reference_parts(golden_dir):
    """network_name="deeplab_r50" (SURVEY.md 0.1 extra): state_dict keys / shapes equal those of the reference's
    ResNetBackbone('resnet50_dilated8') + ASPP('resnet', 8) + low-level conv + SegmentHead assembly the goldens were generated
    from, parameters are in the reference's order, and the optimiser groups of utils/utils.py:125-141 apply unchanged."""
    import warnings
    a = _args()
    a.network_name = "deeplab_r50"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(a)
    g = np.load(os.path.join(golden_dir, "net_deeplab_r50_cs64x96.npz"))
    sd = m.state_dict()
    assert len(sd) == int(g["n_state_keys"]) == 374
    assert zlib.crc32("\n".join(f"{k}:{tuple(v.shape)}" for k, v in sd.items()).encode()) == int(g["state_keys_crc"])
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in g["grad_names"]]
    n = lambda mod: sum(p.numel() for p in mod.parameters())
    assert n(m) == 40351667 and n(m.backbone) == 23508032 and tuple(m.aspp.aspp2.atrous_conv.weight.shape) == (3, 3, 2048, 256)
    assert m.aspp.aspp2.atrous_conv.dilation == 12 and m.aspp.aspp4.atrous_conv.dilation == 36      # aspp.py:43-44, output stride 8
    opt = get_optimizer(a, m)
    assert [gr["lr"] for gr in opt.param_groups] == [5e-5, 5e-4, 5e-4, 5e-4]


def test_fpn_tail_commutes_in_the_reference_arithmetic():
    """The identity FPNSeg's training tail rests on (pixelpick_amd/networks/decoders.py FPNDecoder.run(lowres=True)), stated with
    the reference's own operators: decoders.py:79-81,101  classifier(sum_i interpolate(q_i, x2))  ==  interpolate(classifier(sum_i q_i), x2)
    - a 1x1 convolution with bias, a sum and a bilinear interpolation whose weights sum to one are linear and commute.  float64: exact to
    rounding; float32: the two orders differ by ulps (why the acquisition path keeps the reference's order)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for dt, tol in ((torch.float64, 1e-13), (torch.float32, 2e-6)):
        qs = [torch.randn(2, 128, 12, 20, generator=g, dtype=dt) for _ in range(4)]
        w = torch.randn(19, 128, 1, 1, generator=g, dtype=dt) / 11.0
        b = torch.randn(19, generator=g, dtype=dt)
        dense = F.conv2d(sum(F.interpolate(q, scale_factor=2, mode="bilinear") for q in qs), w, b)
        low = F.interpolate(F.conv2d(sum(qs), w, b), scale_factor=2, mode="bilinear")
        assert (dense - low).abs().max().item() <= tol * dense.abs().max().item()
