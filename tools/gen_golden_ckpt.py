#!/usr/bin/env python3
"""Checkpoint / pretrained-weight FORMAT goldens from the IMPORTED reference (authoring container only).

The released checkpoints and the ImageNet weight files cannot be fetched here, but what they ARE is fixed by the reference's code:
  (a) `mobilenet_v2-6a65762b.pth` - a torchvision-layout MobileNetV2 state_dict (keys `features.N...`, the ImageNet head
      `features.18.*` / `classifier.1.*`, no num_batches_tracked) that MobileNetV2._load_pretrained_model filters by key
      (networks/mobilenet_v2.py:139-147);
  (b) `resnet50-pytorch.pth` - a torchvision-layout ResNet50 state_dict (`conv1.weight`, `bn1.*`, `layerL.B.*`, `fc.*`) that
      ModuleHelper.load_model maps onto `prefix.*` (networks/backbones/module_helper.py:86-107, networks/encoder.py:28);
  (c) `best_miou_model.pt` = {"model": model.state_dict()} (model.py:208-213), loaded by `model.load_state_dict(d["model"])`.
This script lets the REFERENCE write / read such files with formula values (tests/formula_init.py), runs its eval forward and
stores: the key lists + shapes + dtypes of every file (so the GPU test can rebuild byte-equivalent files from the formula without
23 + 100 MB of weights in the repository) and samples of the logits the reference produced from them.  Data only."""
import contextlib
import io
import os
import sys
import tempfile
from argparse import Namespace
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import formula_init as fi  # noqa: E402
import networks.mobilenet_v2 as ref_mnv2  # noqa: E402
import networks.encoder as ref_encoder  # noqa: E402
from utils.utils import get_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ckpt_format.npz")
STRIDE = 17


def _args(network, weight_type, C):
    return Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=C, network_name=network, weight_type=weight_type,
                     use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)


def _template(keys_shapes):
    return OrderedDict((k, torch.zeros(s, dtype=torch.long if k.endswith("num_batches_tracked") else torch.float32)) for k, s in keys_shapes)


def _describe(sd):
    return (np.array(list(sd.keys())), np.array([",".join(str(d) for d in v.shape) for v in sd.values()]),
            np.array([str(v.dtype).replace("torch.", "") for v in sd.values()]))


def _eval_logits(model, C, H, W, key):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.eval()
    with torch.no_grad():
        p = model(fi.formula_input(1, H, W, key=key))["pred"]
    return p.reshape(-1)[::STRIDE].numpy().copy(), fi.summarize(p)


def main():
    out = {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    td = tempfile.mkdtemp()

    # ---- (a) torchvision-layout MobileNetV2 file through the reference's own filter ------------------------------------
    ref_mnv2.MobileNetV2._load_pretrained_model_orig = ref_mnv2.MobileNetV2._load_pretrained_model
    ref_mnv2.MobileNetV2._load_pretrained_model = lambda self: None
    with quiet:
        probe = ref_mnv2.MobileNetV2(output_stride=16, BatchNorm=torch.nn.BatchNorm2d)
    ks = [(k, tuple(v.shape)) for k, v in probe.state_dict().items()
          if k.startswith("features.") and not k.endswith("num_batches_tracked")]
    ks += [("features.18.0.weight", (1280, 320, 1, 1)), ("features.18.1.weight", (1280,)), ("features.18.1.bias", (1280,)),
           ("features.18.1.running_mean", (1280,)), ("features.18.1.running_var", (1280,)),
           ("classifier.1.weight", (1000, 1280)), ("classifier.1.bias", (1000,))]
    mn_file = fi.formula_state_dict(_template(ks), salt="#imagenet")
    p_mn = os.path.join(td, "mobilenet_v2-6a65762b.pth")
    torch.save(mn_file, p_mn)
    out["mnv2_file_keys"], out["mnv2_file_shapes"], _ = _describe(mn_file)
    ref_mnv2.MobileNetV2._load_pretrained_model = ref_mnv2.MobileNetV2._load_pretrained_model_orig
    ref_mnv2.model_zoo.load_url = lambda url, **kw: torch.load(p_mn)           # the download, served from the local file
    with quiet:
        m = get_model(_args("deeplab", "supervised", 19))
    # everything behind the backbone from the formula (the reference initialises it randomly)
    rest = {k: v for k, v in fi.formula_state_dict(m.state_dict()).items() if not k.startswith("backbone.")}
    m.load_state_dict(rest, strict=False)
    assert torch.equal(m.state_dict()["backbone.features.5.conv.3.weight"], mn_file["features.5.conv.3.weight"])
    assert torch.equal(m.state_dict()["backbone.high_level_features.17.conv.7.running_var"], mn_file["features.17.conv.7.running_var"])
    out["deeplab_pretrained_samples"], out["deeplab_pretrained_summary"] = _eval_logits(m, 19, 64, 96, "xckpt")

    # ---- (b) torchvision-layout ResNet50 file through ModuleHelper.load_model ------------------------------------------
    with quiet:
        m0 = get_model(_args("FPN", "random", 19))
    from networks.backbones.resnet_models import ResNet, Bottleneck
    with quiet:
        tv = ResNet(Bottleneck, [3, 4, 6, 3], num_classes=1000, deep_base=False, norm_type='batchnorm')
    ks = [(k[len("prefix."):] if k.startswith("prefix.") else k, tuple(v.shape)) for k, v in tv.state_dict().items()
          if not k.endswith("num_batches_tracked")]
    r50_file = fi.formula_state_dict(_template(ks), salt="#imagenet")
    p_r50 = os.path.join(td, "resnet50-pytorch.pth")
    torch.save(r50_file, p_r50)
    out["r50_file_keys"], out["r50_file_shapes"], _ = _describe(r50_file)
    ref_encoder.resnet[50] = p_r50
    with quiet:
        m = get_model(_args("FPN", "supervised", 19))
    rest = {k: v for k, v in fi.formula_state_dict(m.state_dict()).items() if not k.startswith("encoder.")}
    m.load_state_dict(rest, strict=False)
    assert torch.equal(m.state_dict()["encoder.base.prefix.conv1.weight"], r50_file["conv1.weight"])
    assert torch.equal(m.state_dict()["encoder.base.layer3.4.bn2.running_mean"], r50_file["layer3.4.bn2.running_mean"])
    out["fpn_pretrained_samples"], out["fpn_pretrained_summary"] = _eval_logits(m, 19, 64, 96, "xckpt")

    # ---- (c) {"model": state_dict} checkpoints written by the reference's own torch.save call --------------------------
    ref_mnv2.MobileNetV2._load_pretrained_model = lambda self: None
    for net, tag in (("deeplab", "deeplab"), ("FPN", "fpn")):
        with quiet:
            m = get_model(_args(net, "random", 19))
        sd = fi.tie_aliases(fi.formula_state_dict(m.state_dict(), salt="#ckpt"))
        for k in sd:                     # a training-time side effect a real checkpoint carries: advanced BatchNorm counters
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.tensor(1234, dtype=torch.long)
        m.load_state_dict(sd)
        assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())      # consistent aliases: the file IS the formula dict
        p = os.path.join(td, f"best_miou_model_{tag}.pt")
        torch.save({"model": m.state_dict()}, p)                                  # model.py:208-213
        with quiet:
            m2 = get_model(_args(net, "random", 19))
        m2.load_state_dict(torch.load(p)["model"])
        out[f"{tag}_ckpt_keys"], out[f"{tag}_ckpt_shapes"], out[f"{tag}_ckpt_dtypes"] = _describe(m2.state_dict())
        out[f"{tag}_ckpt_samples"], out[f"{tag}_ckpt_summary"] = _eval_logits(m2, 19, 64, 96, "xckpt")
    np.savez_compressed(OUT, **out)
    print("written", OUT, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
