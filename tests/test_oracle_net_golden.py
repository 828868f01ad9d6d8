"""Pins oracle/net.py (plain-PyTorch restatement of DeepLabv3+-MNv2) against the reference-generated
golden vectors.  CPU only; the larger train-mode check is bounded to one 128x192 step."""
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from oracle.net import OracleDeepLab

STRIDE = 29


def _build(C):
    m = OracleDeepLab(C, 0.0, 0.0, 0.0)
    m.load_state_dict(fi.formula_state_dict(m.state_dict()))
    return m


def test_state_dict_surface_identical_to_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_deeplab_cs128x192.npz"))
    sd = OracleDeepLab(19).state_dict()
    assert len(sd) == int(g["n_state_keys"])
    assert zlib.crc32("\n".join(f"{k}:{tuple(v.shape)}" for k, v in sd.items()).encode()) == int(g["state_keys_crc"])


@pytest.mark.parametrize("tag", ["voc40x56", "cv120x152"])
def test_eval_forward(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"net_deeplab_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C).eval()
    with torch.no_grad():
        pred = m(fi.formula_input(B, H, W, key=f"x{tag}"))
    ref = g["eval_pred_samples"]
    assert np.abs(pred.reshape(-1)[::STRIDE].numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_train_step_gradients(golden_dir):
    tag = "cs128x192"
    g = np.load(os.path.join(golden_dir, f"net_deeplab_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C).train()
    x = fi.formula_input(B, H, W, key=f"x{tag}")
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}")
    pred = m(x)
    loss = F.cross_entropy(pred, y, ignore_index=ign)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])) + 4 * float(g["loss_noise"])
    named = dict(m.named_parameters())
    for i, name in enumerate(g["grad_names"]):
        got, ref, noise = fi.summarize(named[str(name)].grad), g["grad_summary"][i], g["grad_noise"][i]
        assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] + 4 * noise[1], name     # same ATen kernels: essentially exact
    for k in g.files:
        if k.startswith("rs:"):
            np.testing.assert_allclose(m.state_dict()[k[3:]].numpy(), g[k], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------- FPN-ResNet50
def _build_fpn(C):
    from oracle.net import OracleFPN
    m = OracleFPN(C)
    m.load_state_dict(fi.formula_state_dict(m.state_dict()))
    return m


def test_fpn_state_dict_and_eval_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_fpn_cs64x96.npz"))
    m = _build_fpn(19)
    sd = m.state_dict()
    assert len(sd) == int(g["n_state_keys"]) == 372
    assert zlib.crc32("\n".join(f"{k}:{tuple(v.shape)}" for k, v in sd.items()).encode()) == int(g["state_keys_crc"])
    gv = np.load(os.path.join(golden_dir, "net_fpn_voc40x56.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in gv["shape"]]
    mv = _build_fpn(C).eval()
    with torch.no_grad():
        pred = mv(fi.formula_input(B, H, W, key="xvoc40x56"))
    ref = gv["eval_pred_samples"]
    assert np.abs(pred.reshape(-1)[::STRIDE].numpy() - ref).max() <= 1e-4 * np.abs(ref).max()


def test_fpn_train_step_gradients(golden_dir):
    tag = "cs64x96"
    g = np.load(os.path.join(golden_dir, f"net_fpn_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build_fpn(C).train()
    x = fi.formula_input(B, H, W, key=f"x{tag}")
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}")
    loss = F.cross_entropy(m(x), y, ignore_index=ign)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])) + 4 * float(g["loss_noise"])
    named = dict(m.named_parameters())
    assert [str(k) for k in g["grad_names"]] == list(named)
    for i, name in enumerate(g["grad_names"]):
        got, ref, noise = fi.summarize(named[str(name)].grad), g["grad_summary"][i], g["grad_noise"][i]
        assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] + 4 * noise[1], name


# ------------------------------------------------------------------------- DeepLabv3+-ResNet50 (assembled extra)
def _build_r50(C):
    from oracle.net import OracleDeepLabR50
    m = OracleDeepLabR50(C, 0.0, 0.0, 0.0)
    m.load_state_dict(fi.formula_state_dict(m.state_dict()))
    return m


def test_deeplab_r50_oracle_matches_the_assembly_of_reference_parts(golden_dir):
    """The goldens come from the reference's own ResNetBackbone('resnet50_dilated8') + ASPP('resnet', 8) + SegmentHead wired
    as deeplab.py:43-56 (tools/gen_golden_net.py --r50): state_dict surface, eval logits, one train step's loss, every
    parameter gradient and the BatchNorm running statistics."""
    g = np.load(os.path.join(golden_dir, "net_deeplab_r50_cs64x96.npz"))
    m = _build_r50(19)
    sd = m.state_dict()
    assert len(sd) == int(g["n_state_keys"]) == 374
    assert zlib.crc32("\n".join(f"{k}:{tuple(v.shape)}" for k, v in sd.items()).encode()) == int(g["state_keys_crc"])
    gv = np.load(os.path.join(golden_dir, "net_deeplab_r50_voc40x56.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in gv["shape"]]
    mv = _build_r50(C).eval()
    with torch.no_grad():
        pred = mv(fi.formula_input(B, H, W, key="xvoc40x56"))
    ref = gv["eval_pred_samples"]
    assert np.abs(pred.reshape(-1)[::STRIDE].numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    tag = "cs64x96"
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m.train()
    x = fi.formula_input(B, H, W, key=f"x{tag}")
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}")
    loss = F.cross_entropy(m(x), y, ignore_index=ign)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])) + 4 * float(g["loss_noise"])
    named = dict(m.named_parameters())
    assert [str(k) for k in g["grad_names"]] == list(named)
    for i, name in enumerate(g["grad_names"]):
        got, ref, noise = fi.summarize(named[str(name)].grad), g["grad_summary"][i], g["grad_noise"][i]
        assert abs(got[1] - ref[1]) <= 1e-4 * ref[1] + 4 * noise[1], name
    for k in g.files:
        if k.startswith("rs:"):
            np.testing.assert_allclose(m.state_dict()[k[3:]].numpy(), g[k], rtol=1e-5, atol=1e-6)
