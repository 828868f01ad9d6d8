"""GPU parity of the DeepLabv3+-MobileNetV2 mirror against golden vectors produced by the imported
reference (tools/gen_golden_net.py): logits, loss, every parameter gradient (summaries + selected full
tensors) and BatchNorm running statistics.  Tolerance 1e-3 relative to the tensor scale (north_star)."""
import os
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import formula_init as fi
from pixelpick_amd.networks.layers import Dropout
from pixelpick_amd.trainer import FlatTrainer
from pixelpick_amd.utils.utils import get_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def _args(n_classes):
    return Namespace(use_mc_dropout=False, mc_dropout_p=0.2, n_classes=n_classes, network_name="deeplab",
                     weight_type="random", use_dilated_resnet=True, n_layers=50, width_multiplier=1.0)


def _build(n_classes, network="deeplab"):
    a = _args(n_classes)
    a.network_name = network
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = get_model(a)
    m.load_state_dict(fi.formula_state_dict(m.state_dict()))
    for mod in m.modules():
        if isinstance(mod, Dropout):
            mod.p = 0.0
    return m.to(DEV)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _to_oihw(name, g):
    if g.dim() == 4:
        return g.permute(3, 2, 0, 1)
    if g.dim() == 3:
        return g.permute(2, 0, 1).unsqueeze(1)
    return g


STRIDE = 29   # tools/gen_golden_net.py SAMPLE_STRIDE


@pytest.mark.parametrize("tag", ["cs128x192", "cv120x152", "voc40x56"])
def test_eval_forward_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"net_deeplab_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C).eval()
    x = fi.formula_input(B, H, W, key=f"x{tag}").to(DEV)
    with torch.no_grad():
        out = m(x)
    pred = out["pred"]
    assert pred.shape == (B, C, H, W)
    assert _rel(pred.reshape(-1)[::STRIDE].cpu().numpy(), g["eval_pred_samples"]) < TOL
    assert _rel(fi.summarize(pred), g["eval_pred_summary"]) < TOL
    emb = out["emb"]                                   # lazy, full resolution like deeplab.py:58-59
    assert emb.shape == (B, 256, H, W)


def _check_grads(g, grads_by_name, full=True):
    """Every parameter gradient against the reference.  Allowed deviation per tensor: 1e-3 of its scale plus
    4x the reference's OWN deviation under a 1e-6 relative input perturbation (fixture `grad_noise`): ReLU
    mask flips and cancelling sums make some gradients of this train-mode-BN network move by up to 2e-2 under
    such noise in the reference itself, so a flat 1e-3 cannot be met even by the reference vs itself."""
    worst = 0.0
    for i, name in enumerate(g["grad_names"]):
        got = fi.summarize(grads_by_name[str(name)])
        ref, noise = g["grad_summary"][i], g["grad_noise"][i]
        for j, scale_j in ((1, 1), (2, 2), (0, 1)):       # abs-sum, max, sum (sum relative to abs-sum)
            # `noise` = largest deviation over 7 reference probes (6 perturbed fp32 + fp64, tools/gen_golden_net.py).
            # Unit flips are heavy-tailed, so the band is 4x that floor; a wrong kernel is off by orders of magnitude.
            tol = TOL * max(ref[scale_j], 1e-12) + 4 * noise[j]
            assert abs(got[j] - ref[j]) <= tol, f"{name}[{j}]: {got[j]} vs {ref[j]} (tol {tol:.3e}, noise {noise[j]:.3e})"
        worst = max(worst, abs(got[1] - ref[1]) / max(ref[1], 1e-12))
    if full:
        for k in g.files:
            if k.startswith("g:"):
                name = k[2:]
                got = _to_oihw(name, grads_by_name[name]).cpu().numpy()
                err = np.abs(got.astype(np.float64) - g[k]).max()
                assert err <= TOL * np.abs(g[k]).max() + 4 * float(g["gn:" + name]), f"{name}: {err}"
    return worst


@pytest.mark.parametrize("tag", ["cs128x192", "cv120x152"])
def test_train_step_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"net_deeplab_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C).train()
    x = fi.formula_input(B, H, W, key=f"x{tag}").to(DEV)
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}").to(DEV)
    pred = m(x)["pred"]
    loss = F.cross_entropy(pred, y, ignore_index=ign)          # the reference's own call (model.py:116)
    loss.backward()
    ref_s = g["train_pred_samples"]
    err = np.abs(pred.detach().reshape(-1)[::STRIDE].cpu().numpy().astype(np.float64) - ref_s).max()
    assert err <= TOL * np.abs(ref_s).max() + 4 * float(g["train_pred_noise"]), f"logits err {err}"
    assert abs(loss.item() - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    worst = _check_grads(g, {k: p.grad for k, p in m.named_parameters()})
    for k in g.files:
        if k.startswith("rs:"):
            assert _rel(m.state_dict()[k[3:]].cpu().numpy(), g[k]) < TOL, k
    print(f"[{tag}] logits max err {err:.2e}; worst relative abs-sum gradient deviation {worst:.2e}")


def test_flat_trainer_matches_autograd_path_and_is_deterministic(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_deeplab_cs128x192.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    x = fi.formula_input(B, H, W, key="xcs128x192").to(DEV)
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key="ycs128x192").to(DEV)
    m = _build(C).train()
    tr = FlatTrainer(m, ignore_index=ign)
    loss = tr.forward_backward(x, y)
    assert abs(loss.item() - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    g1 = tr.flat_g.clone()
    _check_grads(g, {k: tr._grad_view[id(p)] for k, p in m.named_parameters()})
    # bit-exact repeatability of the whole forward/backward (deterministic reductions, no float atomics)
    m2 = _build(C).train()
    tr2 = FlatTrainer(m2, ignore_index=ign)
    tr2.forward_backward(x, y)
    assert torch.equal(tr2.flat_g, g1)
    # one optimiser step == torch.optim.Adam on the same gradients (utils/utils.py:125-141 groups)
    p_before = tr.flat_p.clone()
    ref_p = p_before.clone().cpu()
    ga = g1.cpu()
    pa = ref_p[:tr.n_split].clone().requires_grad_(True)
    pb = ref_p[tr.n_split:].clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [pa], "lr": 5e-5, "weight_decay": 2e-4}, {"params": [pb], "lr": 5e-4, "weight_decay": 2e-4}],
                           betas=(0.9, 0.999), eps=1e-8)          # FlatTrainer's default = what the reference's Adam really uses
    pa.grad, pb.grad = ga[:tr.n_split].clone(), ga[tr.n_split:].clone()
    opt.step()
    tr.optimizer_step()
    assert (tr.flat_p.cpu() - torch.cat([pa.detach(), pb.detach()])).abs().max().item() < 1e-6
    assert tr.n_split == 1811712 and tr.n == 5815539
    # parameters alias the flat buffer: the module sees the update
    assert m.seg_head.classifier.bias.data_ptr() >= tr.flat_p.data_ptr()


# ---------------------------------------------------------------------------------------------- FPN-ResNet50 (R1-R4)
def test_fpn_eval_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "net_fpn_voc40x56.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C, "FPN").eval()
    with torch.no_grad():
        out = m(fi.formula_input(B, H, W, key="xvoc40x56").to(DEV))
    assert out["pred"].shape == (B, C, H, W) and out["emb"].shape == (B, 128, H, W)
    assert _rel(out["pred"].reshape(-1)[::STRIDE].cpu().numpy(), g["eval_pred_samples"]) < TOL


def test_fpn_train_step_matches_reference(golden_dir):
    tag = "cs64x96"
    g = np.load(os.path.join(golden_dir, f"net_fpn_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C, "FPN").train()
    x = fi.formula_input(B, H, W, key=f"x{tag}").to(DEV)
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}").to(DEV)
    pred = m(x)["pred"]
    loss = F.cross_entropy(pred, y, ignore_index=ign)
    loss.backward()
    ref_s = g["train_pred_samples"]
    err = np.abs(pred.detach().reshape(-1)[::STRIDE].cpu().numpy().astype(np.float64) - ref_s).max()
    assert err <= TOL * np.abs(ref_s).max() + 4 * float(g["train_pred_noise"]), f"logits err {err}"
    assert abs(loss.item() - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    worst = _check_grads(g, {k: p.grad for k, p in m.named_parameters()})
    for k in g.files:
        if k.startswith("rs:"):
            assert _rel(m.state_dict()[k[3:]].cpu().numpy(), g[k]) < TOL, k
    print(f"[fpn {tag}] logits max err {err:.2e}; worst relative abs-sum gradient deviation {worst:.2e}")


# ------------------------------------------------------------------------- DeepLabv3+-ResNet50 (assembled extra)
def test_deeplab_r50_matches_the_assembly_of_reference_parts(golden_dir):
    """network_name="deeplab_r50": dilated ResNet50 + ASPP('resnet', output stride 8: 2048 channels, rates 1/12/24/36) +
    SegmentHead, against goldens generated from the reference's own classes wired as deeplab.py:43-56."""
    g = np.load(os.path.join(golden_dir, "net_deeplab_r50_voc40x56.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C, "deeplab_r50").eval()
    with torch.no_grad():
        out = m(fi.formula_input(B, H, W, key="xvoc40x56").to(DEV))
    assert out["pred"].shape == (B, C, H, W)
    assert _rel(out["pred"].reshape(-1)[::STRIDE].cpu().numpy(), g["eval_pred_samples"]) < TOL
    tag = "cs64x96"
    g = np.load(os.path.join(golden_dir, f"net_deeplab_r50_{tag}.npz"))
    B, H, W, C, ign, n_lab = [int(v) for v in g["shape"]]
    m = _build(C, "deeplab_r50").train()
    assert len(m.state_dict()) == int(g["n_state_keys"]) == 374
    x = fi.formula_input(B, H, W, key=f"x{tag}").to(DEV)
    y = fi.formula_labels(B, H, W, C, ign, n_lab, key=f"y{tag}").to(DEV)
    pred = m(x)["pred"]
    loss = F.cross_entropy(pred, y, ignore_index=ign)
    loss.backward()
    ref_s = g["train_pred_samples"]
    err = np.abs(pred.detach().reshape(-1)[::STRIDE].cpu().numpy().astype(np.float64) - ref_s).max()
    assert err <= TOL * np.abs(ref_s).max() + 4 * float(g["train_pred_noise"]), f"logits err {err}"
    assert abs(loss.item() - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    worst = _check_grads(g, {k: p.grad for k, p in m.named_parameters()})
    for k in g.files:
        if k.startswith("rs:"):
            assert _rel(m.state_dict()[k[3:]].cpu().numpy(), g[k]) < TOL, k
    print(f"[deeplab_r50 {tag}] logits max err {err:.2e}; worst relative abs-sum gradient deviation {worst:.2e}")
    # the flat trainer (sparse low-resolution CE, two streams) walks the same step
    m2 = _build(C, "deeplab_r50").train()
    tr = FlatTrainer(m2, ignore_index=ign)
    l2 = tr.forward_backward(x, y)
    assert abs(l2.item() - float(g["loss"])) < TOL * max(1.0, abs(float(g["loss"])))
    _check_grads(g, {k: tr._grad_view[id(p)] for k, p in m2.named_parameters()})


def test_fpn_low_resolution_training_tail_equals_the_dense_order(monkeypatch):
    """FPNSeg's train step computes the classifier at HALF resolution on the sum of the four branches and interpolates its output at the
    labelled pixels only (FPNDecoder.run(lowres=True) + engine.cross_entropy_lowres(align_corners=False)): classifier, branch sum and x2
    interpolation are linear, so this is the reference's decoders.py:79-81,101 + model.py:116 in another order.  Loss and EVERY parameter
    gradient must equal the dense order (full-resolution emb / pred, dense loss) to fp32 rounding; with keep_logits the full-size logits too."""
    import pixelpick_amd.trainer as T
    C, B, H, W = 19, 2, 64, 96
    x = fi.formula_input(B, H, W, key="fl").to(DEV)
    y = fi.formula_labels(B, H, W, C, C, 20, key="fl").to(DEV)
    res = {}
    for lowres in (True, False):
        monkeypatch.setattr(T, "SPARSE_LOWRES_CE", lowres)
        m = _build(C, "FPN").train()
        tr = FlatTrainer(m, ignore_index=C)
        loss = tr.forward_backward(x, y, keep_logits=True)
        torch.cuda.synchronize()
        res[lowres] = (loss.item(), {k: tr._grad_view[id(p)].clone() for k, p in m.named_parameters()}, tr.last_logits.clone())
    la, ga, za = res[True]
    lb, gb, zb = res[False]
    assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb))
    assert tuple(za.shape) == tuple(zb.shape) == (B, C, H, W)
    assert (za - zb).abs().max().item() <= 2e-5 * zb.abs().max().item()
    worst = max(((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-12)).item() for k in gb)
    assert worst <= 5e-5, worst


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("network", ["deeplab", "FPN"])
def test_launch_plan_replay_matches_eager_steps(network, native, monkeypatch):
    """FlatTrainer.enable_replay: the recorded launch list (C-ABI calls + stream forks / joins, re-issued from a loop) must
    walk the same parameter trajectory as eager steps, bit for bit, with inputs that change from step to step, dropout
    active (device-side seed) and the BatchNorm step counters still advancing."""
    from pixelpick_amd import _lib
    from pixelpick_amd import engine as E
    from pixelpick_amd import trainer as T
    from pixelpick_amd.networks.layers import BatchNorm2d
    # native: the plan lives in the library and a step is ONE pp_plan_replay call (csrc/plan.hip); otherwise the round-2 Python list
    monkeypatch.setattr(T, "NATIVE_PLAN", native)
    C, B, H, W = 19, 2, 64, 96
    data = [(fi.formula_input(B, H, W, key=f"r{i}").to(DEV), fi.formula_labels(B, H, W, C, C, 20, key=f"r{i}").to(DEV)) for i in range(3)]

    def run(use_plan):
        m = _build(C, network)
        for mod in m.modules():
            if isinstance(mod, Dropout):
                mod.p = 0.3
        tr = FlatTrainer(m.train(), ignore_index=C)
        E.set_dropout_device_seed(tr._seed_dev)
        losses = []
        for i in range(5):
            xb, yb = data[i % 3]
            if not use_plan:
                tr.step_count += 1
                tr._stage_hyper()
                losses.append(tr._step_body(xb, yb, False, True).item())
            elif i == 0:
                tr.enable_replay(xb, yb, warmup=0)          # the recorded step is a real step
                assert len(tr._plan) > 100
                assert isinstance(tr._plan, _lib.NativePlan) == native
                if native:
                    # single rank: no host break - the whole step is one stretch; the BatchNorm step counters ride behind it
                    assert not tr._plan.breaks and tr._plan.native_ops() > 100 and len(tr._plan.host_notes) > 10
                losses.append(tr.last_loss.item())
            else:
                losses.append(tr.train_step(xb, yb).item())
        nbt = [int(mod.state_dict()["num_batches_tracked"]) for mod in m.modules() if isinstance(mod, BatchNorm2d)]
        p = tr.flat_p.clone()
        if use_plan:
            tr.disable_replay()
        E.set_dropout_device_seed(None)
        return p, losses, nbt

    p_eager, l_eager, n_eager = run(False)
    p_plan, l_plan, n_plan = run(True)
    assert l_eager == l_plan
    assert torch.equal(p_eager, p_plan)
    assert n_eager == n_plan and (not n_eager or set(n_eager) == {5})


def test_replay_survives_a_larger_eager_forward_that_regrows_the_conv_scratch(monkeypatch):
    """A recorded plan holds the ADDRESS of the shared convolution scratch (split-K partials, bf16x3 planes).  An eager forward
    at a larger shape between two replays (the per-epoch validation, model.py:190-194) makes engine._conv_ws re-allocate it:
    the replaced block must stay alive (engine._SCRATCH_RETIRED), otherwise later replays write into memory the caching
    allocator has handed to other tensors.  The scratch floor is lowered so that the small test shapes force the re-allocation."""
    from pixelpick_amd import engine as E
    C, B, H, W = 19, 2, 64, 96
    data = [(fi.formula_input(B, H, W, key=f"g{i}").to(DEV), fi.formula_labels(B, H, W, C, C, 20, key=f"g{i}").to(DEV)) for i in range(2)]
    big = fi.formula_input(4, 192, 256, key="gbig").to(DEV)
    big_y = fi.formula_labels(4, 192, 256, C, C, 20, key="gbig").to(DEV)

    def run(disturb):
        monkeypatch.setattr(E, "_CONV_WS_MIN", 1 << 16)
        E._CONV_WS_BUF.clear()
        m = _build(C, "deeplab")
        tr = FlatTrainer(m.train(), ignore_index=C)
        tr.enable_replay(*data[0], warmup=0)
        first = next(iter(E._CONV_WS_BUF.values()))
        ptr0, n0 = first.data_ptr(), first.numel()
        sentinels = []
        if disturb:
            # an eager step of ANOTHER model at a larger shape (validation / a second trainer in the process): needs more scratch than
            # the recorded step -> re-allocation
            FlatTrainer(_build(C, "deeplab").train(), ignore_index=C).forward_backward(big, big_y)
            cur = next(iter(E._CONV_WS_BUF.values()))
            assert cur.numel() > n0 and cur.data_ptr() != ptr0
            assert any(b.data_ptr() == ptr0 for b in E._SCRATCH_RETIRED)      # the plan's block is still owned
            del first, cur
            torch.cuda.synchronize()
            # whatever the allocator would have re-used: blocks of the old scratch's size class, filled with a pattern
            sentinels = [torch.full((n0 // 4,), 7.25, device=DEV) for _ in range(6)]
        losses = [tr.train_step(*data[i % 2]).item() for i in range(1, 4)]
        torch.cuda.synchronize()
        assert all(bool((t == 7.25).all()) for t in sentinels)
        p = tr.flat_p.clone()
        tr.disable_replay()
        return losses, p

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert l0 == l1 and torch.equal(p0, p1)


@pytest.mark.parametrize("network,shape", [("deeplab", (2, 72, 88)), ("deeplab", (1, 128, 192)), ("FPN", (2, 64, 96))])
def test_inference_with_fused_conv_bn_act_is_bit_identical(network, shape):
    """Inference folds eval-mode BatchNorm (+ residual + activation) into the producing convolution's epilogue
    (pp_conv2d_fwd_bn_act / pp_dwconv3x3_fwd_bn_act): same arithmetic as conv -> bn_eval_affine -> scale_shift_act,
    so the logits are bit-identical; MC-dropout inference (dropout on in eval mode) takes the same path."""
    from pixelpick_amd import engine as E
    B, H, W = shape
    m = _build(19, network).eval()
    # non-trivial running statistics
    g = torch.Generator().manual_seed(3)
    for mod in m.modules():
        if hasattr(mod, "running_var"):
            mod.running_mean.copy_((torch.randn(mod.running_mean.shape, generator=g) * 0.2).to(DEV))
            mod.running_var.copy_((torch.rand(mod.running_var.shape, generator=g) + 0.5).to(DEV))
    x = fi.formula_input(B, H, W, key="fuse").to(DEV)
    outs = []
    for fuse in (True, False, True):
        E._FUSE_EVAL = fuse
        try:
            with torch.no_grad():
                outs.append(m(x)["pred"].clone())
        finally:
            E._FUSE_EVAL = True
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0


def test_baseline_size_train_steps_are_bitwise_reproducible():
    """Three runs of 8 optimisation steps at the BASELINE shape (B=4, 256x512, 19 classes) from identical state end in
    bit-identical parameters: no atomics on floats, fixed-order reductions, and the cross-XCD exchange of the
    single-launch BatchNorm is coherent (a stale partial showed up here as run-to-run differences)."""
    from pixelpick_amd import engine as E
    sys_path_bench = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    import sys
    sys.path.insert(0, sys_path_bench)
    from bench import synth_train_batch
    x, y = synth_train_batch(4, 19, 256, 512, 20, torch.device(DEV), 1)
    sigs = []
    for _ in range(3):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = get_model(_args(19)).to(DEV).train()
        tr = FlatTrainer(m, ignore_index=19)
        E.set_dropout_seed(1234)
        losses = [tr.train_step(x, y) for _ in range(8)]
        torch.cuda.synchronize()
        sigs.append((tr.flat_p.clone(), torch.stack([l.reshape(()) for l in losses])))
    for p, l in sigs[1:]:
        assert torch.equal(l, sigs[0][1]), "loss trajectory differs between identical runs"
        assert torch.equal(p, sigs[0][0]), "parameters differ between identical runs"


def test_flat_trainer_sgd_is_torch_sgd_with_the_voc_groups():
    """The voc optimiser (SGD momentum 0.9, lr 1e-3 backbone / 1e-2 rest, wd 5e-4 - utils/utils.py:208-240) through
    FlatTrainer: three optimiser steps on the gradients of one forward/backward equal torch.optim.SGD on the same
    gradients and groups (momentum buffer initialised by the first step, as torch does)."""
    from pixelpick_amd.utils.utils import optimizer_spec
    a = _args(7); a.dataset_name = "voc"; a.optimizer_params = {"lr": 1e-2, "weight_decay": 1e-4, "momentum": 0.9}
    kind, slow_lr, lr, wd, mom = optimizer_spec(a)
    assert (kind, slow_lr, lr, wd, mom) == ("sgd", 1e-3, 1e-2, 5e-4, 0.9)
    x = fi.formula_input(2, 64, 96, key="sgdx").to(DEV)
    y = fi.formula_labels(2, 64, 96, 7, 255, 12, key="sgdy").to(DEV)
    m1 = _build(7).train()
    tr = FlatTrainer(m1, lr=lr, slow_lr=slow_lr, weight_decay=wd, optimizer=kind, momentum=mom, ignore_index=255)
    tr.forward_backward(x, y)
    g = tr.flat_g.clone().cpu()
    ref_p = tr.flat_p.clone().cpu()
    pa = ref_p[:tr.n_split].clone().requires_grad_(True)
    pb = ref_p[tr.n_split:].clone().requires_grad_(True)
    opt = torch.optim.SGD([{"params": [pa], "lr": 1e-3, "weight_decay": 5e-4, "momentum": 0.9},
                           {"params": [pb], "lr": 1e-2, "weight_decay": 5e-4, "momentum": 0.9}])
    for step in range(1, 4):
        pa.grad, pb.grad = g[:tr.n_split].clone(), g[tr.n_split:].clone()
        opt.step()
        tr.step_count = step
        tr.optimizer_step()
        assert (tr.flat_p.cpu() - torch.cat([pa.detach(), pb.detach()])).abs().max().item() < 2e-6


def test_mc_dropout_training_variant_runs_through_dropout2d():
    """--use_mc_dropout (mobilenet_v2.py:114-115,133-134): nn.Dropout2d on the high-level and low-level encoder features in
    train mode.  A step is finite, bit-reproducible for a fixed seed, differs from the plain model's (the masks are live),
    and in eval mode (Dropout2d inactive, as the reference) the logits equal the plain model's with the same weights."""
    from pixelpick_amd import engine as E
    a = _args(19)
    a.use_mc_dropout = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mc = get_model(a)
    sd = fi.formula_state_dict(mc.state_dict())
    mc.load_state_dict(sd)
    plain = _build(19)
    for mod in mc.modules():
        if isinstance(mod, Dropout):
            mod.p = 0.0                                   # only the two Dropout2d stay stochastic
    mc = mc.to(DEV)
    x = fi.formula_input(2, 64, 96, key="mc").to(DEV)
    y = fi.formula_labels(2, 64, 96, 19, 19, 20, key="mc").to(DEV)
    with torch.no_grad():
        assert torch.equal(mc.eval()(x)["pred"], plain.eval()(x)["pred"])
    losses, grads = [], []
    for rep in range(2):
        E.set_dropout_seed(7)
        tr = FlatTrainer(mc.train(), ignore_index=19)
        losses.append(tr.forward_backward(x, y).item())
        grads.append(tr.flat_g.clone())
    assert np.isfinite(losses[0]) and losses[0] == losses[1] and torch.equal(grads[0], grads[1])
    tp = FlatTrainer(plain.train(), ignore_index=19)
    assert abs(tp.forward_backward(x, y).item() - losses[0]) > 1e-6


@pytest.mark.parametrize("mode", ["eager", "replay"])
def test_weight_planes_split_at_the_start_of_the_step_change_nothing(mode, monkeypatch):
    """engine._X3_WPL: the bf16x3 layers' weight planes (SegmentHead, decoders.py:107-114) are split at begin_step() on the
    weight-gradient stream from the second step on and handed to the convolutions (pp_conv2d_*_pre2) instead of being split in front
    of each launch.  Same planes, same kernels: the parameter trajectory must not move a bit - eager, and with the step recorded
    after the first eager step (what model.py does) and replayed."""
    from pixelpick_amd import _lib
    from pixelpick_amd import engine as E
    C, B, H, W = 19, 4, 256, 384                    # 4 x 64 x 96 = 24576 rows at the head: the smallest map that takes the bf16x3 kernels
    data = [(fi.formula_input(B, H, W, key=f"w{i}").to(DEV), fi.formula_labels(B, H, W, C, C, 20, key=f"w{i}").to(DEV)) for i in range(2)]
    calls = []
    real = _lib.lib().pp_x3_split_weights

    def run(prefetch):
        monkeypatch.setattr(E, "_X3_WPRE", prefetch)
        E._X3_WPL.clear()
        m = _build(C)
        tr = FlatTrainer(m.train(), ignore_index=C)
        E.set_dropout_device_seed(tr._seed_dev)
        for i in range(4):
            xb, yb = data[i % 2]
            if mode == "replay" and i == 1:
                tr.step_count += 0
                tr.enable_replay(xb, yb, warmup=0)
            elif mode == "replay" and i > 1:
                tr.train_step(xb, yb)
            else:
                tr.step_count += 1
                tr._stage_hyper()
                tr._step_body(xb, yb, False, True)
        torch.cuda.synchronize()
        p = tr.flat_p.clone()
        n_reg = sum(len(e["planes"]) for e in E._X3_WPL.values())
        if mode == "replay":
            tr.disable_replay()
        E.set_dropout_device_seed(None)
        return p, n_reg

    p_off, n_off = run(False)
    p_on, n_on = run(True)
    assert n_off == 0 and n_on == 4, (n_off, n_on)          # two layers x (forward, backward-data) layouts registered
    assert torch.equal(p_on, p_off)
    E._X3_WPL.clear()


def test_dense_labels_take_the_dense_backward_kernels():
    """The reference's fully supervised mode (model.py:106-110 with n_pixels_by_us == 0: every pixel labelled): FlatTrainer(sparse_labels=False)
    leaves the loss gradient unflagged, so the classifier's weight gradient and the BatchNorm backward behind the loss run their dense
    kernels.  Same arithmetic as the row-gather kernels on an all-flagged gradient up to summation order: the gradients agree to 1e-5."""
    C, B, H, W = 19, 2, 64, 96
    x = fi.formula_input(B, H, W, key="dense").to(DEV)
    y = ((torch.arange(B * H * W) * 7 + 3) % C).view(B, H, W).to(DEV)          # every pixel labelled
    grads = {}
    for sparse in (True, False):
        m = _build(C)
        tr = FlatTrainer(m.train(), ignore_index=C, sparse_labels=sparse)
        tr.forward_backward(x, y)
        torch.cuda.synchronize()
        grads[sparse] = tr.flat_g.clone()
    d = (grads[True].double() - grads[False].double()).norm().item() / grads[False].double().norm().item()
    assert d <= 1e-5, d
