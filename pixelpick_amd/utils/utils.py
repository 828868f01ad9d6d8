"""Factories with the reference's names (utils/utils.py): get_model, get_optimizer, get_lr_scheduler."""
import torch

from ..networks.deeplab import DeepLab


def get_model(args):
    """utils/utils.py:15-51."""
    if args.network_name == "deeplab":
        return DeepLab(args)
    if args.network_name == "deeplab_r50":        # not a reference choice (args.py:19): the assembled extra of SURVEY.md 0.1
        return DeepLab(args, backbone='resnet', output_stride=8)
    if args.network_name == "FPN":
        from ..networks.model import FPNSeg
        return FPNSeg(args)
    raise ValueError(args.network_name)


def optimizer_spec(args):
    """What utils/utils.py:112-306 builds, as plain numbers: (kind, slow_lr, lr, weight_decay, momentum).

    cs, and cv/custom with optimizer_type "Adam": Adam(backbone|encoder at lr/10, rest at lr, weight_decay) - the
    reference hands ONLY lr and weight_decay to torch.optim.Adam, so betas and eps are torch's defaults (0.9, 0.999) and
    1e-8 even though args.optimizer_params carries "betas" and "eps": 1e-7.  voc, and cv/custom with optimizer_type "SGD":
    SGD(momentum 0.9) with HARD-CODED lr 1e-3 (backbone|encoder) / 1e-2 (rest) and weight decay 5e-4 (1e-4 for voc + FPN);
    args.optimizer_params is not read at all there."""
    op = args.optimizer_params
    name, fpn = args.dataset_name, args.network_name == "FPN"
    if name == "voc":
        return "sgd", 1e-3, 1e-2, (1e-4 if fpn else 5e-4), 0.9
    if name != "cs" and getattr(args, "optimizer_type", "Adam") == "SGD":
        return "sgd", 1e-3, 1e-2, 5e-4, 0.9
    return "adam", op['lr'] / 10, op['lr'], op['weight_decay'], 0.0


def _param_groups(args, model, slow_lr, lr, weight_decay, extra):
    if args.network_name == "FPN":
        parts = [(model.encoder, slow_lr), (model.decoder, lr)]
    else:
        parts = [(model.backbone, slow_lr), (model.aspp, lr), (model.low_level_conv, lr), (model.seg_head, lr)]
    return [dict({'params': m.parameters(), 'lr': l, 'weight_decay': weight_decay}, **extra) for m, l in parts]


def get_optimizer(args, model):
    """utils/utils.py:112-306 (see optimizer_spec for the quirks that are reproduced)."""
    kind, slow_lr, lr, wd, momentum = optimizer_spec(args)
    if kind == "sgd":
        from torch.optim import SGD
        return SGD(_param_groups(args, model, slow_lr, lr, wd, {'momentum': momentum}))
    from torch.optim import Adam
    return Adam(_param_groups(args, model, slow_lr, lr, wd, {}))


def get_lr_scheduler(args, optimizer, iters_per_epoch=-1):
    """utils/utils.py:309-335."""
    if args.dataset_name == "voc" or args.lr_scheduler_type == "Poly":
        from .lr_scheduler import Poly
        return Poly(optimizer, args.n_epochs, iters_per_epoch)
    return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[20, 40], gamma=0.1)
