"""Factories with the reference's names (utils/utils.py): get_model, get_optimizer, get_lr_scheduler."""
import torch

from ..networks.deeplab import DeepLab


def get_model(args):
    """utils/utils.py:15-51."""
    if args.network_name == "deeplab":
        return DeepLab(args)
    if args.network_name == "FPN":
        from ..networks.model import FPNSeg
        return FPNSeg(args)
    raise ValueError(args.network_name)


def _param_groups(args, model):
    """Backbone / encoder at lr/10, everything else at lr (utils/utils.py:117-139)."""
    op = args.optimizer_params
    if args.network_name == "FPN":
        return [{'params': model.encoder.parameters(), 'lr': op['lr'] / 10, 'weight_decay': op['weight_decay']},
                {'params': model.decoder.parameters(), 'lr': op['lr'], 'weight_decay': op['weight_decay']}]
    groups = [{'params': model.backbone.parameters(), 'lr': op['lr'] / 10, 'weight_decay': op['weight_decay']}]
    for part in (model.aspp, model.low_level_conv, model.seg_head):
        groups.append({'params': part.parameters(), 'lr': op['lr'], 'weight_decay': op['weight_decay']})
    return groups


def get_optimizer(args, model):
    """utils/utils.py:112-306: Adam (cs / cv default / custom) or SGD (voc, cv with optimizer_type SGD)."""
    op = args.optimizer_params
    use_sgd = args.dataset_name == "voc" or (args.dataset_name == "cv" and getattr(args, "optimizer_type", "Adam") == "SGD")
    if use_sgd:
        from torch.optim import SGD
        groups = _param_groups(args, model)
        for g in groups:
            g['momentum'] = op.get('momentum', 0.9)
        return SGD(groups)
    from torch.optim import Adam
    groups = _param_groups(args, model)
    kw = {}
    if 'betas' in op:
        kw['betas'] = op['betas']
    if 'eps' in op:
        kw['eps'] = op['eps']
    return Adam(groups, **kw)


def get_lr_scheduler(args, optimizer, iters_per_epoch=-1):
    """utils/utils.py:309-335."""
    if args.dataset_name == "voc" or args.lr_scheduler_type == "Poly":
        from .lr_scheduler import Poly
        return Poly(optimizer, args.n_epochs, iters_per_epoch)
    return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[20, 40], gamma=0.1)
