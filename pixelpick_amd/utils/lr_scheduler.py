"""Poly learning-rate schedule — mirror of utils/lr_scheduler.py:4-21 (per-iteration poly decay, power 0.9)."""
from torch.optim.lr_scheduler import _LRScheduler


class Poly(_LRScheduler):
    def __init__(self, optimizer, num_epochs, iters_per_epoch, warmup_epochs=0, last_epoch=-1):
        self.iters_per_epoch = iters_per_epoch
        self.cur_iter = 0
        self.N = num_epochs * iters_per_epoch
        self.warmup_iters = warmup_epochs * iters_per_epoch
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        T = self.last_epoch * self.iters_per_epoch + self.cur_iter
        factor = pow((1 - 1.0 * T / self.N), 0.9)
        if self.warmup_iters > 0 and T < self.warmup_iters:
            factor = 1.0 * T / self.warmup_iters
        self.cur_iter %= self.iters_per_epoch
        self.cur_iter += 1
        assert factor >= 0, 'error in lr_scheduler'
        return [base_lr * factor for base_lr in self.base_lrs]


def poly_factor(T: int, N: int, power: float = 0.9) -> float:
    """The scalar the schedule multiplies every base lr with at global iteration T of N."""
    return pow((1 - 1.0 * T / N), power)
