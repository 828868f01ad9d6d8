"""Per-iteration polynomial learning-rate decay with the reference's constructor and call protocol
(utils/lr_scheduler.py:4-21; the driver calls `step(epoch=e-1)` once per iteration, model.py:138-139).

    lr_t = base_lr * (1 - t/T)^0.9,   t = epoch * iters_per_epoch + i,  T = num_epochs * iters_per_epoch
    (linear ramp t / warm-up iterations while t is inside the optional warm-up)

The position inside the epoch is kept here (`step()` receives only the epoch), advancing by one per `get_lr()` call and
wrapping at `iters_per_epoch`, exactly the bookkeeping the reference does; `poly_factor` is the bare scalar, which
`FlatTrainer.set_poly_lr` feeds to the fused Adam kernel."""
from torch.optim.lr_scheduler import _LRScheduler

POLY_POWER = 0.9


def poly_factor(T: int, N: int, power: float = POLY_POWER) -> float:
    """Scale applied to every base learning rate at global iteration T of N."""
    return (1.0 - float(T) / float(N)) ** power


class Poly(_LRScheduler):
    def __init__(self, optimizer, num_epochs, iters_per_epoch, warmup_epochs=0, last_epoch=-1):
        self._per_epoch = iters_per_epoch
        self._total = num_epochs * iters_per_epoch
        self._warmup = warmup_epochs * iters_per_epoch
        self._pos = 0                       # iterations already served in the current epoch
        super().__init__(optimizer, last_epoch)

    # the reference's attribute names, for code written against it
    N = property(lambda self: self._total)
    iters_per_epoch = property(lambda self: self._per_epoch)
    warmup_iters = property(lambda self: self._warmup)
    cur_iter = property(lambda self: self._pos)

    def scale_at(self, t: int) -> float:
        if 0 < self._warmup and t < self._warmup:
            return float(t) / self._warmup
        return poly_factor(t, self._total)

    def get_lr(self):
        t = self.last_epoch * self._per_epoch + self._pos
        scale = self.scale_at(t)
        self._pos = self._pos % self._per_epoch + 1
        if scale < 0:
            raise AssertionError("error in lr_scheduler")
        return [lr * scale for lr in self.base_lrs]
