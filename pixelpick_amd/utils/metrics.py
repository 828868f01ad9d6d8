"""AverageMeter / RunningScore with the reference's interface (utils/metrics.py:85-132,162-206).

RunningScore keeps the reference's numpy `update(label_trues, label_preds)` for host arrays and adds
`update_from_logits(y, logits)`: argmax + confusion-matrix histogram on the device (pp_confusion_matrix_update),
so a train/val step ships C*C int64 to the host instead of two full-resolution maps (model.py:124-125)."""
import numpy as np
import torch

from .. import _lib


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.initialized, self.val, self.avg, self.sum, self.count = False, 0, 0, 0, 0

    def update(self, val, weight=1):
        if not self.initialized:
            self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * weight, weight, True
        else:
            self.val = val
            self.sum = self.sum + val * weight
            self.count = self.count + weight
            self.avg = self.sum / self.count

    @property
    def value(self):
        return self.val

    @property
    def average(self):
        return np.round(self.avg, 5)


class RunningScore(object):
    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.confusion_matrix = np.zeros((n_classes, n_classes))
        self._dev_hist = None

    @staticmethod
    def _fast_hist(label_true, label_pred, n_class):
        mask = (label_true >= 0) & (label_true < n_class)
        return np.bincount(n_class * label_true[mask].astype(int) + label_pred[mask], minlength=n_class ** 2).reshape(n_class, n_class)

    def update(self, label_trues, label_preds):
        for lt, lp in zip(label_trues, label_preds):
            self.confusion_matrix += self._fast_hist(lt.flatten(), lp.flatten(), self.n_classes)

    def update_from_logits(self, y: torch.Tensor, logits: torch.Tensor):
        """y [B,H,W] int64, logits [B,C,H,W] f32, both on the GPU; accumulates on the device, no sync."""
        assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4
        B, C, H, W = logits.shape
        assert C == self.n_classes and logits.stride(3) == 1 and logits.stride(2) == W
        y = y.to(logits.device, torch.int64).contiguous()
        if self._dev_hist is None:
            self._dev_hist = torch.zeros((C, C), dtype=torch.int64, device=logits.device)
        rc = _lib.lib().pp_confusion_matrix_update(logits.data_ptr(), B, C, H * W, logits.stride(0), logits.stride(1), y.data_ptr(),
                                                   self._dev_hist.data_ptr(), _lib.current_stream_ptr(logits.device))
        _lib.check(rc, "pp_confusion_matrix_update")

    def _sync(self):
        if self._dev_hist is not None:
            self.confusion_matrix += self._dev_hist.cpu().numpy()
            self._dev_hist.zero_()

    def get_scores(self):
        self._sync()
        hist = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(hist).sum() / hist.sum()
            acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
            iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
            mean_iu = np.nanmean(iu)
            freq = hist.sum(axis=1) / hist.sum()
            fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
        return ({"Pixel Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu}, dict(zip(range(self.n_classes), iu)))

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))
        if self._dev_hist is not None:
            self._dev_hist.zero_()
