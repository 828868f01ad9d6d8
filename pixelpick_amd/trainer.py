"""Flat-buffer training step for the PixelPick networks on MI355X (model.py:101-122 without torch.autograd).

One step = forward (engine tape) -> sparse cross-entropy (HIP) -> backward (tape replay, every weight
gradient written straight into ONE flat fp32 gradient buffer) -> [RCCL all-reduce of that buffer over
xGMI, one collective per step] -> fused Adam over the flat parameter buffer (pp_adam_step_flat; the
backbone/encoder segment at lr/10, utils/utils.py:125-141).  Parameters are views into one flat buffer,
backbone first, so the two learning-rate groups are two contiguous segments.

Data-parallel semantics: one process per GPU, per-GPU batch and per-GPU BatchNorm
statistics (the reference has no SyncBN), gradients averaged over ranks.  Nothing else crosses GPUs.
"""
from typing import Optional

import torch

from . import _lib
from . import engine as E


class FlatTrainer:
    def __init__(self, model, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-7, weight_decay: float = 2e-4,
                 ignore_index: int = 19, slow_module_names=("backbone", "encoder"), process_group=None):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.ignore_index = ignore_index
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        slow, fast = [], []
        seen = set()
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            (slow if name.split(".")[0] in slow_module_names else fast).append(p)
        self.params = slow + fast
        dev = self.params[0].device
        assert dev.type == "cuda", "FlatTrainer needs the model on the GPU"
        n = sum(p.numel() for p in self.params)
        self.n, self.n_split = n, sum(p.numel() for p in slow)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grad_view = {}
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)          # parameters alias the flat buffer
            self._grad_view[id(p)] = self.flat_g[off:off + k].view(p.shape)
            off += k
        self.step_count = 0
        self.lr_factor = 1.0
        self.last_loss: Optional[torch.Tensor] = None
        self.last_logits: Optional[torch.Tensor] = None

    # -----------------------------------------------------------------------------------------------
    def forward_backward(self, x: torch.Tensor, y: torch.Tensor, keep_logits: bool = False) -> torch.Tensor:
        """x [B,3,H,W] f32, y [B,H,W] int64 with ignore_index at unlabelled pixels (model.py:108-110)."""
        tape = E.Tape(enabled=True)
        tape.param_grad_dst = lambda p: self._grad_view.get(id(p))
        pred, _ = self.model._run(tape, x)
        loss, dlogits = E.cross_entropy_nchw(pred.t, y, self.ignore_index)
        self.last_logits = pred.t if keep_logits else None
        tape.backward(pred, dlogits)
        self.last_loss = loss
        return loss

    def all_reduce_grads(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.flat_g, op=torch.distributed.ReduceOp.SUM, group=self.pg)

    def optimizer_step(self):
        self.step_count += 1
        L = _lib.lib()
        rc = L.pp_adam_step_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                 self.exp_avg_sq.data_ptr(), self.n, self.n_split, self.lr / 10 * self.lr_factor,
                                 self.lr * self.lr_factor, self.betas[0], self.betas[1], self.eps, self.wd,
                                 self.step_count, 1.0 / self.world, _lib.current_stream_ptr())
        _lib.check(rc, "pp_adam_step_flat")

    def train_step(self, x: torch.Tensor, y: torch.Tensor, keep_logits: bool = False) -> torch.Tensor:
        self.model.train()
        loss = self.forward_backward(x, y, keep_logits)
        self.all_reduce_grads()
        self.optimizer_step()
        return loss

    def set_poly_lr(self, T: int, N: int, power: float = 0.9):
        """utils/lr_scheduler.py:15-17 applied to both segments."""
        self.lr_factor = pow((1 - 1.0 * T / N), power)
