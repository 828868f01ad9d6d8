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

import os

import torch

from . import _lib
from . import engine as E


# PIXELPICK_OVERLAP_ALLREDUCE=0: one all-reduce of the whole flat gradient after backward (the round-1 behaviour)
OVERLAP_ALLREDUCE = os.environ.get("PIXELPICK_OVERLAP_ALLREDUCE", "1") != "0"
# PIXELPICK_FORCE_COLLECTIVES=1: issue the gradient all-reduces even in a one-rank process group (lets a single-GPU box
# exercise the RCCL call path - communicator setup, the overlapped bucket on the helper stream - end to end)
# PIXELPICK_SPARSE_LOWRES_CE=0: materialise the full-size logits and run the dense loss kernels even for models whose
# last op is the x4 bilinear upsample (DeepLab)
SPARSE_LOWRES_CE = os.environ.get("PIXELPICK_SPARSE_LOWRES_CE", "1") != "0"
FORCE_COLLECTIVES = os.environ.get("PIXELPICK_FORCE_COLLECTIVES", "0") == "1"
# PIXELPICK_NATIVE_PLAN=0: enable_replay() keeps the recorded step as a Python list and re-issues it call by call (the round-2
# form) instead of handing it to the library's executor (pp_plan_replay, csrc/plan.hip: one foreign call per step)
NATIVE_PLAN = os.environ.get("PIXELPICK_NATIVE_PLAN", "1") != "0"


_RESERVE = {"users": 0, "before": None}     # collective trainers alive in this process / the library's reserve before the first of them


def default_comm_cu_reserve(backend: str) -> int:
    """CUs set aside for the resident communication kernel (pp_set_comm_cu_reserve): one per RCCL channel - NCCL_MAX_NCHANNELS
    when the job caps it, else 32 - and none for a backend whose collectives do not run on the CUs (gloo)."""
    if backend != "nccl":
        return 0
    try:
        n = int(os.environ.get("NCCL_MAX_NCHANNELS", "32"))
    except ValueError:
        n = 32
    return max(0, min(n, 128))


class FlatTrainer:
    def __init__(self, model, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 2e-4,
                 ignore_index: int = 19, slow_module_names=("backbone", "encoder"), process_group=None,
                 optimizer: str = "adam", momentum: float = 0.9, slow_lr: float = None, sparse_labels: bool = True):
        """optimizer "adam": torch.optim.Adam(lr/10 for the backbone|encoder, lr for the rest) - what the reference builds for
        cs / cv (utils/utils.py:114-141; NOTE it passes only lr and weight_decay to Adam, so betas/eps are torch's defaults
        (0.9, 0.999), 1e-8 whatever args.optimizer_params says).  "sgd": torch.optim.SGD(momentum) with `slow_lr` for the
        backbone|encoder (default lr/10) - voc and optimizer_type "SGD" (utils/utils.py:208-270)."""
        assert optimizer in ("adam", "sgd")
        self.model = model
        self.optimizer, self.momentum = optimizer, momentum
        self.slow_lr = lr / 10 if slow_lr is None else slow_lr
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.ignore_index = ignore_index
        # True (model.py:108-110 with n_pixels_by_us != 0: a handful of labelled pixels per image): the loss gradient carries row flags and
        # the classifier / BatchNorm backward behind it visit only the flagged rows.  False (the reference's fully supervised mode,
        # n_pixels_by_us == 0: every pixel labelled): no flags - with every row flagged the row-gather kernels are slower than the dense ones
        self.sparse_labels = bool(sparse_labels)
        self.pg = process_group
        self.world = 1
        self.collectives = False
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.collectives = self.world > 1 or (FORCE_COLLECTIVES and torch.distributed.is_available() and torch.distributed.is_initialized())
        if self.collectives and not E._BN_FUSED_DIST:
            E.disable_fused_bn_for_collectives()
        if self.collectives:
            # RCCL's channel blocks stay resident on some CUs for the whole of an overlapped all-reduce while launches whose blocks
            # wait for each other (single-launch BatchNorm, convolution + BatchNorm) need their WHOLE grid resident: those launches
            # size themselves against occupancy x (CUs - reserve).  PIXELPICK_COMM_CU_RESERVE (default 32 under RCCL, 0 otherwise;
            # tests/test_dist_gpu.py parks an occupier kernel on 16 / 32 / 64 CUs for 200 two-rank steps).
            # Default under RCCL: one CU per channel the communicator may open - NCCL_MAX_NCHANNELS when the job sets it, else 32
            # (RCCL's default channel ceiling on an 8-GPU xGMI node); 0 for any other backend.  The value in force before the first
            # collective trainer is restored when the last one closes (close() / __del__): the library setting is process-wide.
            reserve = int(os.environ.get("PIXELPICK_COMM_CU_RESERVE", default_comm_cu_reserve(torch.distributed.get_backend(self.pg))))
            if _RESERVE["users"] == 0:
                _RESERVE["before"] = _lib.lib().pp_get_comm_cu_reserve()
            _RESERVE["users"] += 1
            self._holds_reserve = True
            if reserve != _lib.lib().pp_get_comm_cu_reserve():
                _lib.lib().pp_set_comm_cu_reserve(reserve)
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
        # a second boundary INSIDE the encoder (backbones that mark it and name the modules behind it: late_modules()): flat_g[n_mid:n_split] are the
        # gradients of its late blocks, complete long before the backward pass ends
        self.n_mid = 0
        late = next((m for m in model.modules() if callable(getattr(m, "late_modules", None))), None)
        if late is not None:
            late_ids = {id(p) for m in late.late_modules() for p in m.parameters()}
            off, first = 0, None
            ok = True
            for p in slow:
                if id(p) in late_ids:
                    if first is None:
                        first = off
                elif first is not None:
                    ok = False                    # a non-late parameter behind the first late one: no contiguous segment
                off += p.numel()
            if ok and first is not None and 0 < first < self.n_split:
                self.n_mid = first
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
        if self.world > 1:
            # replicas must start identical whatever each rank's RNG drew at construction (DistributedDataParallel does the
            # same at wrap time): rank 0's parameters and BatchNorm statistics win
            src = torch.distributed.get_global_rank(self.pg, 0) if self.pg is not None else 0
            torch.distributed.broadcast(self.flat_p, src=src, group=self.pg)
            self.sync_buffers()
        self.step_count = 0
        self.lr_factor = 1.0
        self.last_loss: Optional[torch.Tensor] = None
        self.last_logits: Optional[torch.Tensor] = None
        # run-time hyper-parameters live in device memory so that the whole step can be replayed from a recorded launch list:
        # hyper = [lr_backbone, lr_head, 1-beta1^t, sqrt(1-beta2^t)], seed = per-step dropout base seed
        # (a ring of pinned staging slots: the host runs ahead of graph replays, so a slot is only rewritten once the copy
        # that read it has executed - its event is waited for first)
        self._stage = [(torch.zeros(4, dtype=torch.float32).pin_memory(), torch.zeros(1, dtype=torch.int64).pin_memory(),
                        torch.cuda.Event()) for _ in range(4)]
        self._stage_used = [False] * len(self._stage)
        self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self._seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._plan = self._plan_pool = self._plan_stream = None
        self._gx = self._gy = None

    # -----------------------------------------------------------------------------------------------
    def _ensure_train_mode(self):
        """model.train() (model.py:104) without its ~400 setattr calls when every module already is in train mode."""
        mods = self.__dict__.get("_all_modules")
        if mods is None:
            mods = self.__dict__["_all_modules"] = list(self.model.modules())
        for m in mods:
            if not m.training:
                self.model.train()
                return

    def forward_backward(self, x: torch.Tensor, y: torch.Tensor, keep_logits: bool = False) -> torch.Tensor:
        """x [B,3,H,W] f32, y [B,H,W] int64 with ignore_index at unlabelled pixels (model.py:108-110)."""
        tape = E.Tape(enabled=True)
        tape.param_grad_dst = lambda p: self._grad_view.get(id(p))
        self._early_work = None
        self._mid_work = None
        if self.collectives and OVERLAP_ALLREDUCE and self.n_split < self.n:
            tape.hooks["encoder_done"] = self._early_all_reduce
            if self.n_mid > 0:
                tape.hooks["encoder_late_done"] = self._mid_all_reduce
        if SPARSE_LOWRES_CE and getattr(self.model, "LOWRES_LOGITS", False):
            # deeplab.py:55-56 + model.py:116 without the [B,C,H,W] logits: the loss kernels interpolate the classifier output
            # at the labelled pixels only (80 of 524 288 here) and hand its gradient straight to the classifier conv
            low, _ = self.model._run(tape, x, upsample=False)
            size = tuple(x.shape[2:])
            align = bool(getattr(self.model, "LOWRES_ALIGN_CORNERS", True))     # DeepLab: align_corners=True x4; FPNSeg: False, x2
            loss, dlow = E.cross_entropy_lowres(low.t, size, y, self.ignore_index, align_corners=align, sparse=self.sparse_labels)
            self.last_logits = (E.bilinear(E.Tape(False), low, size, align, 0.0 if align else float(getattr(self.model, "LOWRES_SCALE_FACTOR", 0.0)),
                                           out_nchw=True).t if keep_logits else None)
            tape.backward(low, dlow)
        else:
            pred, _ = self.model._run(tape, x)
            loss, dlogits = E.cross_entropy_nchw(pred.t, y, self.ignore_index, sparse=self.sparse_labels)
            self.last_logits = pred.t if keep_logits else None
            tape.backward(pred, dlogits)
        self.last_loss = loss
        return loss

    def _early_all_reduce(self, tape):
        _lib.plan_note(self._early_all_reduce_on, tuple(tape.side_streams_in_use()))

    def _early_all_reduce_on(self, sides):
        """Backward hook at the encoder boundary: the gradients of everything BEHIND the encoder (flat_g[n_split:], 16 of
        the 23 MB for DeepLabv3+-MNv2) are complete once the work already enqueued on the main and the weight-gradient
        streams has run, while the whole encoder backward (~3 ms) is still ahead.  Their all-reduce is issued now from a
        helper stream that waits on exactly that work, so it runs under the encoder backward; the main stream only
        waits for it in all_reduce_grads()."""
        main = torch.cuda.current_stream()
        comm = self.__dict__.get("_comm_stream")
        if comm is None:
            comm = self.__dict__["_comm_stream"] = torch.cuda.Stream(device=self.flat_g.device)
        comm.wait_stream(main)
        for s in sides:
            comm.wait_stream(s)
        with torch.cuda.stream(comm):
            t0 = self._comm_mark(comm)
            self._early_work = torch.distributed.all_reduce(self.flat_g[self.n_split:], op=torch.distributed.ReduceOp.SUM,
                                                            group=self.pg, async_op=True)
            self._comm_mark(comm, "behind_encoder", t0)
        E.refresh_stream()

    def _mid_all_reduce(self, tape):
        _lib.plan_note(self._mid_all_reduce_on, tuple(tape.side_streams_in_use()))

    def _mid_all_reduce_on(self, sides):
        """Backward hook inside the encoder (Tape.mark "encoder_late_done"): flat_g[n_mid:n_split] - the late encoder blocks, 6.3 of
        the encoder's 7.2 MB for MobileNetV2 - goes out on the helper stream behind the first bucket, under the ~1.5 ms of
        backward that are still ahead; what is left for after the join is 0.9 MB."""
        main = torch.cuda.current_stream()
        comm = self.__dict__.get("_comm_stream")
        if comm is None:
            comm = self.__dict__["_comm_stream"] = torch.cuda.Stream(device=self.flat_g.device)
        comm.wait_stream(main)
        for s in sides:
            comm.wait_stream(s)
        with torch.cuda.stream(comm):
            t0 = self._comm_mark(comm)
            self._mid_work = torch.distributed.all_reduce(self.flat_g[self.n_mid:self.n_split], op=torch.distributed.ReduceOp.SUM,
                                                          group=self.pg, async_op=True)
            self._comm_mark(comm, "encoder_late", t0)
        E.refresh_stream()

    def _comm_mark(self, stream, tag=None, start=None):
        """bench.py: `time_collectives = True` brackets every gradient all-reduce of the step with events on the stream it
        runs on (`comm_times`: [(bucket, start event, end event)]), so the line can show what each bucket costs INSIDE the
        step - under the encoder backward / after the join - not only stand-alone."""
        if not getattr(self, "time_collectives", False):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        if tag is not None:
            self.__dict__.setdefault("comm_times", []).append((tag, start, ev))
        return ev

    def all_reduce_grads(self):
        if self.collectives:
            main = torch.cuda.current_stream()
            t0 = self._comm_mark(main)
            if self._early_work is not None:
                rest = self.n_mid if self._mid_work is not None else self.n_split
                torch.distributed.all_reduce(self.flat_g[:rest], op=torch.distributed.ReduceOp.SUM, group=self.pg)
                self._comm_mark(main, "encoder" if self._mid_work is None else "encoder_early", t0)
                self._early_work.wait()                   # main stream waits for the overlapped part(s)
                if self._mid_work is not None:
                    self._mid_work.wait()
                main.wait_stream(self._comm_stream)
                self._early_work = None
                self._mid_work = None
            else:
                torch.distributed.all_reduce(self.flat_g, op=torch.distributed.ReduceOp.SUM, group=self.pg)
                self._comm_mark(main, "whole_gradient", t0)

    def _stage_hyper(self):
        """Host -> pinned -> device copies of the per-step scalars (enqueued on the current stream)."""
        t = self.step_count
        slot = t % len(self._stage)
        hyper_host, seed_host, ev = self._stage[slot]
        if self._stage_used[slot]:
            ev.synchronize()                              # the H2D copies of step t-4 have read this slot
        hyper_host[0] = self.slow_lr * self.lr_factor
        hyper_host[1] = self.lr * self.lr_factor
        hyper_host[2] = 1.0 - self.betas[0] ** t
        hyper_host[3] = (1.0 - self.betas[1] ** t) ** 0.5
        seed_host[0] = 0x5DEECE66D * t + 11
        self._hyper_dev.copy_(hyper_host, non_blocking=True)
        self._seed_dev.copy_(seed_host, non_blocking=True)
        ev.record()
        self._stage_used[slot] = True

    def optimizer_step(self, use_device_hyper: bool = False):
        L = _lib.lib()
        hyper = self._hyper_dev.data_ptr() if use_device_hyper else None
        if self.optimizer == "sgd":
            rc = L.pp_sgd_step_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.n, self.n_split,
                                    self.slow_lr * self.lr_factor, self.lr * self.lr_factor, self.momentum, self.wd,
                                    max(self.step_count, 1), 1.0 / self.world, hyper, _lib.current_stream_ptr())
            E.refresh_stream()
            _lib.check(rc, "pp_sgd_step_flat")
            return
        rc = L.pp_adam_step_flat(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                 self.exp_avg_sq.data_ptr(), self.n, self.n_split, self.slow_lr * self.lr_factor,
                                 self.lr * self.lr_factor, self.betas[0], self.betas[1], self.eps, self.wd,
                                 max(self.step_count, 1), 1.0 / self.world, hyper, _lib.current_stream_ptr())
        E.refresh_stream()
        _lib.check(rc, "pp_adam_step_flat")

    def _step_body(self, x, y, keep_logits, device_hyper):
        E.begin_step()
        loss = self.forward_backward(x, y, keep_logits)
        if self.collectives:
            _lib.plan_note(self.all_reduce_grads)          # (a host break of a native plan: only where there is something to exchange)
        self.optimizer_step(device_hyper)
        E.end_step()
        return loss

    def train_step(self, x: torch.Tensor, y: torch.Tensor, keep_logits: bool = False) -> torch.Tensor:
        """One optimisation step (model.py:101-122).  After enable_replay() the recorded launch list is re-issued."""
        self._ensure_train_mode()
        self.step_count += 1
        if self._plan is not None:
            if tuple(x.shape) != tuple(self._gx.shape) or tuple(y.shape) != tuple(self._gy.shape):
                raise ValueError(f"replayed train_step needs the recorded shapes {tuple(self._gx.shape)} / {tuple(self._gy.shape)}, "
                                 f"got {tuple(x.shape)} / {tuple(y.shape)} (disable_replay() first)")
            self._gx.copy_(x, non_blocking=True)              # copy_ converts strides / dtype into the recorded layout
            self._gy.copy_(y, non_blocking=True)
            self._stage_hyper()
            if torch.cuda.current_stream().cuda_stream != self._plan_stream:
                raise RuntimeError("train_step after enable_replay() must run on the stream the plan was recorded on")
            self._plan.replay()
            E.end_step()               # (the replay does not run begin_step / end_step: the recorded step's planes die here as well)
            return self.last_loss
        return self._step_body(x, y, keep_logits, False)

    def enable_replay(self, x: torch.Tensor, y: torch.Tensor, warmup: int = 1):
        """Record the launches of one step for this input shape (~415 C-ABI calls, ~110 stream fork / join operations, the
        all-reduces at N > 1) as a _lib.LaunchPlan and re-issue them from a tight loop in every later train_step(): the
        Python around each launch (5.4 ms per 6.85 ms step) is paid once.  The GPU schedule is the eager one - main stream,
        weight-gradient stream, all-reduce under the encoder backward - which a captured hipGraph does not keep
        (profiles/r02_graph_replay.txt; the hipGraph variant measured there - 7.8-8.0 vs 6.8 ms eager - was removed in round 3).
        The step runs on private copies of x / y, per-step scalars
        (learning rates, Adam bias corrections, dropout seed) are read from device memory, and the recorded step allocates
        from a private memory pool that is kept, so every address in the plan stays valid and is never handed to another
        tensor.  `warmup` eager steps (real optimisation steps, as is the recorded one) first grow the scratch buffers."""
        assert self._plan is None
        self.model.train()
        E.set_dropout_device_seed(self._seed_dev)
        # the private copies have the layout the recorded launches read: torch-side conversions inside the step
        # (x.contiguous(), target.to(int64)) run once at record time and are NOT part of the plan
        self._gx, self._gy = x.contiguous().clone(), y.to(torch.int64).contiguous().clone()
        for _ in range(warmup):
            self.step_count += 1
            self._stage_hyper()
            self._step_body(self._gx, self._gy, True, True)
        self.step_count += 1
        self._stage_hyper()
        pool = torch.cuda.MemPool()
        with torch.cuda.use_mem_pool(pool, device=x.device):
            with _lib.record_plan(native=NATIVE_PLAN) as plan:
                self._step_body(self._gx, self._gy, True, True)
        self._plan, self._plan_pool = plan, pool
        self._plan_stream = torch.cuda.current_stream(x.device).cuda_stream
        return self

    def sync_buffers(self, src: int = 0):
        """Data-parallel runs keep per-GPU BatchNorm statistics while training (the reference has no SyncBN); before the
        replicas are used for validation / acquisition their running statistics are made identical by broadcasting rank
        `src`'s buffers (what DistributedDataParallel(broadcast_buffers=True) does every forward), so that a sharded
        acquisition round scores every image with the same network."""
        if self.world <= 1:
            return
        bufs = [b for b in self.model.buffers() if b.dtype.is_floating_point]
        if not bufs:
            return
        flat = torch.cat([b.reshape(-1) for b in bufs])
        torch.distributed.broadcast(flat, src=torch.distributed.get_global_rank(self.pg, src) if self.pg is not None else src,
                                    group=self.pg)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()

    def disable_replay(self):
        """Back to eager steps; also removes the process-wide device seed word enable_replay() installed, so that later
        eager trainers / MC-dropout forwards draw their masks from the host counter again."""
        if self._plan is not None:
            self.last_loss = self.last_logits = None      # they live in the plan's memory pool
            # the plan holds bound methods of this trainer (all_reduce_grads, _early_all_reduce_on): clear it so that no
            # trainer <-> plan reference cycle keeps the memory pool (a full step of activations) alive until a GC pass
            if hasattr(self._plan, "close"):
                self._plan.close()
            self._plan.calls.clear()
            self._plan = self._plan_pool = None
            self._gx = self._gy = None
        if E._dropout_seed_dev[0] is self._seed_dev:
            E.set_dropout_device_seed(None)

    def close(self):
        """Give back what this trainer changed process-wide: the recorded plan, the device seed word, and - when it is the last
        collective trainer - the library's comm-CU reserve (back to the value in force before the first one set it)."""
        self.disable_replay()
        if self.__dict__.get("_holds_reserve"):
            self._holds_reserve = False
            _RESERVE["users"] -= 1
            if _RESERVE["users"] == 0 and _RESERVE["before"] is not None:
                if _lib.lib().pp_get_comm_cu_reserve() != _RESERVE["before"]:
                    _lib.lib().pp_set_comm_cu_reserve(_RESERVE["before"])
                _RESERVE["before"] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_poly_lr(self, T: int, N: int, power: float = 0.9):
        """utils/lr_scheduler.py:15-17 applied to both segments."""
        self.lr_factor = pow((1 - 1.0 * T / N), power)
