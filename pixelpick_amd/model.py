"""Active-learning driver — the control flow of the reference's model.py:14-264 (`Model(args)()`) on the HIP path.

The reference's `Model` builds its dataloaders from `datasets/` (PIL / torchvision / cv2 file loaders — host I/O,
out of scope).  Here the three loaders are injected (any iterable yielding the reference's batch
dicts `{'x','y','queries','p_img'}` whose `.dataset` offers `.queries`, `.label_queries(dict, nth)` and
`.n_pixels_total`, base_dataset.py:24-45,182-188); `pixelpick_amd.synthetic.SyntheticDataset` is such a dataset.
Everything between the loaders is the reference's loop:

    for nth_query in range(n_stages):                         model.py:70-84
        model = self._train()            fresh model, n_epochs x (_train_epoch; _val), best-mIoU checkpoint
        queries = self.query_selector(nth_query, model)
        self.dataloader.dataset.label_queries(queries, nth_query + 1)

with the train step on `FlatTrainer` (HIP forward/backward, sparse CE, fused Adam, Poly lr per iteration) and the
per-step metrics on the device (`RunningScore.update_from_logits`).  PNG dumps (`Visualiser`) are omitted.
"""
import os
from math import ceil

import torch
import torch.nn.functional as F

from . import acquisition as acq
from . import dist_utils
from .query import QuerySelector
from .trainer import FlatTrainer
from .utils.metrics import AverageMeter, RunningScore
from .utils.utils import get_model, optimizer_spec


def write_log(fp, list_entities=None, header=None):
    """utils/utils.py:66-72 CSV logger."""
    with open(fp, "w" if header else "a") as f:
        row = header if header else list_entities
        f.write(",".join(str(e) for e in row) + "\n")


class Model:
    def __init__(self, args, dataloader, dataloader_query, dataloader_val, device=None, augmenter=None):
        # Host threads: every remaining torch CPU op in the loop (DataLoader collate = torch.stack of 11 MB per batch)
        # forks torch's intra-op pool, whose workers then spin; with the default of one thread per core (128 on the
        # MI355X hosts) the main thread that feeds the GPU is starved: measured 98 images/s through this driver vs
        # 496 with <= 8 threads (tools/driver_bench.py).  The GPU path needs no CPU parallelism; OMP_NUM_THREADS wins.
        if "OMP_NUM_THREADS" not in os.environ and torch.get_num_threads() > 8:
            torch.set_num_threads(8)
        self.args = args
        self.best_miou = -1.0
        self.dataset_name = args.dataset_name
        self.debug = args.debug
        # not in the reference: re-issue the train step's recorded launch list (native executor, csrc/plan.hip) instead of walking the
        # network in Python every step - same parameters bit for bit (tests/test_networks_gpu.py), a ragged batch falls back to an
        # eager step.  Default ON since round 4: through this loop the host is the limit of an eager step (660-720 images/s against
        # 733-735 replayed, tools/driver_bench.py).  args.replay_train_step = False or PIXELPICK_REPLAY_TRAIN=0 switch it off.
        rt = getattr(args, "replay_train_step", None)
        env = os.environ.get("PIXELPICK_REPLAY_TRAIN")
        self._replay_train = (env != "0") if env is not None else (True if rt is None else bool(rt))
        self.device = device or torch.device("cuda:0")
        self.dir_checkpoints = f"{args.dir_root}/checkpoints/{args.experim_name}"
        self.experim_name = args.experim_name
        self.ignore_index = args.ignore_index
        self.init_n_pixels = args.n_init_pixels
        self.max_budget = args.max_budget
        self.n_classes = args.n_classes
        self.n_epochs = args.n_epochs
        self.n_pixels_by_us = args.n_pixels_by_us
        self.network_name = args.network_name
        self.nth_query = -1
        self.stride_total = args.stride_total
        self.dataloader, self.dataloader_query, self.dataloader_val = dataloader, dataloader_query, dataloader_val
        self.lr_scheduler_type = args.lr_scheduler_type
        self.query_selector = QuerySelector(args, self.dataloader_query, device=self.device)
        # Data parallel (the reference is single-process): with torch.distributed initialised every rank runs this
        # same driver on its own GPU.  Train batches and validation images are taken round-robin (batch i -> rank i mod W;
        # the loaders must yield the SAME order on every rank - seed them identically - so that the shards are disjoint),
        # gradients are averaged by FlatTrainer's all-reduce, confusion matrices are summed over ranks, BatchNorm running
        # statistics are broadcast from rank 0 at the end of every epoch, the acquisition round is sharded by QuerySelector
        # and only rank 0 writes files.
        self.rank, self.world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        # Host side of the shards: every rank's loaders only enumerate (read, augment, collate) its OWN batches / images
        # (dist_utils.ShardedBatchSampler: the global batch order comes from an explicit seed + epoch, identical on every rank
        # by construction; batch i -> rank i mod W).  Loaders that are not torch DataLoaders fall back to enumerate-and-skip,
        # which needs identical RNG states on all ranks and W times the host I/O.
        self._train_loader = self._val_loader = None
        if self.world > 1:
            seed = int(getattr(args, "seed", 0))
            self._train_loader = dist_utils.shard_dataloader(dataloader, self.rank, self.world, equal_steps=True, seed=seed)
            self._val_loader = dist_utils.shard_dataloader(dataloader_val, self.rank, self.world, equal_steps=False)
        # Device data path (datasets/base_dataset.py:48-141,151-190): when an augmenter is given (or
        # args.device_augment is set) the TRAIN loader yields raw uint8 batches {'x_u8','y_u8','queries'} and the geometric /
        # photometric augmentation, to_tensor and normalisation run on the GPU; the host only draws the random parameters.
        self.augmenter = augmenter
        if self.augmenter is None and getattr(args, "device_augment", False):
            from .augment import DeviceAugmenter
            self.augmenter = DeviceAugmenter.from_args(args, device=self.device)
        if self.augmenter is not None and self.world > 1 and hasattr(self.augmenter, "use_private_rng"):
            # identical process-wide RNG streams on every rank (the acquisition round needs them) must not mean identical augmentation
            self.augmenter.use_private_rng(dist_utils.augment_seed(int(getattr(args, "seed", 0)), self.rank))
        self.on_train_batch = None        # optional callback(dict_data, x, y, mask, aug_params): tests / debugging
        self.running_loss, self.running_score = AverageMeter(), RunningScore(args.n_classes)
        self.history = []

    # ------------------------------------------------------------------ model.py:53-86
    def __call__(self):
        if self.n_pixels_by_us == 0:
            d = f"{self.dir_checkpoints}/fully_sup"
            os.makedirs(d, exist_ok=True)
            self._open_logs(d)
            self._train()
            return
        n_stages = self.max_budget // self.n_pixels_by_us
        n_stages += 1 if self.init_n_pixels > 0 else 0
        print("n_stages:", n_stages)
        for nth_query in range(n_stages):
            d = f"{self.dir_checkpoints}/{nth_query}_query"
            os.makedirs(d, exist_ok=True)
            self._open_logs(d)
            self.nth_query = nth_query
            model = self._train()
            queries = self.query_selector(nth_query, model)
            # base_dataset.py:43-45 dumps queries.pkl whenever nth_query is an int: only rank 0 may write the file, the other
            # ranks merge in memory (nth_query=None skips the dump) and wait until the file is complete
            self.dataloader.dataset.label_queries(queries, nth_query + 1 if self.rank == 0 else None)
            if self.world > 1:
                torch.distributed.barrier()
            if nth_query == n_stages - 1:
                break
        return

    def _open_logs(self, d):
        self.log_train, self.log_val = f"{d}/log_train.txt", f"{d}/log_val.txt"
        if self.rank != 0:
            return
        write_log(self.log_train, header=["epoch", "mIoU", "pixel_acc", "loss"])
        write_log(self.log_val, header=["epoch", "mIoU", "pixel_acc"])

    # ------------------------------------------------------------------ host -> device uploads off the compute stream
    def _upload(self):
        """Context in which `.to(device)` runs on a dedicated copy stream (a no-op context on a CPU-only device string)."""
        import contextlib
        if torch.device(self.device).type != "cuda":
            return contextlib.nullcontext()
        st = self.__dict__.get("_copy_stream")
        if st is None:
            st = self.__dict__["_copy_stream"] = torch.cuda.Stream(device=self.device)
        return torch.cuda.stream(st)

    def _uploaded(self, *tensors):
        """The compute stream waits for the copy stream; the uploaded tensors are marked as used by the compute stream."""
        st = self.__dict__.get("_copy_stream")
        if st is None:
            return
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(st)
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(main)

    # ------------------------------------------------------------------ model.py:88-159
    def _train_epoch(self, epoch, model, trainer, n_iters_total):
        model.train()
        miou = pixel_acc = float("nan")
        if self.lr_scheduler_type == "MultiStepLR":
            # model.py:144-145 calls MultiStepLR([20, 40], 0.1).step(epoch=epoch-1) at the END of every epoch, so the rate
            # in force DURING epoch E is base * 0.1^#{m <= E-2}: the drops take effect from epochs 22 and 42
            trainer.lr_factor = 0.1 ** sum(1 for m in (20, 40) if m <= epoch - 2)
        if self._train_loader is not None:
            loader = self._train_loader
            loader.batch_sampler.set_epoch(max(self.nth_query, 0) * self.n_epochs + epoch)   # a new shared permutation per epoch
            skip = False
            if len(loader) == 0:
                raise ValueError(f"{len(self.dataloader)} train batches for {self.world} ranks: every rank needs at least one step per epoch")
        else:
            loader = self.dataloader
            skip = self.world > 1
            if skip and len(loader) < self.world:
                raise ValueError(f"{len(loader)} train batches for {self.world} ranks: every rank needs at least one step per epoch")
        n_batches = len(self.dataloader)
        n_local = n_batches // self.world if n_batches >= self.world else 0      # equal step counts on every rank
        local_it = -1
        for it, dict_data in enumerate(loader):
            if skip:
                if it >= n_local * self.world:
                    break                                                   # ragged tail: dropped, like drop_last
                if it % self.world != self.rank:
                    continue
            local_it += 1
            mask, aug_params = None, None
            if self.augmenter is not None and 'x_u8' in dict_data:
                # raw uint8 batch: upload once, then random scale / pad / crop / flip of image, label map and query mask,
                # colour jitter / grayscale / blur, to_tensor + normalize - all on the device (csrc/augment.hip)
                def up(v):                      # stacked tensor (default collate) or a list of per-image tensors (ragged sizes)
                    return v.to(self.device) if torch.is_tensor(v) else [torch.as_tensor(t).to(self.device) for t in v]
                with self._upload():            # 1.5 MB of uint8 per batch instead of 6.3 MB of floats, off the compute stream
                    xu, yu = up(dict_data['x_u8']), up(dict_data['y_u8'])
                    qu = up(dict_data['queries']) if self.n_pixels_by_us != 0 else None
                flat = [t for v in (xu, yu, qu) if v is not None for t in (v if isinstance(v, list) else [v])]
                self._uploaded(*flat)
                out = self.augmenter(xu, yu, qu)
                x, y, mask, aug_params = out['x'], out['y'], out['queries'], out['params']
            else:
                # model.py:106-108 uploads on the compute stream: a pageable copy there queues behind the previous step's
                # kernels and blocks the host until they finish, so the host could never enqueue ahead of the GPU.  Upload on
                # a copy stream instead and let the compute stream wait for it.
                with self._upload():
                    x, y = dict_data['x'].to(self.device), dict_data['y'].to(self.device)
                    if self.n_pixels_by_us != 0:
                        mask = dict_data['queries'].to(self.device)
                self._uploaded(x, y, mask)
            if self.n_pixels_by_us != 0:                                   # model.py:108-110
                mask = mask.view(y.shape)
                # same values as `y.flatten()[~mask.flatten()] = ignore_index`, without the nonzero() + host sync
                # that boolean-index assignment performs on every step
                y = torch.where(mask != 0, y, torch.full_like(y, self.ignore_index))
            if self.on_train_batch is not None:
                self.on_train_batch(dict_data, x, y, mask, aug_params)
            if self.lr_scheduler_type == "Poly":                           # per-iteration poly decay (lr_scheduler.py:15-17)
                trainer.set_poly_lr((epoch - 1) * self._steps_per_epoch() + local_it, n_iters_total)
            if self._replay_train and trainer._plan is None and trainer.step_count >= 1:
                # static shapes (drop_last loader): record this step's launches once, re-issue them afterwards (halves the
                # host cost of a step; FlatTrainer.enable_replay).  The recorded step IS this iteration's step.  (The trainer's
                # first step runs eagerly: it is the one in which the layers that keep per-step state register themselves - the
                # bf16x3 weight planes split at begin_step, engine._X3_WPL - so that the recorded step contains that work.)
                trainer._ensure_train_mode()
                trainer.enable_replay(x, y, warmup=0)
            else:
                if trainer._plan is not None and tuple(x.shape) != tuple(trainer._gx.shape):
                    trainer.disable_replay()                                # a ragged batch: back to eager steps
                trainer.train_step(x, y, keep_logits=True)
            self.running_score.update_from_logits(y, trainer.last_logits)  # device-side confusion matrix (L6)
            self.running_loss.update(trainer.last_loss)
            if self.debug:
                break
        trainer.sync_buffers()
        self._all_reduce_scores()
        scores = self.running_score.get_scores()[0]
        miou, pixel_acc = scores['Mean IoU'], scores['Pixel Acc']
        avg_loss = float(self.running_loss.avg) if torch.is_tensor(self.running_loss.avg) else self.running_loss.avg
        if self.world > 1:                                                  # mean over the ranks' equally long shards
            t = torch.tensor([avg_loss], dtype=torch.float64)
            if torch.distributed.get_backend() == "nccl":
                t = t.to(self.device)
            torch.distributed.all_reduce(t)
            avg_loss = float(t.item()) / self.world
        if self.rank == 0:
            write_log(self.log_train, list_entities=[epoch, miou, pixel_acc, avg_loss])
        self.history.append(("train", self.nth_query, epoch, miou, pixel_acc, avg_loss))
        self._reset_meters()
        return model

    def _train(self):
        print(f"\n({self.experim_name}) training...\n")
        model = get_model(self.args).to(self.device)
        kind, slow_lr, lr, wd, momentum = optimizer_spec(self.args)          # utils/utils.py:112-306, quirks included
        trainer = FlatTrainer(model, lr=lr, slow_lr=slow_lr, weight_decay=wd, optimizer=kind, momentum=momentum,
                              ignore_index=self.ignore_index, sparse_labels=self.n_pixels_by_us != 0)
        n_total = self.n_epochs * self._steps_per_epoch()
        try:
            for e in range(1, 1 + self.n_epochs):
                self._train_epoch(e, model, trainer, n_total)
                self._val(e, model)
                if self.debug:
                    break
        finally:
            # a recorded launch plan owns a private memory pool (one step of activations) and installs a process-wide device
            # dropout seed: release both NOW, not when the cyclic GC gets to the trainer - every round builds a new trainer
            trainer.disable_replay()
        self.best_miou = -1.0
        return model

    def _steps_per_epoch(self) -> int:
        if self._train_loader is not None:
            return len(self._train_loader)
        n = len(self.dataloader)
        return n if self.world == 1 else (n // self.world if n >= self.world else 0)

    def _all_reduce_scores(self):
        """Sum the confusion matrices of the ranks' shards (exact integer counts) so that every rank reports the scores
        of the whole epoch."""
        if self.world <= 1:
            return
        self.running_score._sync()
        t = torch.from_numpy(self.running_score.confusion_matrix).to(torch.float64)
        if torch.distributed.get_backend() == "nccl":
            t = t.to(self.device)
        torch.distributed.all_reduce(t)
        self.running_score.confusion_matrix = t.cpu().numpy()

    # ------------------------------------------------------------------ model.py:175-238
    @torch.no_grad()
    def _val(self, epoch, model):
        model.eval()
        # The reference validates one image per forward (val loader batch_size 1, model.py:36-37).  In eval mode the
        # result per image does not depend on the batch, and at B=1 the forward is launch-bound (1.9 ms for ~200
        # launches), so consecutive images of equal size are forwarded `val_batch_size` at a time (default 8).
        vbs = int(getattr(self.args, "val_batch_size", 8))
        pend_x, pend_y = [], []

        def flush():
            if not pend_x:
                return
            xs, ys = torch.cat(pend_x, dim=0), torch.cat(pend_y, dim=0)
            if self.dataset_name == "voc":
                h, w = ys.shape[1:]
                pad_h = ceil(h / self.stride_total) * self.stride_total - xs.shape[2]
                pad_w = ceil(w / self.stride_total) * self.stride_total - xs.shape[3]
                xs = F.pad(xs, pad=(0, pad_w, 0, pad_h), mode='reflect')
                logits = model(xs)['pred'][:, :, :h, :w].contiguous()
            else:
                logits = model(xs)['pred']
            self.running_score.update_from_logits(ys, logits)
            pend_x.clear()
            pend_y.clear()

        val_loader = self._val_loader if self._val_loader is not None else self.dataloader_val
        for iv, dict_data in enumerate(val_loader):
            if self._val_loader is None and self.world > 1 and iv % self.world != self.rank:
                continue
            with self._upload():
                x, y = dict_data['x'].to(self.device), dict_data['y'].to(self.device)
            self._uploaded(x, y)
            if pend_x and (pend_x[0].shape[1:] != x.shape[1:] or pend_y[0].shape[1:] != y.shape[1:]
                           or sum(t.shape[0] for t in pend_x) + x.shape[0] > vbs):
                flush()
            pend_x.append(x)
            pend_y.append(y)
            if self.debug:
                break
        flush()
        self._all_reduce_scores()
        scores = self.running_score.get_scores()[0]
        miou, pixel_acc = scores['Mean IoU'], scores['Pixel Acc']
        if miou > self.best_miou:
            sub = f"{self.nth_query}_query" if self.n_pixels_by_us != 0 else "fully_sup"
            if self.rank == 0:
                torch.save({"model": model.state_dict()}, f"{self.dir_checkpoints}/{sub}/best_miou_model.pt")
            self.best_miou = miou
        if self.rank == 0:
            write_log(self.log_val, list_entities=[epoch, miou, pixel_acc])
        self.history.append(("val", self.nth_query, epoch, miou, pixel_acc))
        self._reset_meters()

    @staticmethod
    def _query(prob, query_strategy):
        """model.py:241-260: uncertainty map from probabilities (HIP)."""
        if query_strategy == "random":
            b, _, h, w = prob.shape
            return torch.rand((b, h, w))
        if query_strategy not in ("least_confidence", "margin_sampling", "entropy"):
            raise ValueError
        return acq.uncertainty_from_prob(prob, query_strategy)

    def _reset_meters(self):
        self.running_loss.reset()
        self.running_score.reset()
