"""Host-side mirror of the reference's `query.py` call surface on top of the HIP acquisition kernels.

Same class / function names, argument meaning, return formats and error behaviour as the reference
(`/root/reference/query.py`; line numbers below refer to it), so a driver written against the
reference (`model.py:44,83`, `train.py:9,82`, `datasets/*.py` codecs) runs unchanged.  Differences:

  * softmax -> score -> exclusion -> top-k run fused in ONE HIP kernel pass over the logits
    (csrc/acq.hip via the C ABI) instead of five ATen passes (query.py:190-204,57-61);
  * top-k ties break towards the lower flat index and NaN scores (0*log 0, query.py:230) sort
    first — the reference inherits an implementation-defined order from torch.topk;
  * the MC-dropout branch (query.py:177-187, NameError `up_map` in the reference) is implemented as
    the evident intent: mean over `mc_n_steps` stochastic passes;
  * for models that expose `forward_lowres` (DeepLab) the x4 bilinear upsample of deeplab.py:55-56 is folded into the
    scoring kernel as well (pp_acq_lowres_score_topk): identical queries, no full-size logits;
  * there is no CPU fallback: tensors must live on the GPU and the extension must be built.
"""
import os
import pickle as pkl
from math import ceil
from pathlib import Path
from typing import Dict, List, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import acquisition as acq
from . import dist_utils

_LARGEST_STRATEGIES = ("entropy", "least_confidence")
# PIXELPICK_FUSED_LOWRES=0: always materialise the full-resolution logits (model(x)["pred"]) before scoring
FUSED_LOWRES = os.environ.get("PIXELPICK_FUSED_LOWRES", "1") != "0"
# PIXELPICK_QUERY_PIPELINE=0: finish every batch of an acquisition round (index / entropy read-back, masks, statistics)
# before the next one is loaded, instead of doing that host work while the GPU already runs the next batch
QUERY_PIPELINE = os.environ.get("PIXELPICK_QUERY_PIPELINE", "1") != "0"


class QuerySelector:
    """query.py:12-221."""

    def __init__(self, args, dataloader, device=torch.device("cuda:0")):
        # same attribute set as query.py:14-31
        self.dataset_name = args.dataset_name
        self.dataloader = dataloader
        self.debug = args.debug
        self.device = device
        self.dir_checkpoints = f"{args.dir_root}/checkpoints/{args.experim_name}"
        self.ignore_index = args.ignore_index
        self.mc_n_steps = args.mc_n_steps
        self.n_classes = args.n_classes
        self.n_pixels_by_us = args.n_pixels_by_us
        self.network_name = args.network_name
        self.query_stats = QueryStats(args)
        self.query_strategy = args.query_strategy
        self.reverse_order = args.reverse_order
        self.stride_total = args.stride_total
        self.top_n_percent = args.top_n_percent
        self.uncertainty_sampler = UncertaintySampler(args.query_strategy)
        self.use_mc_dropout = args.use_mc_dropout
        self.vote_type = args.vote_type
        # not in the reference: images of equal size are forwarded `query_batch_size` at a time (the reference's
        # query loader has batch_size 1, model.py:36-37; in eval mode the per-image results do not depend on the batch)
        # Default 8: identical picks (tested), 385 -> ~1500 images/s for an acquisition round on MI355X (tools/query_bench.py)
        self.query_batch_size = int(getattr(args, "query_batch_size", 8))
        self.mc_chunk = int(getattr(args, "mc_chunk", 32))     # stochastic passes per forward in the MC-dropout branch

    # ------------------------------------------------------------------ selection (query.py:33-69)
    @property
    def _largest(self) -> bool:
        return self.query_strategy in _LARGEST_STRATEGIES

    def _k_topk(self, h: int, w: int) -> int:
        return int(h * w * self.top_n_percent) if self.top_n_percent > 0. else self.n_pixels_by_us

    def _reverse_order_sampling_mask(self, h: int, w: int) -> np.ndarray:
        """query.py:38-42: a random candidate set of k = 5 % pixels (numpy global RNG, as the reference)."""
        assert self.top_n_percent > 0.
        k = self._k_topk(h, w)
        ind = np.random.choice(range(h * w), k, False)
        sampling_mask = np.zeros(h * w, dtype=np.bool_)
        sampling_mask[ind] = True
        return sampling_mask

    def _choose(self, ind_sorted: np.ndarray) -> np.ndarray:
        """query.py:63-64: optional random subsample (numpy global RNG) of the value-sorted top-k -> flat indices."""
        if (not self.reverse_order) and self.top_n_percent > 0.:
            ind_sorted = np.random.choice(ind_sorted, self.n_pixels_by_us, False)
        return ind_sorted

    def _finish_selection(self, ind_sorted: np.ndarray, h: int, w: int) -> np.ndarray:
        """query.py:63-68: optional random subsample of the value-sorted top-k, then bool mask."""
        query = np.zeros(h * w, dtype=np.bool_)
        query[self._choose(ind_sorted)] = True
        return query.reshape(h, w)

    def _select_queries(self, uc_map) -> np.ndarray:
        """uc_map [h,w] (torch tensor, any device, or numpy) -> bool [h,w].  Top-k on the GPU
        (pp_topk_select); RNG steps on the host with numpy's global state exactly as the reference."""
        uc = torch.as_tensor(uc_map)
        h, w = uc.shape[-2:]
        uc = uc.reshape(1, h * w).to(self.device, torch.float32)
        if self.reverse_order:
            sampling_mask = torch.from_numpy(self._reverse_order_sampling_mask(h, w)).to(self.device)
            uc = uc.clone()
            uc[0, ~sampling_mask] = 0. if self._largest else 1.0
            k = self.n_pixels_by_us
        else:
            k = self._k_topk(h, w)
        idx, _ = acq.topk_select(uc, k, self._largest)
        return self._finish_selection(idx[0].cpu().numpy().astype(np.int64), h, w)

    def _select_from_logits(self, logits: torch.Tensor, exclude: np.ndarray) -> np.ndarray:
        """Fused path used by __call__: logits [1,C,h,w] on the GPU, exclude bool [h,w] (host)."""
        h, w = logits.shape[-2:]
        if self.reverse_order:
            exclude = exclude | ~self._reverse_order_sampling_mask(h, w).reshape(h, w)
            k = self.n_pixels_by_us
        else:
            k = self._k_topk(h, w)
        idx, _, _ = acq.score_topk(logits, torch.from_numpy(np.ascontiguousarray(exclude))[None],
                                   self.query_strategy, k)
        return self._finish_selection(idx[0].cpu().numpy().astype(np.int64), h, w)

    def _random_topk(self, rmaps: np.ndarray, exclude: np.ndarray, k: int) -> np.ndarray:
        """`random` strategy (args.py:27; query.py:242-244,195-201,57-61): rmaps [n,h,w] are the host `torch.rand` maps the
        reference's UncertaintySampler._random draws (a CPU tensor there as well), exclude bool [n,h,w] (already or-ed with
        the complement of the reverse-order candidate set when that mode is on).  Excluded pixels are filled with 1.0 and
        the k SMALLEST values win; the selection runs on the GPU (pp_topk_select, ties towards the lower flat index).
        -> value-sorted flat indices int64 [n,k]."""
        n, h, w = rmaps.shape
        rmaps = np.where(exclude, np.float32(1.0), rmaps.astype(np.float32, copy=False))
        idx, _ = acq.topk_select(torch.from_numpy(np.ascontiguousarray(rmaps.reshape(n, h * w))).to(self.device), k, False)
        return idx.cpu().numpy().astype(np.int64)

    # ------------------------------------------------------------------ codecs (query.py:71-142)
    @staticmethod
    def encode_query(p_img: str, size: Tuple[int, int], query: np.ndarray) -> Dict[str, dict]:
        y_coords, x_coords = np.nonzero(query)  # row-major order, int64 (query.py:77)
        return {p_img: {"height": size[0], "width": size[1], "x_coords": x_coords, "y_coords": y_coords}}

    @staticmethod
    def decode_queries(encoded_query: Dict[str, dict], ignore_index: int = 255, return_as_dict: bool = False
                       ) -> Union[List[np.ndarray], Dict[str, np.ndarray]]:
        def decode_one(info: dict) -> np.ndarray:
            ys = np.asarray(info["y_coords"], dtype=np.int64)
            xs = np.asarray(info["x_coords"], dtype=np.int64)
            labels = info.get("category_id", None)
            if labels is None:
                out = np.zeros((info["height"], info["width"]), dtype=np.bool_)
                out[ys, xs] = True
            else:
                out = np.full((info["height"], info["width"]), ignore_index, dtype=np.int64)
                n = min(len(ys), len(labels))  # zip() semantics of query.py:93,101
                out[ys[:n], xs[:n]] = np.asarray(labels, dtype=np.int64)[:n]
            return out

        if len(encoded_query) == 0:
            raise ValueError(len(encoded_query))  # query.py:129
        items = sorted(encoded_query.items()) if len(encoded_query) > 1 else list(encoded_query.items())
        if return_as_dict:
            return {p_img: decode_one(info) for p_img, info in items}
        return [decode_one(info) for _, info in items]

    # ------------------------------------------------------------------ the loop (query.py:144-221)
    def _forward_logits(self, model, x, h, w) -> torch.Tensor:
        return model(x)["pred"][:, :, :h, :w]

    def _dist(self):
        """(rank, world, group) of the acquisition round.  `self.process_group` (None = the default group) is used when
        torch.distributed is initialised; otherwise the round is single-rank."""
        g = getattr(self, "process_group", None)
        rank, world = dist_utils.rank_world(g)
        return rank, world, g

    def _draw(self, h: int, w: int) -> dict:
        """Every host random number ONE image consumes, drawn when the loop reaches the image - the reference's order
        (query.py:190 sampler -> :40 / :64 numpy choice).  Drawing per image, not per flushed batch, keeps the torch / numpy
        global streams in the reference's sequence whatever the batching, and lets a rank of a sharded round advance the
        streams past images it does not own."""
        d = {}
        if self.query_strategy == "random":
            # UncertaintySampler._random (query.py:242-244): one CPU torch.rand((1,h,w)) per image - per stochastic pass in
            # the MC-dropout branch (query.py:183-185 sums them; their mean is used)
            n_draws = self.mc_n_steps if self.use_mc_dropout else 1
            rmap = torch.rand((1, h, w))[0].numpy()
            for _ in range(n_draws - 1):
                rmap = rmap + torch.rand((1, h, w))[0].numpy()
            d["rmap"] = rmap / np.float32(n_draws) if n_draws > 1 else rmap
        if self.reverse_order:
            d["cand"] = self._reverse_order_sampling_mask(h, w).reshape(h, w)                 # query.py:40
        elif self.top_n_percent > 0.:
            # query.py:63-64 `np.random.choice(ind, n_pixels_by_us, False)`: numpy draws permutation(len)[:n] and indexes the
            # array with it, so drawing the POSITIONS consumes the same random numbers and picks the same pixels
            # (tests/test_host_logic.py) - and tells which of the read-back entropies belong to them
            d["pos"] = np.random.choice(self._k_topk(h, w), self.n_pixels_by_us, False)
        return d

    def _k_launch(self, h: int, w: int) -> int:
        return self.n_pixels_by_us if self.reverse_order else self._k_topk(h, w)

    def __call__(self, nth_query, model, human_labels: bool = False):
        """One acquisition round (query.py:144-221).  With torch.distributed initialised the images are sharded
        `i -> rank i mod world`: every rank forwards / scores only its own images, ONE all_gather of the
        per-image picks (and statistics contributions) follows, and every rank ends with the same `dict_queries`, the same
        QueryStats and the same `dataset.label_queries` side effect as a single-rank round; rank 0 writes query_stats.pkl."""
        dataset = self.dataloader.dataset
        prev_queries = dataset.list_labelled_queries if human_labels else dataset.queries
        rank, world, group = self._dist()

        model.eval()
        if self.use_mc_dropout:
            model.turn_on_dropout()

        print(f"Choosing pixels by {self.query_strategy}")
        is_random = self.query_strategy == "random"
        records = []      # (image index, p_img, h, w, sorted flat indices int64, statistics contribution | None)
        y = None
        n_seen = 0

        pending = []      # _Item: x [1,3,H,W] on device, y numpy | None, exclude bool [h,w], p_img, (h, w), draws, index
        inflight = []     # at most one launched-but-unfinished batch of the pipelined path
        want_any_stats = not human_labels
        # Low-resolution scoring, no reverse-order sampling: the GPU work of a batch is only ENQUEUED by flush(); its results
        # come back through pinned buffers and are turned into masks / statistics after the next batch has been loaded and
        # enqueued, so host and GPU overlap.
        pipelined = (QUERY_PIPELINE and FUSED_LOWRES and not self.use_mc_dropout and hasattr(model, "forward_lowres")
                     and not self.reverse_order and not is_random and torch.device(self.device).type == "cuda")
        # DeepLab: x4 align_corners=True (deeplab.py:55-56); FPNSeg: x2 align_corners=False (decoders.py:101)
        lowres_align = bool(getattr(model, "LOWRES_ALIGN_CORNERS", True))
        copy_stream = self.__dict__.get("_copy_stream")
        if pipelined and copy_stream is None:
            copy_stream = self.__dict__["_copy_stream"] = torch.cuda.Stream(device=self.device)

        def emit(it, cand, cand_ent):
            """cand: the chosen flat indices of image `it` (any order), cand_ent: their entropies in the same order | None."""
            h, w = it.size
            order = np.argsort(cand, kind="stable")
            sel = np.asarray(cand, dtype=np.int64)[order]
            contrib = None
            if cand_ent is not None:
                contrib = QueryStats.contribution(it.y, np.asarray(cand_ent)[order].tolist(), (sel // w, sel % w))
            records.append((it.index, it.p_img, h, w, sel, contrib))

        def choose(it, idx_sorted, ent_sorted=None):
            """query.py:63-64 on the value-sorted top-k of one image -> (flat indices, their entropies | None)."""
            pos = it.draws.get("pos")
            if pos is not None:
                return idx_sorted[pos], (ent_sorted[pos] if ent_sorted is not None else None)
            return idx_sorted, ent_sorted

        def finish(hd):
            """Host half of a pipelined batch: wait for ITS read-back (not for whatever the GPU runs now)."""
            items, idx_host, ent_host, ev = hd
            ev.synchronize()
            idx_h = idx_host.numpy().astype(np.int64)
            ent_h = ent_host.numpy() if ent_host is not None else None
            for j, it in enumerate(items):
                emit(it, *choose(it, idx_h[j], ent_h[j] if ent_h is not None else None))

        def launch_pipelined():
            (h, w), = {it.size for it in pending}
            main = torch.cuda.current_stream(self.device)
            excl = np.ascontiguousarray(np.stack([it.exclude for it in pending]))
            with torch.cuda.stream(copy_stream):               # a pageable upload on the main stream would block the host
                excl_dev = torch.from_numpy(excl).view(torch.uint8).to(self.device)    # behind everything enqueued there
            main.wait_stream(copy_stream)                      # images and exclusion masks of this batch have arrived
            excl_dev.record_stream(main)
            xs = torch.cat([it.x for it in pending], dim=0)
            for it in pending:
                it.x.record_stream(main)
            low, full_size = model.forward_lowres(xs)
            k = self._k_topk(h, w)
            idx, _, _ = acq.score_topk_lowres(low, full_size, excl_dev, self.query_strategy, k, crop=(h, w), align_corners=lowres_align)
            n = len(pending)
            idx_host = torch.empty((n, k), dtype=torch.int32, pin_memory=True)
            idx_host.copy_(idx, non_blocking=True)
            ent_host = None
            if want_any_stats and all(it.y is not None for it in pending):
                img = torch.arange(n, device=idx.device, dtype=torch.int32).repeat_interleave(k)
                ent = acq.score_at_lowres(low, full_size, img, idx.reshape(-1), "entropy", crop=(h, w), align_corners=lowres_align)
                ent_host = torch.empty((n, k), dtype=torch.float32, pin_memory=True)
                ent_host.copy_(ent.reshape(n, k), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(main)
            handle = (list(pending), idx_host, ent_host, ev)
            pending.clear()
            if inflight:
                finish(inflight.pop())                         # the PREVIOUS batch, while the GPU works on this one
            inflight.append(handle)

        def flush():
            if not pending:
                return
            if pipelined and len({it.size for it in pending}) == 1:
                launch_pipelined()
                return
            if inflight:
                finish(inflight.pop())
            if pipelined:                                      # images were uploaded (and voc-padded) on copy_stream
                main = torch.cuda.current_stream(self.device)
                main.wait_stream(copy_stream)
                for it in pending:
                    it.x.record_stream(main)
            xs = torch.cat([it.x for it in pending], dim=0)
            logits_b = None
            sizes = {it.size for it in pending}
            # DeepLab exposes its classifier output in front of the x4 upsample; the scoring
            # kernel interpolates on the fly and the full-resolution logits are never written
            fused = (FUSED_LOWRES and not self.use_mc_dropout and len(sizes) == 1 and hasattr(model, "forward_lowres"))
            if fused:
                low, full_size = model.forward_lowres(xs)
            elif not self.use_mc_dropout:
                logits_b = model(xs)["pred"]
            if not self.use_mc_dropout and len(sizes) == 1:
                # one scoring launch, one index read-back and one entropy read-back for the whole batch
                (h, w), = sizes
                excl = np.stack([it.exclude for it in pending])
                if not fused:
                    lg = logits_b[:, :, :h, :w]
                if self.reverse_order:
                    excl = excl | ~np.stack([it.draws["cand"] for it in pending])
                if is_random:
                    # the forward above only feeds the statistics (the reference runs it too, query.py:190)
                    idx_h = self._random_topk(np.stack([it.draws["rmap"] for it in pending]), excl, self._k_launch(h, w))
                elif fused:
                    idx, _, _ = acq.score_topk_lowres(low, full_size, torch.from_numpy(excl), self.query_strategy,
                                                      self._k_launch(h, w), crop=(h, w), align_corners=lowres_align)
                    idx_h = idx.cpu().numpy().astype(np.int64)
                else:
                    idx, _, _ = acq.score_topk(lg, torch.from_numpy(excl), self.query_strategy, self._k_launch(h, w))
                    idx_h = idx.cpu().numpy().astype(np.int64)
                chosen = [choose(it, idx_h[j])[0] for j, it in enumerate(pending)]
                want_stats = (not human_labels) and all(it.y is not None for it in pending)
                ent_all = None
                if want_stats:
                    flat = np.concatenate(chosen)
                    img = np.repeat(np.arange(len(pending)), [len(c) for c in chosen])
                    if fused:
                        ent_all = acq.score_at_lowres(low, full_size, img, flat, "entropy", crop=(h, w), align_corners=lowres_align).cpu().numpy()
                    else:
                        dev = lg.device
                        picked = lg[torch.from_numpy(img).to(dev), :, torch.from_numpy(flat // w).to(dev), torch.from_numpy(flat % w).to(dev)]
                        ent_all = acq.score_map(picked.t().reshape(1, picked.shape[1], 1, -1).contiguous(), None, "entropy").reshape(-1).cpu().numpy()
                off = 0
                for j, it in enumerate(pending):
                    n_j = len(chosen[j])
                    emit(it, chosen[j], ent_all[off:off + n_j] if want_stats else None)
                    off += n_j
                pending.clear()
                return
            for j, it in enumerate(pending):
                h, w = it.size
                excl_j = it.exclude | ~it.draws["cand"] if self.reverse_order else it.exclude
                if self.use_mc_dropout:
                    # mean uncertainty / mean probability over mc_n_steps stochastic passes: the passes differ only in their
                    # dropout masks (eval-mode BatchNorm is per sample), so they run as ONE forward over mc_n_steps copies of
                    # the image instead of mc_n_steps launch-bound forwards at batch 1 (`mc_chunk` copies at a time)
                    uc_map = torch.empty((h, w), dtype=torch.float32, device=self.device)
                    prob = torch.empty((1, self.n_classes, h, w), dtype=torch.float32, device=self.device)
                    left, first = self.mc_n_steps, True
                    while left > 0:
                        t = min(left, self.mc_chunk)
                        logits = self._forward_logits(model, it.x.expand(t, -1, -1, -1).contiguous(), h, w)
                        # uc_map += score(softmax(logits)) / n ; prob += softmax(logits) / n   (query.py:181-187), one HIP pass
                        acq.mc_accumulate_(logits, prob[0], None if is_random else uc_map, "entropy" if is_random else self.query_strategy,
                                           1.0 / self.mc_n_steps, accumulate=not first)
                        left -= t
                        first = False
                    if is_random:                       # the drawn map is already the mean of the mc_n_steps host draws
                        idx_sorted = self._random_topk(it.draws["rmap"][None], excl_j[None], self._k_launch(h, w))[0]
                    else:
                        uc_map[torch.from_numpy(excl_j).to(self.device)] = 0.0 if self._largest else 1.0
                        idx_t, _ = acq.topk_select(uc_map.reshape(1, h * w), self._k_launch(h, w), self._largest)
                        idx_sorted = idx_t[0].cpu().numpy().astype(np.int64)
                    cand = choose(it, idx_sorted)[0]
                    ent = None
                    if not human_labels and it.y is not None:
                        qmask = np.zeros(h * w, dtype=np.bool_)
                        qmask[cand] = True
                        ent_rowmajor = QueryStats._get_entropy(qmask.reshape(h, w), prob)     # row-major = ascending flat index
                        ent = np.empty(len(cand), dtype=np.float64)
                        ent[np.argsort(cand, kind="stable")] = ent_rowmajor
                    emit(it, cand, ent)
                else:
                    logits = logits_b[j:j + 1, :, :h, :w]
                    if is_random:
                        idx_sorted = self._random_topk(it.draws["rmap"][None], excl_j[None], self._k_launch(h, w))[0]
                    else:
                        idx_t, _, _ = acq.score_topk(logits, torch.from_numpy(np.ascontiguousarray(excl_j))[None],
                                                     self.query_strategy, self._k_launch(h, w))
                        idx_sorted = idx_t[0].cpu().numpy().astype(np.int64)
                    cand = choose(it, idx_sorted)[0]
                    ent = None
                    if not human_labels and it.y is not None:
                        qmask = np.zeros(h * w, dtype=np.bool_)
                        qmask[cand] = True
                        ent_rowmajor = QueryStats._get_entropy_from_logits(qmask.reshape(h, w), logits)
                        ent = np.empty(len(cand), dtype=np.float64)
                        ent[np.argsort(cand, kind="stable")] = ent_rowmajor
                    emit(it, cand, ent)
            pending.clear()

        # Host side of the shard: a rank's loader only reads and collates ITS images (index-level shard,
        # dist_utils.ShardedBatchSampler).  The host RNG streams must still advance through EVERY image in loader order (random
        # strategy, reverse-order candidates, top-5 % sub-sample are drawn per image); for images of other ranks that only needs
        # (h, w), which the dataset provides as metadata (`image_sizes` / `image_size(i)`).  A dataset without it, in a mode that
        # draws, falls back to enumerating the whole loader and skipping.
        needs_rng = is_random or self.reverse_order or self.top_n_percent > 0.
        sharded, sizes = None, None
        if world > 1 and getattr(self.dataloader, "batch_size", None) == 1:
            sizes = dist_utils.dataset_image_sizes(dataset)
            if sizes is not None or not needs_rng:
                sharded = dist_utils.shard_dataloader(self.dataloader, rank, world, equal_steps=False)

        def indexed_batches():
            if sharded is None:
                yield from enumerate(self.dataloader)
            else:
                yield from zip([b[0] for b in sharded.batch_sampler], sharded)

        next_draw = 0
        with torch.no_grad():
            for batch_ind, dict_data in indexed_batches():
                h, w = dict_data['x'].shape[2:]
                if sharded is not None:
                    if needs_rng:
                        for j in range(next_draw, batch_ind):  # other ranks' images: advance the streams from metadata only
                            self._draw(*sizes[j])
                    next_draw = batch_ind + 1
                else:
                    n_seen += 1
                draws = self._draw(h, w)                       # every rank advances the host RNG streams for every image
                y = dict_data.get('y', None)
                if dist_utils.owner_rank(batch_ind, world) != rank:
                    continue                                   # another rank's image (i -> rank i mod W)
                if pipelined and not dict_data['x'].is_cuda:
                    with torch.cuda.stream(copy_stream):       # the upload must not queue behind the previous batch's kernels
                        x = dict_data['x'].to(self.device, non_blocking=False)
                else:
                    x = dict_data['x'].to(self.device)
                mask = np.asarray(prev_queries[batch_ind])  # h x w

                exclude = (mask != self.ignore_index) if human_labels else mask.astype(np.bool_)
                y_np = None
                if y is not None:
                    y_np = y.squeeze(dim=0).numpy()  # h x w
                    exclude = exclude | (y_np == self.ignore_index)

                if self.dataset_name == "voc":  # query.py:171-174
                    pad_h = ceil(h / self.stride_total) * self.stride_total - h
                    pad_w = ceil(w / self.stride_total) * self.stride_total - w
                    if pipelined and x.is_cuda:
                        # x was uploaded on copy_stream: pad it THERE, so that the tensor kept in `pending` is the one the
                        # batch reads and the un-padded upload is released on the stream that owns it (padding on the
                        # main stream would free it for the next upload while the pad kernel is still queued)
                        with torch.cuda.stream(copy_stream):
                            x = F.pad(x, pad=(0, pad_w, 0, pad_h), mode='reflect')
                    else:
                        x = F.pad(x, pad=(0, pad_w, 0, pad_h), mode='reflect')

                if pending and (pending[0].x.shape != x.shape or len(pending) >= self.query_batch_size):
                    flush()
                pending.append(_Item(x, y_np, exclude, dict_data["p_img"][0], (h, w), draws, batch_ind))
                if len(pending) >= self.query_batch_size:
                    flush()
            flush()
            if inflight:
                finish(inflight.pop())
        if sharded is not None:
            n_seen = len(dataset)
            if needs_rng:
                for j in range(next_draw, n_seen):             # trailing images of other ranks: leave the streams where a
                    self._draw(*sizes[j])                      # single-rank round would leave them

        records = dist_utils.gather_records(records, world, group)     # ~100 B per image, the only exchange of the round; loader order
        assert len(records) == n_seen and n_seen > 0, f"no queries are chosen!" if n_seen == 0 else (len(records), n_seen)
        dict_queries: dict = dict()
        n_pixels = 0
        for _, p_img, h, w, sel, contrib in records:
            # encode_query(p_img, size, query) without re-scanning a mask: the sorted flat indices ARE np.nonzero order
            dict_queries.update({p_img: {"height": h, "width": w, "x_coords": sel % w, "y_coords": sel // w}})
            n_pixels += len(sel)
            if contrib is not None:
                self.query_stats.apply(contrib)
        # the branch (it holds a barrier) is decided from the gathered records, identical on every rank - a rank with an empty shard
        # has seen no `y` of its own
        has_labels = any(r[5] is not None for r in records) or (world == 1 and y is not None)
        if not human_labels and has_labels:
            if rank == 0:
                self.query_stats.save(nth_query)
            print(f"{n_pixels} labelled pixels  are chosen by {self.query_strategy} strategy")
            # updates labels for the query dataloader only (query.py:219-220).  The reference's datasets dump queries.pkl when
            # nth_query is an int (base_dataset.py:43-45): in a sharded round only rank 0 writes, the others merge in memory
            dataset.label_queries(dict_queries, nth_query if rank == 0 else None)
            if world > 1:
                torch.distributed.barrier(group=group)
        return dict_queries


class _Item:
    __slots__ = ("x", "y", "exclude", "p_img", "size", "draws", "index")

    def __init__(self, x, y, exclude, p_img, size, draws, index):
        self.x, self.y, self.exclude, self.p_img, self.size, self.draws, self.index = x, y, exclude, p_img, size, draws, index


class UncertaintySampler:
    """query.py:224-247.  `prob` is an already-softmaxed [b,c,h,w] GPU tensor -> [b,h,w]."""

    def __init__(self, query_strategy):
        self.query_strategy = query_strategy

    @staticmethod
    def _entropy(prob):
        return acq.uncertainty_from_prob(prob, "entropy")

    @staticmethod
    def _least_confidence(prob):
        return acq.uncertainty_from_prob(prob, "least_confidence")

    @staticmethod
    def _margin_sampling(prob):
        return acq.uncertainty_from_prob(prob, "margin_sampling")

    @staticmethod
    def _random(prob):
        b, _, h, w = prob.shape
        return torch.rand((b, h, w))  # CPU tensor, as query.py:244

    def __call__(self, prob):
        return getattr(self, f"_{self.query_strategy}")(prob)


class QueryStats:
    """query.py:250-308: label histogram, entropy at the picked pixels, #unique labels, mean pairwise distance."""

    def __init__(self, args):
        self.dir_checkpoints = f"{args.dir_root}/checkpoints/{args.experim_name}"
        self.list_entropy, self.list_n_unique_labels, self.list_spatial_coverage = list(), list(), list()
        self.dict_label_cnt = {l: 0 for l in range(args.n_classes)}

    def _count_labels(self, query, y, coords=None):
        for l in (y[coords] if coords is not None else y.flatten()[query.flatten()]):
            self.dict_label_cnt[l] += 1

    @staticmethod
    def _get_entropy(query, prob):
        """Entropy of `prob` ([1,C,h,w], GPU) at the queried pixels, row-major order (query.py:261-264)."""
        ys, xs = np.nonzero(query)
        picked = prob[0][:, torch.from_numpy(ys).to(prob.device), torch.from_numpy(xs).to(prob.device)]  # C x n
        ent = acq.uncertainty_from_prob(picked.reshape(1, picked.shape[0], 1, -1).contiguous(), "entropy")
        return ent.reshape(-1).cpu().numpy().tolist()

    @staticmethod
    def _get_entropy_from_logits(query, logits):
        """Same quantity from the LOGITS at the picked pixels only: no full-size prob, no full-map D2H."""
        ys, xs = np.nonzero(query)
        picked = logits[0][:, torch.from_numpy(ys).to(logits.device), torch.from_numpy(xs).to(logits.device)]
        ent = acq.score_map(picked.reshape(1, picked.shape[0], 1, -1).contiguous(), None, "entropy")
        return ent.reshape(-1).cpu().numpy().tolist()

    @staticmethod
    def _n_unique_labels(query, y, coords=None):
        return len(set(y[coords] if coords is not None else y.flatten()[query.flatten()]))

    @staticmethod
    def _spatial_coverage(query, coords=None):
        a, b = coords if coords is not None else np.nonzero(query)
        n = a.shape[0]
        if n < 2:
            return np.nan
        pts = np.stack([a, b], axis=1).astype(np.float64)
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        return d[~np.eye(n, dtype=np.bool_)].reshape(n, -1).mean()

    def save(self, nth_query):
        dict_stats = {
            "label_distribution": self.dict_label_cnt,
            "avg_entropy": np.mean(self.list_entropy),
            "avg_n_unique_labels": np.mean(self.list_n_unique_labels),
            "avg_spatial_coverage": np.mean(self.list_spatial_coverage)
        }
        for k, v in dict_stats.items():
            print(f"{k}: {v}")
        os.makedirs(f"{self.dir_checkpoints}/{nth_query}_query", exist_ok=True)
        with open(f"{self.dir_checkpoints}/{nth_query}_query/query_stats.pkl", "wb") as f:
            pkl.dump(dict_stats, f)

    @staticmethod
    def contribution(y, pixel_entropy, coords) -> tuple:
        """What ONE image adds to the statistics (query.py:262-289), as plain python data so that a sharded round can
        ship it: (labels at the picked pixels in row-major order, their entropies, #unique labels, mean pairwise distance)."""
        labels = [int(l) for l in y[coords]]
        return (labels, list(pixel_entropy), len(set(labels)), QueryStats._spatial_coverage(None, coords))

    def apply(self, contrib):
        labels, pixel_entropy, n_unique, coverage = contrib
        for l in labels:
            self.dict_label_cnt[l] += 1
        self.list_entropy.extend(pixel_entropy)
        self.list_n_unique_labels.append(n_unique)
        self.list_spatial_coverage.append(coverage)

    def update_from_picked(self, query, y, pixel_entropy, coords=None):
        """Host-only part of update(): everything except the entropy evaluation.  coords = (rows, cols) of the picked
        pixels in row-major order, when the caller already has them (saves three scans of the mask)."""
        self._count_labels(query, y, coords)
        self.list_entropy.extend(pixel_entropy)
        self.list_n_unique_labels.append(self._n_unique_labels(query, y, coords))
        self.list_spatial_coverage.append(self._spatial_coverage(query, coords))

    def update(self, query, y, prob):
        self.update_from_picked(query, y, self._get_entropy(query, prob))

    def update_from_logits(self, query, y, logits):
        self.update_from_picked(query, y, self._get_entropy_from_logits(query, logits))


def gather_previous_query_files(dir_base: str, ext="pkl") -> List[str]:
    """query.py:311-313."""
    pattern = f"*/queries.{ext}" if ext is not None else "*"
    return [str(p) for p in Path(dir_base).rglob(pattern)]


def merge_previous_query_files(list_previous_query_files: List[str], ignore_index: int, verbose: bool = True
                               ) -> Dict[str, np.ndarray]:
    """query.py:316-351: per image, overlay the label maps of all rounds (later files win)."""
    per_image: Dict[str, List[np.ndarray]] = dict()
    for p_file in list_previous_query_files:
        with open(p_file, "rb") as f:
            encoded = pkl.load(f)
        decoded = QuerySelector.decode_queries(encoded, ignore_index=ignore_index, return_as_dict=True)
        for img_path, q in decoded.items():
            per_image.setdefault(img_path, []).append(q)

    cnt = 0
    merged_all: Dict[str, np.ndarray] = dict()
    for p_img, maps in per_image.items():
        merged = np.full_like(maps[0], ignore_index, dtype=np.int64)
        for q in maps:
            sel = q != ignore_index
            merged[sel] = q[sel]
            cnt += sel.sum()
        merged_all[p_img] = merged
    if verbose:
        print(f"# merged pixels: {cnt}")
    return merged_all
