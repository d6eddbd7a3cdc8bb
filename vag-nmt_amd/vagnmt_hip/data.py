"""Device-resident batch pipeline: the MI355X-native counterpart of the reference's
``preprocessing.data_generator_tl_mtv`` (preprocessing.py:308-384).

The reference pads python lists, reorders image rows one by one and copies three tensors to the device for every batch.
Multi30K is tiny against 288 GB of HBM (29k pairs x 80 tokens x 8 B = 19 MB, 2048-d features 238 MB), so the whole
tokenised corpus and the feature matrix are uploaded ONCE; a batch is then one small index upload plus three row gathers
on the device.  Batch composition and ordering follow the reference exactly: target-length buckets
(``BucketBatchSampler``), rows sorted by source length, descending, with numpy's argsort-then-reverse tie order
(:354-356), padding with 0 to the batch maximum, tuple layout ``(x, y, im, x_lengths_sorted, y_lengths_sorted)``.
Data parallel: every rank walks the same (common-seed) batch order and keeps batches rank, rank+world, ..."""
import numpy as np
import torch

from machine_translation_vision.samplers import BucketBatchSampler

from ._lib import call, ptr, stream


class LengthList(list):
    """The reference's ``x_lengths_sorted`` (a Python list, preprocessing.py:384) that also carries the same numbers as an int32
    device tensor (``.device_tensor``, uploaded together with the batch's row indices): TrainStep.step takes that instead of
    uploading the list again from pageable memory -- a copy the host would wait on behind all queued GPU work."""
    device_tensor = None


class DeviceCorpus:
    RING = 16                 # pinned upload slots: a slot is rewritten only after RING further batches (its copy has long run)

    def __init__(self, data_pairs, data_im, device):
        n = len(data_pairs)
        self.x_len = np.array([len(p[0]) for p in data_pairs], dtype=np.int64)
        self.y_len = np.array([len(p[1]) for p in data_pairs], dtype=np.int64)
        lx, ly = int(self.x_len.max()), int(self.y_len.max())
        x = np.zeros((n, lx), dtype=np.int64)
        y = np.zeros((n, ly), dtype=np.int64)
        for i, (sx, sy) in enumerate(data_pairs):
            x[i, :len(sx)] = sx
            y[i, :len(sy)] = sy
        self.device = device
        self.x = torch.from_numpy(x).to(device)
        self.y = torch.from_numpy(y).to(device)
        self.im = torch.as_tensor(np.asarray(data_im), dtype=torch.float32).contiguous().to(device) if data_im is not None else None
        self._ring, self._slot, self._events = None, 0, None

    def _upload(self, idx, x_len_sorted):
        """Row indices (int64) and source lengths (int32) of a batch to the device in ONE asynchronous copy from a pinned slot:
        the host does not wait for the GPU (a pageable copy is stream-ordered behind every queued kernel and blocks the host until
        the device gets there: one drained pipeline per batch)."""
        b = len(idx)
        is_cuda = torch.device(self.device).type == "cuda"
        if not is_cuda:
            dev_idx = torch.from_numpy(np.ascontiguousarray(idx)).to(self.device)
            return dev_idx, torch.tensor(x_len_sorted, dtype=torch.int32, device=self.device)
        cap = 4 * max(b, 64)                   # (rows of the ring start 8-byte aligned: the int64 view below)
        if self._ring is None or self._ring.shape[1] < 3 * b:
            self._ring = torch.empty(self.RING, cap, dtype=torch.int32).pin_memory()
            self._events = [None] * self.RING
            self._slot = 0
        k = self._slot
        self._slot = (k + 1) % self.RING
        if self._events[k] is not None:
            self._events[k].synchronize()          # (RING batches ago: long done)
        host = self._ring[k]
        host[:2 * b].view(torch.int64).copy_(torch.from_numpy(np.ascontiguousarray(idx)))
        host[2 * b:3 * b].copy_(torch.from_numpy(np.asarray(x_len_sorted, dtype=np.int32)))
        dev = host[:3 * b].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[k] = ev
        return dev[:2 * b].view(torch.int64), dev[2 * b:3 * b]

    def batch(self, bidx):
        """Assemble one batch from sample indices (any order); returns the reference's tuple."""
        bidx = np.asarray(bidx)
        xl = self.x_len[bidx]
        order = np.argsort(xl)[::-1]                       # preprocessing.py:354-356: argsort ascending, then reversed
        idx = bidx[order]
        x_len_sorted = LengthList(int(v) for v in self.x_len[idx])
        y_len_sorted = [int(v) for v in self.y_len[idx]]
        wx, wy = max(x_len_sorted), max(y_len_sorted)
        b = len(idx)
        dev_idx, x_len_sorted.device_tensor = self._upload(idx, x_len_sorted)
        bx = torch.empty(b, wx, dtype=torch.int64, device=self.device)
        by = torch.empty(b, wy, dtype=torch.int64, device=self.device)
        s = stream()
        call("vag_gather_rows_i64", ptr(self.x, torch.int64), self.x.shape[1], ptr(dev_idx, torch.int64), b, wx,
             ptr(bx, torch.int64), s)
        call("vag_gather_rows_i64", ptr(self.y, torch.int64), self.y.shape[1], ptr(dev_idx, torch.int64), b, wy,
             ptr(by, torch.int64), s)
        bim = None
        if self.im is not None:
            bim = torch.empty(b, self.im.shape[1], dtype=torch.float32, device=self.device)
            call("vag_embed_fwd", ptr(dev_idx, torch.int64), b, ptr(self.im), self.im.shape[1], ptr(bim), s)
        return bx, by, bim, x_len_sorted, y_len_sorted


def shard_batches(batches, rank, world_size):
    """Rank r takes batches r, r+world, ... of the common order, truncated to a multiple of world_size so that every
    rank runs the same number of optimiser steps (each step ends in a collective; an extra batch on some ranks would
    leave them waiting in it forever).  At most world_size-1 batches per epoch are skipped; the per-epoch reshuffle of
    the sampler rotates which ones."""
    n = len(batches) // world_size * world_size
    return [batches[i] for i in range(rank, n, world_size)]


def data_generator_tl_mtv(corpus, batch_size, rank=0, world_size=1, seed=None):
    """Same contract as preprocessing.data_generator_tl_mtv, over a DeviceCorpus; batches of equal target length.
    Data parallel: pass the same ``seed`` (e.g. base_seed + epoch) on every rank -- the sampler shuffles with numpy's
    global generator, which is seeded here for the duration of the shuffle only and restored afterwards."""
    if world_size > 1 and seed is None:
        raise ValueError("data-parallel batch streams need a common seed (every rank must walk the same batch order)")
    state = None
    if seed is not None:
        state = np.random.get_state()
        np.random.seed(seed)
    try:
        batches = [np.asarray(b) for b in BucketBatchSampler(corpus.y_len, batch_size)]
    finally:
        if state is not None:
            np.random.set_state(state)
    for bidx in shard_batches(batches, rank, world_size):
        yield corpus.batch(bidx)
