"""Length-bucketed batch sampler, drop-in for samplers/bucket.py of the reference.

Every yielded batch holds sample indices whose (target) lengths are equal; the last batch of a bucket may be smaller
than ``batch_size``.  Same constructor, ``__iter__``/``__len__`` contract and the same consumption of ``numpy.random``
as the reference (one permutation per bucket in insertion order, then one permutation of the bucket-visit list), so a
given ``numpy`` seed yields the same batches (tests/test_host_logic.py)."""
import numpy as np


class BucketBatchSampler(object):
    def __init__(self, lengths, batch_size, max_len=None):
        self.batch_size = batch_size
        self.max_len = 10000 if max_len is None else max_len
        order = {}                                   # length -> indices, buckets in first-seen order (samplers/bucket.py:44-47)
        for idx, n in enumerate(lengths):
            n = int(n)
            if n <= self.max_len:
                order.setdefault(n, []).append(idx)
        self.buckets = {n: np.asarray(v) for n, v in order.items()}
        self.bucket_names = list(self.buckets.keys())
        visits = []
        for n, members in self.buckets.items():
            visits.extend([n] * (-(-members.size // self.batch_size)))      # ceil(size / batch_size) visits (:59-60)
        self.bucket_idxs = np.asarray(visits)
        self.n_batches = len(self.bucket_idxs)

    def __iter__(self):
        offsets = {n: 0 for n in self.buckets}
        views = {n: np.random.permutation(len(m)) for n, m in self.buckets.items()}      # (:79-82)
        for n in np.random.permutation(self.bucket_idxs):                                # (:86)
            n = int(n)
            take = views[n][offsets[n]: offsets[n] + self.batch_size]
            offsets[n] += len(take)
            yield self.buckets[n][take]

    def __len__(self):
        return self.n_batches
