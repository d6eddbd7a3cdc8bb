"""Image / caption retrieval metrics, drop-in for utils/im_retrieval_eval.py of the reference (t2i, i2t).

The reference ranks one query at a time (``torch.mm`` + ``torch.sort`` + ``.cpu()`` per query, :15-24); here the N x N
score matrix is one product on the matrix pipes and the ranks one kernel (``vag_retrieval_ranks``); only the N integer
ranks come back to the host for the recall / median statistics (:25-30)."""
import numpy as np
import torch

from vagnmt_hip._lib import call, ptr, stream


def _ranks(queries, keys):
    queries, keys = queries.contiguous().float(), keys.contiguous().float()
    n, s = queries.shape
    scores = torch.empty(n, n, dtype=torch.float32, device=queries.device)
    ranks = torch.empty(n, dtype=torch.int32, device=queries.device)
    call("vag_retrieval_ranks", ptr(queries), ptr(keys), n, s, ptr(scores), ptr(ranks, torch.int32), stream())
    return ranks.cpu().numpy().astype(np.float64)


def _metrics(ranks):
    """Recall@{1,5,10} in percent and the 1-based median rank (utils/im_retrieval_eval.py:25-30)."""
    recall = [100.0 * int(np.count_nonzero(ranks < k)) / len(ranks) for k in (1, 5, 10)]
    return (recall[0], recall[1], recall[2], float(np.floor(np.median(ranks))) + 1)


def t2i(images, captions):
    """Text -> Image.  images, captions: (N,K) embedding matrices.  Returns (R@1, R@5, R@10, median rank)."""
    return _metrics(_ranks(captions, images))


def i2t(images, captions):
    """Image -> Text."""
    return _metrics(_ranks(images, captions))
