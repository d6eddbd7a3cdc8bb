import torch

from vagnmt_hip import ops


class PairwiseRankingLoss(torch.nn.Module):
    """sum_{i!=j} max(0, m - S_jj + S_ij) + max(0, m - S_ii + S_ij), S = im s^T.
    Mirrors losses/PairwiseRankingLoss.py:4-24 of the reference (same constructor and forward signature);
    the BxB similarity product, both hinges, the diagonal mask and the reduction run in one HIP launch pair
    instead of 2B single-element index writes."""

    def __init__(self, margin=1.0):
        super(PairwiseRankingLoss, self).__init__()
        self.margin = margin

    def forward(self, im, s):
        return ops.RankLoss.apply(im, s, self.margin, 0)
