import torch

from vagnmt_hip import ops


class ImageRetrievalRankingLoss(torch.nn.Module):
    """sum_{i!=j} max(0, m - S_jj + S_ij), S = im s^T  (losses/ImageRetrievalRankingLoss.py:4-21)."""

    def __init__(self, margin=1.0):
        super(ImageRetrievalRankingLoss, self).__init__()
        self.margin = margin

    def forward(self, im, s):
        return ops.RankLoss.apply(im, s, self.margin, 1)
