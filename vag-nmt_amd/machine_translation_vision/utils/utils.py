"""utils/utils.py of the reference: row-wise L2 normalisation (utils/utils.py:6-10)."""
import torch

from vagnmt_hip import _lib
from vagnmt_hip._lib import ptr, stream


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        nrm = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        _lib.call("vag_l2norm_fwd", ptr(x), x.shape[0], x.shape[1], ptr(nrm), ptr(out), stream())
        ctx.save_for_backward(x, nrm, out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, nrm, out = ctx.saved_tensors
        d_out = d_out.contiguous()
        dx = torch.empty_like(x)
        _lib.call("vag_l2norm_bwd", ptr(x), ptr(nrm), ptr(out), ptr(d_out), x.shape[0], x.shape[1], ptr(dx), stream())
        return dx


def l2norm(input, p=2.0, dim=1, eps=1e-12):
    """Row-wise input / max(||input||_2, eps)  (utils/utils.py:6-10).  Only the reference's own use
    (2-D input, p=2, dim=1, eps=1e-12) is on the hot path and supported."""
    if input.dim() != 2 or p != 2.0 or dim != 1 or eps != 1e-12:
        raise NotImplementedError("l2norm: only the reference call pattern (2-D, p=2, dim=1, eps=1e-12) is implemented")
    return _L2Norm.apply(input)
