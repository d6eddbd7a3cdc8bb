"""GPU parity of the dense-product kernels behind the C ABI (vag_gemm_f32 / vag_linear_*), against fp64 numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from vagnmt_hip import _lib
    return _lib


def _gemm(M, N, K, a_kc, b_kc, alpha=1.0, beta=0.0, bias=False, act=0, seed=0):
    L = _lib()
    rs = np.random.RandomState(seed)
    A = rs.randn(M, K).astype(np.float32)
    Bm = rs.randn(K, N).astype(np.float32)
    C0 = rs.randn(M, N).astype(np.float32)
    bv = rs.randn(N).astype(np.float32) if bias else None
    dev = "cuda:0"
    # memory layouts: A k-contiguous (M,K) or m-contiguous (stored as (K,M)); B k-contiguous = stored (N,K)
    At = torch.from_numpy(A if a_kc else np.ascontiguousarray(A.T)).to(dev)
    Bt = torch.from_numpy(np.ascontiguousarray(Bm.T) if b_kc else Bm).to(dev)
    Ct = torch.from_numpy(C0.copy()).to(dev)
    bt = torch.from_numpy(bv).to(dev) if bias else None
    sam, sak = (K, 1) if a_kc else (1, M)
    sbk, sbn = (1, K) if b_kc else (N, 1)
    L.call("vag_gemm_f32", M, N, K, alpha, L.ptr(At), sam, sak, L.ptr(Bt), sbk, sbn, beta, L.ptr(Ct), N,
           L.ptr(bt), act, L.stream())
    ref = alpha * (A.astype(np.float64) @ Bm.astype(np.float64)) + beta * C0
    if bias:
        ref = ref + bv
    if act:
        ref = np.tanh(ref)
    got = Ct.cpu().numpy()
    scale = np.abs(ref).max() + 1e-6
    err = np.abs(got - ref).max() / scale
    assert err < 2e-5, (M, N, K, a_kc, b_kc, err)


@pytest.mark.parametrize("a_kc", [True, False])
@pytest.mark.parametrize("b_kc", [True, False])
def test_gemm_layouts(a_kc, b_kc):
    for (M, N, K) in [(64, 64, 16), (128, 128, 64), (100, 77, 50), (257, 130, 33), (5, 9, 7), (1, 1, 1)]:
        _gemm(M, N, K, a_kc, b_kc)
    _gemm(300, 520, 96, a_kc, b_kc, alpha=0.5, beta=1.0, bias=True)
    _gemm(96, 40, 64, a_kc, b_kc, bias=True, act=1)


def test_gemm_big_tile_and_splitk():
    _gemm(2560, 1024, 256, True, True, bias=True)                 # 128x128 tiles
    _gemm(1536, 512, 2560, False, False, beta=1.0)                # weight-gradient shape: split-K atomics
    _gemm(9391, 256, 640, False, False, beta=1.0)                 # odd M
    _gemm(640, 256, 9391, True, False)                            # odd K (d tmid = dlogits W_out)
    _gemm(640, 9391, 256, True, True, bias=True)                  # head logits


@pytest.mark.parametrize("M", [1, 5, 16, 64, 100, 128])
def test_linear_small_m(M):
    L = _lib()
    rs = np.random.RandomState(M)
    for (N, K, act) in [(48, 24, 0), (1536, 512, 0), (512, 1024, 1), (60, 16, 0), (1024, 2560, 0), (9391, 256, 0)]:
        x = rs.randn(M, K).astype(np.float32)
        W = (rs.randn(N, K) / np.sqrt(K)).astype(np.float32)
        b = rs.randn(N).astype(np.float32)
        xt, Wt, bt = [torch.from_numpy(v).cuda() for v in (x, W, b)]
        y = torch.empty(M, N, device="cuda")
        L.call("vag_linear_fwd", M, N, K, L.ptr(xt), L.ptr(Wt), L.ptr(bt), act, L.ptr(y), L.stream())
        ref = x.astype(np.float64) @ W.astype(np.float64).T + b
        if act:
            ref = np.tanh(ref)
        err = np.abs(y.cpu().numpy() - ref).max() / (np.abs(ref).max() + 1e-6)
        assert err < 2e-5, (M, N, K, err)


def test_linear_bwd():
    L = _lib()
    rs = np.random.RandomState(3)
    M, N, K = 37, 52, 44
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32)).double().requires_grad_(True)
    W = torch.from_numpy(rs.randn(N, K).astype(np.float32)).double().requires_grad_(True)
    b = torch.from_numpy(rs.randn(N).astype(np.float32)).double().requires_grad_(True)
    y = torch.tanh(x @ W.t() + b)
    dy = torch.from_numpy(rs.randn(M, N).astype(np.float32)).double()
    y.backward(dy)
    xt, Wt, yt, dyt = [v.detach().float().cuda() for v in (x, W, y, dy)]
    dx = torch.empty(M, K, device="cuda")
    gW = torch.zeros(N, K, device="cuda")
    gb = torch.zeros(N, device="cuda")
    L.call("vag_linear_bwd", M, N, K, L.ptr(xt), L.ptr(Wt), L.ptr(yt), L.ptr(dyt), 1, L.ptr(dx), 0, L.ptr(gW), L.ptr(gb),
           L.stream())
    for got, want in ((dx, x.grad), (gW, W.grad), (gb, b.grad)):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert err < 2e-5, err


@pytest.mark.parametrize("M,H", [(64, 512), (5, 24), (16, 128), (33, 40)])
def test_gru_cell_fwd_bwd_match_torch_autograd(M, H):
    """vag_gru_cell_fwd / vag_gru_cell_bwd (one recurrence step each way) against torch.nn.GRUCell + autograd, fp32
    tolerance 1e-4 of the largest entry: two chained steps, so that the backward of the first step receives its
    hidden-state gradient through the recurrent projection of the second (the fused form the sequence kernels use)."""
    L = _lib()
    torch.manual_seed(M * 131 + H)
    cell = torch.nn.GRUCell(H, H)
    x1, x2, h0 = torch.randn(M, H), torch.randn(M, H), torch.randn(M, H, requires_grad=True)
    h1 = cell(x1, h0)
    h2 = cell(x2, h1)
    d_out1, d_out2 = torch.randn(M, H), torch.randn(M, H)
    (h1 * d_out1).sum().add((h2 * d_out2).sum()).backward()
    dev = "cuda:0"
    W_ih, W_hh, b_ih, b_hh = [p.detach().to(dev) for p in (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    gi1 = (x1.to(dev) @ W_ih.t() + b_ih).contiguous()
    gi2 = (x2.to(dev) @ W_ih.t() + b_ih).contiguous()
    h0d = h0.detach().to(dev)
    h1d, h2d = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    s1, s2 = torch.empty(4, M, H, device=dev), torch.empty(4, M, H, device=dev)
    L.call("vag_gru_cell_fwd", L.ptr(gi1), L.ptr(h0d), L.ptr(W_hh), L.ptr(b_hh), M, H, L.ptr(h1d), L.ptr(s1), L.stream())
    L.call("vag_gru_cell_fwd", L.ptr(gi2), L.ptr(h1d), L.ptr(W_hh), L.ptr(b_hh), M, H, L.ptr(h2d), L.ptr(s2), L.stream())
    assert (h1d.cpu() - h1.detach()).abs().max() < 1e-4 and (h2d.cpu() - h2.detach()).abs().max() < 1e-4
    # last step: no later step -> zero recurrent contribution (a zero dgh_next), no carry
    WT = W_hh.t().contiguous()                       # (H, 3H)
    zero = torch.zeros(M, 3 * H, device=dev)
    dgi2, dgh2, c2 = torch.empty(M, 3 * H, device=dev), torch.empty(M, 3 * H, device=dev), torch.empty(M, H, device=dev)
    L.call("vag_gru_cell_bwd", L.ptr(zero), L.ptr(WT), None, L.ptr(d_out2.to(dev)), L.ptr(s2), L.ptr(h1d), M, H,
           L.ptr(dgi2), L.ptr(dgh2), L.ptr(c2), L.stream())
    dgi1, dgh1, c1 = torch.empty(M, 3 * H, device=dev), torch.empty(M, 3 * H, device=dev), torch.empty(M, H, device=dev)
    L.call("vag_gru_cell_bwd", L.ptr(dgh2), L.ptr(WT), L.ptr(c2), L.ptr(d_out1.to(dev)), L.ptr(s1), L.ptr(h0d), M, H,
           L.ptr(dgi1), L.ptr(dgh1), L.ptr(c1), L.stream())
    # parameter gradients follow from the per-step gate gradients; d h0 = dgh1 W_hh + z1 * dh1
    got = {"weight_ih": dgi1.t() @ x1.to(dev) + dgi2.t() @ x2.to(dev), "weight_hh": dgh1.t() @ h0d + dgh2.t() @ h1d,
           "bias_ih": dgi1.sum(0) + dgi2.sum(0), "bias_hh": dgh1.sum(0) + dgh2.sum(0)}
    for n, g in got.items():
        ref = getattr(cell, n).grad
        assert (g.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item()), n
    d_h0 = dgh1 @ W_hh + c1
    assert (d_h0.cpu() - h0.grad).abs().max() <= 1e-4 * max(1.0, h0.grad.abs().max().item())
