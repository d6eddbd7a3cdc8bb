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
