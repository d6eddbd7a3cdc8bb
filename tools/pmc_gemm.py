"""Workload for PMC passes on the split-precision GEMM kernel: a few launches of one large product per layout."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
M = N = 4096; K = 1024
for lay in ("NT", "TN"):
    if lay == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); sa = (K, 1); sb = (1, K)
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); sa = (1, M); sb = (N, 1)
    C = torch.zeros(M, N, device=dev)
    for _ in range(3):
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], 0.0, L.ptr(C), N, None, 0, L.stream())
    torch.cuda.synchronize()
print("done")
