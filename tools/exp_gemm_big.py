import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (2048, 2048, 2048), (4096, 4096, 256), (4096, 4096, 1024)]:
    for lay in ("NT", "NN", "TN"):
        if lay == "NT":
            A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); sa = (K, 1); sb = (1, K)
        elif lay == "NN":
            A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); sa = (K, 1); sb = (N, 1)
        else:
            A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); sa = (1, M); sb = (N, 1)
        C = torch.zeros(M, N, device=dev)
        def run():
            L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], 0.0, L.ptr(C), N, None, 0, L.stream())
        run(); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 5 * 1e3
        print("%dx%dx%d %s %9.1f us %6.1f TF/s" % (M, N, K, lay, us, 2.0 * M * N * K / us / 1e6), flush=True)
