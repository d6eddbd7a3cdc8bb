"""Split-K sweep of the step's single (ungrouped) accumulating products: are they bound by their fp32 atomics?  (d out.weight: 9.6 MB of
output x 12 k-slices = 115 MB of atomic adds; MI355X_MICROARCH.md prices float atomics at ~1.3 TB/s chip-wide.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
SHAPES = [("d out.weight", 9391, 256, 2560, False, False, 1), ("d tmid", 2560, 256, 9391, True, False, 0),
          ("wp = W_ih2 W_c2h", 1536, 1024, 512, True, False, 0), ("g txt_w", 512, 1024, 64, False, False, 1)]
for name, M, N, K, a_kc, b_kc, beta in SHAPES:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A = torch.randn((M, K) if a_kc else (K, lda), device=dev)
    Bm = torch.randn((N, K) if b_kc else (K, ldb), device=dev)
    ldc = (N + 3) // 4 * 4
    Cm = torch.zeros(M, ldc, device=dev)
    sa = (K, 1) if a_kc else (1, lda)
    sb = (1, K) if b_kc else (ldb, 1)
    row = []
    for sk in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        L.set_option("gemm_force_tile", 128 if sk else 0); L.set_option("gemm_force_splitk", sk)
        def many():
            for _ in range(10):
                L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta), L.ptr(Cm), ldc, None, 0, L.stream())
        try:
            t = bench._time_graph(many, reps=5) / 10
            row.append("%s %.1f" % ("model" if sk == 0 else "sk%d" % sk, t * 1e6))
        except Exception:
            row.append("sk%d -" % sk)
    L.set_option("gemm_force_tile", 0); L.set_option("gemm_force_splitk", 0)
    print("%-18s M=%5d N=%5d K=%5d (us): %s" % (name, M, N, K, " | ".join(row)), flush=True)
