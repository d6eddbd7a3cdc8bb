"""The step's product shapes through vag_gemm_f32 (whole calls replayed from a graph, hot operands)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import _lib as L

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, a_kc, b_kc, beta
    ("head logits", 2560, 9391, 256, True, True, 0),
    ("attn keys", 2560, 1024, 1024, True, True, 0),
    ("enc in-proj", 2560, 1536, 256, True, True, 0),
    ("encwp", 2560, 1536, 1024, True, True, 0),
    ("d tmid", 2560, 256, 9391, True, False, 0),
    ("d out.weight", 9391, 256, 2560, False, False, 1),
    ("g W_hh", 1536, 512, 2560, False, False, 1),
    ("g wcat", 2560, 512, 2560, False, False, 1),
    ("g attn_e", 1024, 1024, 2560, False, False, 1),
    ("d_enc += d_pe attn_e", 2560, 1024, 1024, True, False, 1),
    ("4096^3 NT", 4096, 4096, 4096, True, True, 0),
    ("4096^3 TN", 4096, 4096, 4096, False, False, 0),
    ("cfg5 logits chunk", 4096, 40000, 256, True, True, 0),
    ("cfg5 keys", 20480, 2048, 2048, True, True, 0),
]
for name, M, N, K, a_kc, b_kc, beta in SHAPES:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A = torch.randn((M, K) if a_kc else (K, lda), device=dev)
    Bm = torch.randn((N, K) if b_kc else (K, ldb), device=dev)
    ldc = (N + 3) // 4 * 4
    Cm = torch.zeros(M, ldc, device=dev)
    sa = (K, 1) if a_kc else (1, lda)
    sb = (1, K) if b_kc else (ldb, 1)
    fn = lambda: L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta),
                        L.ptr(Cm), ldc, None, 0, L.stream())
    t = bench._time_graph(fn, reps=10)
    fl = 2.0 * M * N * K
    print("%-22s M=%5d N=%5d K=%5d  %8.1f us %6.1f TF" % (name, M, N, K, t * 1e6, fl / t / 1e12), flush=True)
