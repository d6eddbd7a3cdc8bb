"""Round 6: the step's single-launch bf16x6 products under forced tile / split-K choices (vag_set_option gemm_force_tile / gemm_force_splitk)
against the cost model's own choice.  python tools/exp_gemm_force.py > gpurun_out/exp_gemm_force.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import _lib as L

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, a_kc, b_kc, beta
    ("encwp = uk W_ih2^T", 2560, 1536, 512, True, True, 0),
    ("head logits", 2560, 9391, 256, True, True, 0),
    ("d tmid = dlogits out.weight", 2560, 256, 9391, True, False, 0),
    ("du = dgi2 W_ih2", 2560, 512, 1536, True, False, 0),
    ("attn keys", 2560, 1024, 1024, True, True, 0),
    ("d_enc += d_pe attn_e", 2560, 1024, 1024, True, False, 1),
    ("dx = d_xp W_ih", 2560, 256, 1536, True, False, 1),
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
    fl = 2.0 * M * N * K
    L.set_option("gemm_force_tile", 0); L.set_option("gemm_force_splitk", 0)
    t0 = bench._time_graph(fn, reps=10)
    row = ["model %.1f us (%.0f TF)" % (t0 * 1e6, fl / t0 / 1e12)]
    best = (t0, "model")
    for tile in (128, 64):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            if K // sp < 128:
                continue
            L.set_option("gemm_force_tile", tile); L.set_option("gemm_force_splitk", sp)
            t = bench._time_graph(fn, reps=10)
            row.append("%d/%d %.1f" % (tile, sp, t * 1e6))
            if t < best[0]:
                best = (t, "%d/%d" % (tile, sp))
    L.set_option("gemm_force_tile", 0); L.set_option("gemm_force_splitk", 0)
    print("%-30s M=%5d N=%5d K=%5d beta=%d | best %s %.1f us | %s" % (name, M, N, K, beta, best[1], best[0] * 1e6, "  ".join(row)), flush=True)
