"""A/B of the bf16x6 product kernels: the single-stage kernel against the variant behind a library option (default gemm_pp: the
ping-pong kernel) on the step's product shapes (whole vag_gemm_f32 calls replayed from a graph, hot operands), with a bitwise
comparison of the results (the six products enter every accumulator in the same order in both kernels).
Usage: python tools/exp_gemm_ab.py [option name]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import _lib as L

OPT = sys.argv[1] if len(sys.argv) > 1 else "gemm_pp"
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
    ("dx = dgi W_ih", 2560, 256, 1536, True, False, 0),
    ("4096^3 NT", 4096, 4096, 4096, True, True, 0),
    ("4096^3 TN", 4096, 4096, 4096, False, False, 0),
]
print("%-22s %28s %10s %10s %7s %s" % ("product", "shape", "baseline", OPT, "ratio", "bitwise"))
for name, M, N, K, a_kc, b_kc, beta in SHAPES:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A = torch.randn((M, K) if a_kc else (K, lda), device=dev)
    Bm = torch.randn((N, K) if b_kc else (K, ldb), device=dev)
    ldc = (N + 3) // 4 * 4
    sa = (K, 1) if a_kc else (1, lda)
    sb = (1, K) if b_kc else (ldb, 1)
    res, ts = [], []
    for w4 in (0, 1):
        L.set_option(OPT, w4)
        Cm = torch.zeros(M, ldc, device=dev)
        fn = lambda: L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta),
                            L.ptr(Cm), ldc, None, 0, L.stream())
        fn()
        torch.cuda.synchronize()
        res.append(Cm.clone())
        ts.append(bench._time_graph(fn, reps=10))
    L.set_option(OPT, 0)
    fl = 2.0 * M * N * K
    same = torch.equal(res[0], res[1])
    err = (res[0] - res[1]).abs().max().item() / max(res[0].abs().max().item(), 1e-30)
    print("%-22s M=%5d N=%5d K=%5d  %6.1f us %5.0f TF  %6.1f us %5.0f TF  %5.2f  %s" %
          (name, M, N, K, ts[0] * 1e6, fl / ts[0] / 1e12, ts[1] * 1e6, fl / ts[1] / 1e12, ts[0] / ts[1],
           "equal" if same else "max rel diff %.1e (split-K atomics reorder sums)" % err), flush=True)
