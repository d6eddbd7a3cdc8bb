"""Rates of the ONE-plane product kernels (the 2-byte storage mode's large products: bf16 operands for gradients, fp16 forward) on
configs[4]'s shapes, through the lab build's "gemm_planes" switch; 5 launches per graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
L.use_lab_build()
dev = torch.device("cuda:0")
SHAPES = [("logits chunk", 4096, 40000, 256, True, True, 0), ("attn keys", 20480, 2048, 2048, True, True, 0),
          ("encwp", 20480, 3072, 2048, True, True, 0), ("d_enc += d_encwp Wp", 20480, 2048, 3072, True, False, 1),
          ("g W_hh1", 3072, 1024, 20480, False, False, 1), ("g wp", 3072, 2048, 20480, False, False, 1),
          ("d tmid chunk", 4096, 256, 40000, True, False, 0), ("4096^3 NT", 4096, 4096, 4096, True, True, 0),
          ("4096^3 TN", 4096, 4096, 4096, False, False, 0), ("8192^3 NT", 8192, 8192, 8192, True, True, 0)]
for name, M, N, K, a_kc, b_kc, beta in SHAPES:
    lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
    A = torch.randn((M, K) if a_kc else (K, lda), device=dev)
    Bm = torch.randn((N, K) if b_kc else (K, ldb), device=dev)
    ldc = (N + 3) // 4 * 4
    Cm = torch.zeros(M, ldc, device=dev)
    sa = (K, 1) if a_kc else (1, lda)
    sb = (1, K) if b_kc else (ldb, 1)
    row = []
    outs = {}
    for pl, big in ((3, 0), (1, 0), (1, 1), (11, 0), (11, 1)):
        L.set_option("gemm_planes", pl)
        L.set_option("gemm_big", big)
        Cm.zero_()
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta), L.ptr(Cm), ldc, None, 0, L.stream())
        torch.cuda.synchronize()
        outs[(pl, big)] = Cm.clone()
        def many():
            for _ in range(5):
                L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta), L.ptr(Cm), ldc, None, 0, L.stream())
        t = bench._time_graph(many, reps=3) / 5
        row.append("pl %2d%s %7.1f us %5.0f TF" % (pl, " big" if big else "    ", t * 1e6, 2.0 * M * N * K / t / 1e12))
    L.set_option("gemm_planes", 3); L.set_option("gemm_big", 1)
    errs = []
    for pl in (1, 11):
        ref = outs[(pl, 0)]
        errs.append("%.1e" % ((outs[(pl, 1)] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)))
    print("%-20s M=%5d N=%5d K=%5d  %s | big vs 128-tile max rel diff %s" % (name, M, N, K, " | ".join(row), ",".join(errs)), flush=True)
