"""256-tile one-plane kernel against the 128-tile one-plane kernel on small / ragged shapes, all four layouts, beta 0 / 1 (lab build)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
L.use_lab_build()
dev = torch.device("cuda:0")
bad = 0
for (M, N, K) in ((256, 512, 256), (256, 2048, 512), (512, 2048, 256), (200, 300, 96), (257, 193, 100), (1024, 256, 2048), (192, 192, 32), (300, 4097, 64), (2048, 512, 256)):
    for a_kc, b_kc, beta in itertools.product((True, False), (True, False), (0, 1)):
        lda, ldb = (M + 3) // 4 * 4, (N + 3) // 4 * 4
        lka = (K + 3) // 4 * 4
        A = torch.randn((M, lka) if a_kc else (K, lda), device=dev)
        Bm = torch.randn((N, lka) if b_kc else (K, ldb), device=dev)
        ldc = (N + 3) // 4 * 4
        C0 = torch.randn(M, ldc, device=dev)
        sa = (lka, 1) if a_kc else (1, lda)
        sb = (1, lka) if b_kc else (ldb, 1)
        outs = []
        for pl in (1, 11):
            for big in (0, 1):
                L.set_option("gemm_planes", pl); L.set_option("gemm_big", big)
                Cm = C0.clone()
                L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(Bm), sb[0], sb[1], float(beta), L.ptr(Cm), ldc, None, 0, L.stream())
                torch.cuda.synchronize()
                outs.append(Cm[:, :N].clone())
        L.set_option("gemm_planes", 3)
        for pl, (r0, r1) in ((1, (outs[0], outs[1])), (11, (outs[2], outs[3]))):
            err = (r0 - r1).abs().max().item() / max(r0.abs().max().item(), 1e-30)
            if err > 1e-5:
                bad += 1
                print("MISMATCH M=%d N=%d K=%d akc=%d bkc=%d beta=%d planes=%d: max rel diff %.3e" % (M, N, K, a_kc, b_kc, beta, pl, err), flush=True)
print("done, mismatches:", bad)
