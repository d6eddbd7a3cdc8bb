"""Round 6: vag_linear_fwd (skinny kernels up to M = 256 rows, bf16x6 tiles beyond) against vag_gemm_f32 at batch-sized M -- the products
of the visual-grounding branch and the initial state at B = 64 .. 256 (VSE_Imagine_Enc.py:123, V11.py:118)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
for name, N, K in (("image projection", 512, 2048), ("text embedding", 512, 2048), ("emb2ctx", 2048, 512), ("dec init", 1024, 2048), ("B x B similarity", 0, 512)):
    for M in (64, 96, 128, 192, 256):
        n = N if N else M
        x = torch.randn(M, K, device=dev); W = torch.randn(n, K, device=dev); y = torch.zeros(M, n, device=dev)
        f1 = lambda: L.call("vag_linear_fwd", M, n, K, L.ptr(x), L.ptr(W), None, 0, L.ptr(y), L.stream())
        f2 = lambda: L.call("vag_gemm_f32", M, n, K, 1.0, L.ptr(x), K, 1, L.ptr(W), 1, K, 0.0, L.ptr(y), n, None, 0, L.stream())
        t1, t2 = bench._time_graph(f1, reps=10), bench._time_graph(f2, reps=10)
        print("%-18s M=%3d N=%4d K=%4d   vag_linear_fwd %7.1f us   vag_gemm_f32 %7.1f us" % (name, M, n, K, t1 * 1e6, t2 * 1e6), flush=True)
