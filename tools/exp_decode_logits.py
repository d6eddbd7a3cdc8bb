"""The decode step's logits product (M = B*k = 192 rows, V = 9391, E = 256): f32-MFMA 64x64 tiles (what the cost model picks) against
the bf16x6 128x128 kernel (gemm_force_tile), and M = 16 (greedy: the skinny kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
for M in (192, 128, 96, 64, 16):
    N, K = 9391, 256
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, (N + 3) // 4 * 4, device=dev)
    row = []
    for tile, sk in ((0, 0), (64, 1), (128, 1)):
        L.set_option("gemm_force_tile", tile); L.set_option("gemm_force_splitk", sk)
        try:
            t = bench._time_graph(lambda: L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(x), K, 1, L.ptr(W), 1, K, 0.0, L.ptr(y), y.shape[1], L.ptr(b), 0, L.stream()), reps=20)
            row.append("%s %.1f us" % ("model" if tile == 0 else "T=%d" % tile, t * 1e6))
        except Exception as e:
            row.append("T=%d failed" % tile)
    L.set_option("gemm_force_tile", 0); L.set_option("gemm_force_splitk", 0)
    yd = torch.empty(M, N, device=dev)
    for f32, name in ((0, "linear_fwd (tall-skinny bf16x6)"), (1, "linear_fwd with gemm_f32mfma=1 (round-3 path)")):
        L.set_option("gemm_f32mfma", f32)
        t = bench._time_graph(lambda: L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), L.ptr(b), 0, L.ptr(yd), L.stream()), reps=20)
        row.append("%s %.1f us" % (name, t * 1e6))
    L.set_option("gemm_f32mfma", 0)
    print("M=%3d: %s" % (M, " | ".join(row)), flush=True)
