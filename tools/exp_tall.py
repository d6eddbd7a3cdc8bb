import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
for (M, N, K) in ((16, 4096, 256), (16, 9391, 256), (16, 18782, 256), (16, 40000, 256), (192, 4096, 256), (192, 9391, 256), (192, 18782, 256), (192, 40000, 256), (192, 9391, 64), (192, 9391, 512)):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    yd = torch.empty(M, N, device=dev)
    def many():                     # 20 launches per graph: a single short kernel per replay only measures the replay floor (~10 us)
        for _ in range(20):
            L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), L.ptr(b), 0, L.ptr(yd), L.stream())
    res = []
    for f32 in (0, 1):
        L.set_option("gemm_f32mfma", f32)
        res.append(bench._time_graph(many, reps=10) / 20)
    L.set_option("gemm_f32mfma", 0)
    t = res[0]
    print("M=%3d N=%5d K=%3d: tall-skinny %.1f us  (W %.1f MB, out %.1f MB -> %.2f TB/s) | round-3 path (gemm_f32mfma=1) %.1f us" % (M, N, K, t * 1e6, N * K * 4 / 1e6, M * N * 4 / 1e6, (N * K * 4 + M * N * 4) / t / 1e12, res[1] * 1e6), flush=True)
