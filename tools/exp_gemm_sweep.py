"""Experiment: exhaustive (tile, split-K) sweep per product shape of a cfg2 step against the cost model's choice
(VAG_GEMM_FORCE tuning hook)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
import importlib.util
spec = importlib.util.spec_from_file_location("shapes", os.path.join(ROOT, "tools", "exp_gemm_shapes.py"))
dev = torch.device("cuda:0")
R = 2560
SHAPES = [("enc/dec in-proj", R, 1536, 256, "NT", 0), ("attn keys pe", R, 1024, 1024, "NT", 0),
          ("head W2", R, 256, 1024, "NT", 1), ("head W1", R, 256, 512, "NT", 0), ("logits", R, 9391, 256, "NT", 0),
          ("dW_out", 9391, 256, R, "TN", 1), ("d tmid", R, 256, 9391, "NN", 0), ("dW2", 256, 1024, R, "TN", 1),
          ("d_c head", R, 1024, 256, "NN", 0), ("d_h2 head", R, 512, 256, "NN", 0),
          ("dW_hh (3HxH)", 1536, 512, R, "TN", 1), ("dW_h (CxH)", 1024, 512, R, "TN", 1), ("dWp (3HxC)", 1536, 1024, R, "TN", 0),
          ("dW_ih2 chain", 1536, 512, 1024, "NT", 1), ("dW_c2h chain", 512, 1024, 1536, "TN", 1),
          ("dW_ih1 (3HxE)", 1536, 256, R, "TN", 1), ("de", R, 256, 1536, "NN", 0), ("d_enc pe", R, 1024, 1024, "NN", 1),
          ("dW_e", 1024, 1024, R, "TN", 1), ("Wp fold", 1536, 1024, 512, "NN", 0)]
def timed(run, reps=10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
tot_d = tot_b = 0.0
for name, M, N, K, lay, beta in SHAPES:
    if lay == "NT":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); sa = (K, 1); sb = (1, K)
    elif lay == "NN":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); sa = (K, 1); sb = (N, 1)
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); sa = (1, M); sb = (N, 1)
    ldc = (N + 3) // 4 * 4
    C = torch.zeros(M, ldc, device=dev)
    def run():
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], float(beta), L.ptr(C), ldc, None, 0, L.stream())
    os.environ.pop("VAG_GEMM_FORCE", None)
    run(); torch.cuda.synchronize()
    base = timed(run)
    best = (base, "model")
    res = []
    for T in (64, 128):
        if T == 128 and (M <= 64 or N <= 64):
            continue
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if sp > 1 and K // sp < 128:
                continue
            os.environ["VAG_GEMM_FORCE"] = "%d,%d" % (T, sp)
            t = timed(run)
            res.append((t, "%d/%d" % (T, sp)))
            if t < best[0]:
                best = (t, "%d/%d" % (T, sp))
    os.environ.pop("VAG_GEMM_FORCE", None)
    tot_d += base; tot_b += best[0]
    res.sort()
    print("%-16s %5dx%5dx%5d %s b%d  model %6.1f us | best %6.1f us (%s) | top: %s" % (name, M, N, K, lay, beta, base, best[0], best[1],
          " ".join("%s=%.1f" % (c, t) for t, c in res[:4])), flush=True)
print("sum model %.1f us, sum best %.1f us" % (tot_d, tot_b))
