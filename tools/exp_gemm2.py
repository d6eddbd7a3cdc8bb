"""[historic: the experiment builds this script switched between (VAG_GEMM_VARIANT) were removed after the
measurement in profiles/r02_exp_gemm_variants.txt; it now times the shipped kernel only]
Experiment: the cfg2 training step's dense products (shapes from the round-2 step timeline, grouped ones listed singly)
timed one by one through vag_gemm_f32 for each kernel variant (VAG_GEMM_VARIANT) -- interleaved in ONE process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
# (name, M, N, K, akc, bkc, beta)   C = A B; akc: A k-contiguous; bkc: B k-contiguous (stored (N,K))
SHAPES = [
    ("enc in-proj x2", 2560, 1536, 256, 1, 1, 0), ("pe", 2560, 1024, 1024, 1, 1, 0), ("encwp", 2560, 1536, 1024, 1, 1, 0),
    ("dec in-proj", 2560, 1536, 256, 1, 1, 0), ("head pre W2", 2560, 256, 1024, 1, 1, 1), ("logits", 2560, 9391, 256, 1, 1, 0),
    ("dt", 2560, 256, 9391, 1, 0, 0), ("d_c", 2560, 1024, 256, 1, 0, 0), ("d_enc Wp", 2560, 1024, 1536, 1, 0, 1),
    ("d_enc pe", 2560, 1024, 1024, 1, 0, 1), ("de", 2560, 256, 1536, 1, 0, 1), ("enc dx x2", 2560, 256, 1536, 1, 0, 1),
    ("g out_w", 9391, 256, 2560, 0, 0, 1), ("g W_hh x5", 1536, 512, 2560, 0, 0, 1), ("g dWp", 1536, 1024, 2560, 0, 0, 0),
    ("g W_ih x3", 1536, 256, 2560, 0, 0, 1), ("g attn_h", 1024, 512, 2560, 0, 0, 1), ("g attn_e", 1024, 1024, 2560, 0, 0, 1),
    ("g W2", 256, 1024, 2560, 0, 0, 1), ("Wp", 1536, 1024, 512, 1, 0, 0),
]
COUNT = {"enc in-proj x2": 2, "g W_hh x5": 5, "g W_ih x3": 3, "enc dx x2": 2}


def timed(run, reps=10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best


variants = sys.argv[1:] or ["0", "1", "2"]
tot = {v: 0.0 for v in variants}
for name, M, N, K, akc, bkc, beta in SHAPES:
    A = torch.randn(M, K, device=dev) if akc else torch.randn(K, M, device=dev)
    sa = (K, 1) if akc else (1, M)
    B = torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)
    sb = (1, K) if bkc else (N, 1)
    ldc = (N + 3) // 4 * 4
    C = torch.zeros(M, ldc, device=dev)

    def run():
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], float(beta), L.ptr(C), ldc, None, 0,
               L.stream())
    line = "%-16s %5dx%5dx%5d " % (name, M, N, K)
    for v in variants:
        if v == "0":
            os.environ.pop("VAG_GEMM_VARIANT", None)
        else:
            os.environ["VAG_GEMM_VARIANT"] = v
        run(); torch.cuda.synchronize()
        t = timed(run)
        tot[v] += t * COUNT.get(name, 1)
        line += "| v%s %6.1f us %6.1f TF " % (v, t, 2.0 * M * N * K / t / 1e6)
    print(line, flush=True)
os.environ.pop("VAG_GEMM_VARIANT", None)
print("sum over a step: " + "  ".join("v%s %.1f us" % (v, tot[v]) for v in variants))
