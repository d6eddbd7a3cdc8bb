"""Experiment: measured time of every distinct vag_gemm launch of a cfg2 training step (shapes taken from
tools/list_gemms.py), cost-model choice vs the best (tile, split-K) found by exhaustive search."""
import os, sys, re, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "list_gemms.py")], capture_output=True, text=True).stdout
shapes = []
for l in out.splitlines():
    m = re.match(r"\s*(\d+)x M=(\d+) N=(\d+) K=(\d+) akc=(\d) bkc=(\d) beta=(\d)", l)
    if m:
        shapes.append(tuple(int(x) for x in m.groups()))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
def timed(run, reps=10):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
full = "--sweep" in sys.argv
tot_d = tot_b = 0.0
for cnt, M, N, K, akc, bkc, beta in shapes:
    if akc:
        A = torch.randn(M, K, device=dev); sa = (K, 1)
    else:
        A = torch.randn(K, M, device=dev); sa = (1, M)
    if bkc:
        B = torch.randn(N, K, device=dev); sb = (1, K)
    else:
        B = torch.randn(K, N, device=dev); sb = (N, 1)
    ldc = (N + 3) // 4 * 4
    C = torch.zeros(M, ldc, device=dev)
    def run():
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), sa[0], sa[1], L.ptr(B), sb[0], sb[1], float(beta), L.ptr(C), ldc, None, 0, L.stream())
    os.environ.pop("VAG_GEMM_FORCE", None)
    run(); torch.cuda.synchronize()
    base = timed(run)
    best = (base, "model")
    if full:
        for T in (64, 128):
            if T == 128 and (M <= 64 or N <= 64):
                continue
            for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
                if sp > 1 and K // sp < 128:
                    continue
                os.environ["VAG_GEMM_FORCE"] = "%d,%d" % (T, sp)
                t = timed(run)
                if t < best[0]:
                    best = (t, "%d/%d" % (T, sp))
        os.environ.pop("VAG_GEMM_FORCE", None)
    tot_d += cnt * base; tot_b += cnt * best[0]
    gf = 2.0 * M * N * K
    print("%dx %5dx%5dx%5d akc=%d bkc=%d b%d  %6.1f us %6.1f TF/s | best %6.1f (%s)" % (cnt, M, N, K, akc, bkc, beta, base, gf / base / 1e6, best[0], best[1]), flush=True)
print("sum over the step: model choice %.1f us, best %.1f us" % (tot_d, tot_b))
