"""Experiment: per-kernel cost of dependent small kernels inside a captured graph (chain of 400)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
N_CHAIN = 400

def bench(name, fn):
    fn(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(N_CHAIN):
            fn(s)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 5 / N_CHAIN * 1e6
    print("%-44s %6.2f us/kernel" % (name, us), flush=True)

x1 = torch.zeros(64, device=dev)
rng = torch.zeros(2, dtype=torch.int64, device=dev)
bench("rng_advance (1 thread of work)", lambda s: L.call("vag_rng_advance", L.ptr(rng, torch.int64), s))
for (M, N, K) in [(64, 512, 512), (64, 1536, 512), (64, 2560, 512), (64, 512, 1024), (64, 512, 1536), (64, 512, 2560),
                  (16, 512, 512), (64, 64, 512), (64, 512, 64)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 30; y = torch.empty(M, N, device=dev)
    # ping-pong so each launch depends on the previous one's output when shapes allow
    bench("linear M=%d N=%d K=%d (W %.1f MB)" % (M, N, K, N * K * 4 / 1e6),
          lambda s, x=x, W=W, y=y, M=M, N=N, K=K: L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), None, 0, L.ptr(y), s))
