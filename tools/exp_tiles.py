import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
N_CHAIN = 300
def bench(fn):
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
    return (time.perf_counter() - t0) / 5 / N_CHAIN * 1e6
M = 64
for (N, K) in [(512, 2560), (512, 1536), (1024, 1536), (2560, 512), (512, 512)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 30; y = torch.empty(M, N, device=dev)
    out = []
    for mt in (1, 2, 4):
        for cols in (16, 8):
            os.environ["VAG_SKINNY_TILE"] = "%d,%d" % (mt, cols)
            us = bench(lambda s: L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), None, 0, L.ptr(y), s))
            out.append("mt%d/c%-2d %5.2f" % (mt, cols, us))
    print("N=%4d K=%4d: " % (N, K) + " | ".join(out), flush=True)
