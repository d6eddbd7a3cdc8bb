"""Experiment: do parallel branches of a captured HIP graph run concurrently on this ROCm?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L

dev = torch.device("cuda:0")
M, N, K = 16, 1536, 512
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = 200
xs = [torch.randn(M, K, device=dev) for _ in range(G)]
ys = [torch.empty(M, N, device=dev) for _ in range(G)]
zs = [torch.empty(M, K, device=dev) for _ in range(G)]
W = torch.randn(N, K, device=dev) / 30
W2 = torch.randn(K, N, device=dev) / 30


def chain(g, stream):
    s = stream.cuda_stream
    for _ in range(STEPS):
        L.call("vag_linear_fwd", M, N, K, L.ptr(xs[g]), L.ptr(W), None, 0, L.ptr(ys[g]), s)
        L.call("vag_linear_fwd", M, K, N, L.ptr(ys[g]), L.ptr(W2), None, 1, L.ptr(zs[g]), s)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cur = torch.cuda.current_stream()
chain(0, cur); torch.cuda.synchronize()
# serial graph: G chains back to back on one stream
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    for g in range(G):
        chain(g, torch.cuda.current_stream())
# parallel graph: G chains on G streams
streams = [torch.cuda.Stream() for _ in range(G)]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    for g in range(G):
        streams[g].wait_stream(main)
        with torch.cuda.stream(streams[g]):
            chain(g, streams[g])
    for g in range(G):
        main.wait_stream(streams[g])
t1 = timeit(g1.replay)
t2 = timeit(g2.replay)
print("G=%d chains x %d kernels: serial graph %.3f ms (%.2f us/kernel), branched graph %.3f ms -> speedup %.2fx"
      % (G, 2 * STEPS, t1, t1 * 1e3 / (G * 2 * STEPS), t2, t1 / t2))
# eager multi-stream for reference
def eager_par():
    main = torch.cuda.current_stream()
    for g in range(G):
        streams[g].wait_stream(main)
        chain(g, streams[g])
    for g in range(G):
        main.wait_stream(streams[g])
print("eager multi-stream: %.3f ms" % timeit(eager_par, 3))
