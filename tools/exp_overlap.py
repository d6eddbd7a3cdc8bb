"""Experiment: does a chain of small dependent kernels overlap with large GEMMs on a parallel graph branch?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
M, N, K = 64, 512, 1536
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 30; y = torch.empty(M, N, device=dev)
A = torch.randn(1536, 2560, device=dev); Bm = torch.randn(2560, 1024, device=dev); C = torch.zeros(1536, 1024, device=dev)
def chain(s, n=100):
    for _ in range(n):
        L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), None, 0, L.ptr(y), s)
def gemms(s, n=10):
    for _ in range(n):
        L.call("vag_gemm_f32", 1536, 1024, 2560, 1.0, L.ptr(A), 2560, 1, L.ptr(Bm), 1024, 1, 0.0, L.ptr(C), 1024, None, 0, s)
def timeit(g, reps=5):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
cur = torch.cuda.current_stream().cuda_stream
chain(cur, 2); gemms(cur, 1); torch.cuda.synchronize()
def cap(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g
gA = cap(lambda: chain(torch.cuda.current_stream().cuda_stream))
gB = cap(lambda: gemms(torch.cuda.current_stream().cuda_stream))
side = torch.cuda.Stream()
def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemms(side.cuda_stream)
    chain(main.cuda_stream)
    main.wait_stream(side)
gAB = cap(both)
def serial():
    s = torch.cuda.current_stream().cuda_stream
    gemms(s); chain(s)
gS = cap(serial)
print("chain alone %.3f ms | gemms alone %.3f ms | serial %.3f ms | two branches %.3f ms" % (timeit(gA), timeit(gB), timeit(gS), timeit(gAB)))
