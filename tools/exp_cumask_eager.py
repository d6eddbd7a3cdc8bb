"""Eager (not graph-replayed) chains of dependent small launches on a CU-masked stream vs an ordinary stream."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip._lib import call, ptr
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
def masked(lo, hi):
    m = (C.c_uint32 * 8)()
    for i in range(lo, hi): m[i // 32] |= 1 << (i % 32)
    s = C.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m) == 0
    return torch.cuda.ExternalStream(s.value, device=dev)
B, H, N = 64, 512, 200
gi = torch.randn(B, 3 * H, device=dev); hp = torch.randn(B, H, device=dev); ho = torch.empty(B, H, device=dev)
whh = torch.randn(3 * H, H, device=dev) / 30; bhh = torch.zeros(3 * H, device=dev); sv = torch.empty(4, B, H, device=dev)
def chain(s):
    for _ in range(N):
        call("vag_gru_cell_fwd", ptr(gi), ptr(hp), ptr(whh), ptr(bhh), B, H, ptr(ho), ptr(sv), s.cuda_stream)
for name, s in (("plain", torch.cuda.Stream()), ("masked 0-256", masked(0, 256)), ("masked 0-160", masked(0, 160))):
    chain(s); torch.cuda.synchronize()
    t0 = time.perf_counter(); chain(s); th = time.perf_counter() - t0; torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        chain(s)
    with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize(); tg = time.perf_counter() - t0
    print("%-14s eager: host %.2f us/launch, total %.2f us/launch | graph %.2f us/launch" % (name, th / N * 1e6, t1 / N * 1e6, tg / N * 1e6), flush=True)
