"""Experiment (round 2): CU-masked streams (hipExtStreamCreateWithCUMask) for running the dense side work BESIDE a recurrent
chain instead of after it.
  A. does a mask restrict a kernel (a 4096^3 product on 256 / 192 / 128 / 64 CUs), eagerly and from a captured graph
     replayed on the masked stream?
  B. chain of dependent backward-step launches on `--chain-cus` CUs and weight-gradient products on the remaining CUs:
     chain alone (all CUs / masked), side alone (all CUs / masked), serial, concurrent.
Masks: bit i of the mask = "CU i" in the driver's numbering (the driver spreads consecutive bits over the XCDs)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
from vagnmt_hip._lib import call, ptr

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(lo, hi, total=256):
    words = (total + 31) // 32
    m = (C.c_uint32 * words)()
    for i in range(lo, hi):
        m[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def timeit(fn, stream, reps=5):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(reps):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cap(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


cur = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
# ---------------- A ----------------
n = 4096
a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev); c = torch.empty(n, n, device=dev)


def big():
    call("vag_gemm_f32", n, n, n, 1.0, ptr(a), n, 1, ptr(b), 1, n, 0.0, ptr(c), n, None, 0, cur())


plain = torch.cuda.Stream()
print("A. 4096^3 product: unmasked stream %.3f ms" % timeit(big, plain), flush=True)
for ncu in (() if os.environ.get("SKIP_A") else (256, 192, 128, 64, 32)):
    s = masked_stream(0, ncu)
    te = timeit(big, s)
    g = cap(big, s)
    tg = timeit(g.replay, s)
    gp = cap(big, plain)
    tgp = timeit(gp.replay, s)       # captured on a plain stream, replayed on the masked one
    print("   mask %3d CUs: eager %.3f ms | graph captured+replayed on it %.3f ms | graph captured elsewhere, replayed on it "
          "%.3f ms" % (ncu, te, tg, tgp), flush=True)

# ---------------- B ----------------
B, H = 64, 512
NCH = int(os.environ.get("NCH", "100"))
NG = int(os.environ.get("NG", "6"))
dgh = torch.randn(B, 3 * H, device=dev); wt = torch.randn(H, 3 * H, device=dev) / 30
carry = torch.randn(B, H, device=dev); d_out = torch.randn(B, H, device=dev)
sv = torch.rand(4, B, H, device=dev) * 0.8 + 0.1; hp = torch.randn(B, H, device=dev)
dgi = torch.empty(B, 3 * H, device=dev); dgh_o = torch.empty(B, 3 * H, device=dev); cout = torch.empty(B, H, device=dev)
gi = torch.randn(B, 3 * H, device=dev); ho = torch.empty(B, H, device=dev); whh = torch.randn(3 * H, H, device=dev) / 30
bhh = torch.zeros(3 * H, device=dev); sv2 = torch.empty(4, B, H, device=dev)
R = 2560
dY = torch.randn(R, 3 * H, device=dev); X = torch.randn(R, H, device=dev)
gW = [torch.zeros(3 * H, H, device=dev) for _ in range(NG)]


def chain_bwd():
    s = cur()
    for _ in range(NCH):
        call("vag_gru_cell_bwd", ptr(dgh), ptr(wt), ptr(carry), ptr(d_out), ptr(sv), ptr(hp), B, H, ptr(dgi), ptr(dgh_o),
             ptr(cout), s)


def chain_fwd():
    s = cur()
    for _ in range(NCH):
        call("vag_gru_cell_fwd", ptr(gi), ptr(hp), ptr(whh), ptr(bhh), B, H, ptr(ho), ptr(sv2), s)


def side():
    s = cur()
    for g in gW:
        call("vag_gemm_f32", 3 * H, H, R, 1.0, ptr(dY), 1, 3 * H, ptr(X), H, 1, 1.0, ptr(g), H, None, 0, s)


for name, chain in (("gru_bwd_step (128 WGs x 1024 thr)", chain_bwd), ("gru cell fwd (256 WGs x 512 thr)", chain_fwd)):
    gA0 = cap(chain, plain)
    gB0 = cap(side, plain)
    tA0, tB0 = timeit(gA0.replay, plain), timeit(gB0.replay, plain)
    print("B. chain %s x %d: %.3f ms on all CUs; side (%d weight-gradient products) %.3f ms on all CUs; serial %.3f"
          % (name, NCH, tA0, NG, tB0, tA0 + tB0), flush=True)
    for ncu in (224, 192, 160, 128):
        sA, sB = masked_stream(0, ncu), masked_stream(ncu, 256)
        gA, gB = cap(chain, sA), cap(side, sB)
        tA, tB = timeit(gA.replay, sA), timeit(gB.replay, sB)

        def both():
            with torch.cuda.stream(sA):
                gA.replay()
            with torch.cuda.stream(sB):
                gB.replay()
        both(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            both()
        torch.cuda.synchronize()
        tAB = (time.perf_counter() - t0) / 5 * 1e3
        print("   chain on %3d CUs %.3f ms | side on %3d CUs %.3f ms | concurrent %.3f ms (serial on all CUs %.3f)"
              % (ncu, tA, 256 - ncu, tB, tAB, tA0 + tB0), flush=True)

        def both2():          # chain on an UNMASKED stream, only the side work confined
            with torch.cuda.stream(plain):
                gA0.replay()
            with torch.cuda.stream(sB):
                gB.replay()
        both2(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            both2()
        torch.cuda.synchronize()
        print("      chain unmasked beside side on %3d CUs: concurrent %.3f ms" % (256 - ncu, (time.perf_counter() - t0) / 5 * 1e3),
              flush=True)
