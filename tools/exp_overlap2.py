"""Experiment (round 2): can weight-gradient products hide under a recurrent chain when their blocks are SMALL-footprint
(64x64 f32-MFMA tiles: 256 threads, ~48 VGPRs, 35 KB LDS) so that they co-reside with the chain's 1024-thread workgroups?
chain = N dependent gru_bwd_step launches (the dominant chain kernel, M=64, H=512, K=3H) or the forward cell;
side  = weight-gradient shaped products (1536x512x2560, both operands outer-contiguous, beta=1).
Prints: chain alone, side alone, serial, two graph branches -- for the side products on 128x128 bf16x6 blocks and on
64x64 blocks with several split-K factors."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
from vagnmt_hip._lib import call, ptr
dev = torch.device("cuda:0")
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


def chain_bwd(s):
    for _ in range(NCH):
        call("vag_gru_cell_bwd", ptr(dgh), ptr(wt), ptr(carry), ptr(d_out), ptr(sv), ptr(hp), B, H, ptr(dgi), ptr(dgh_o),
             ptr(cout), s)


def chain_fwd(s):
    for _ in range(NCH):
        call("vag_gru_cell_fwd", ptr(gi), ptr(hp), ptr(whh), ptr(bhh), B, H, ptr(ho), ptr(sv2), s)


def side(s):
    for g in gW:      # g_W[m,n] += sum_r dY[r,m] X[r,n]
        call("vag_gemm_f32", 3 * H, H, R, 1.0, ptr(dY), 1, 3 * H, ptr(X), H, 1, 1.0, ptr(g), H, None, 0, s)


def timeit(g, reps=5):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cap(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


cur = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
sidestream = torch.cuda.Stream()
for name, chain in (("gru_bwd_step K=1536 (1024 thr)", chain_bwd), ("gru cell fwd (512 thr)", chain_fwd)):
    chain(cur()); torch.cuda.synchronize()
    gA = cap(lambda: chain(cur()))
    tA = timeit(gA)
    print("== chain: %s x %d: %.3f ms (%.2f us/launch)" % (name, NCH, tA, tA * 1e3 / NCH), flush=True)
    for force in (None, "64,1", "64,2", "64,4", "64,8", "128,1", "128,2"):
        if force:
            os.environ["VAG_GEMM_FORCE"] = force
        else:
            os.environ.pop("VAG_GEMM_FORCE", None)
        side(cur()); torch.cuda.synchronize()
        gB = cap(lambda: side(cur()))

        def both():
            main = torch.cuda.current_stream()
            sidestream.wait_stream(main)
            with torch.cuda.stream(sidestream):
                side(sidestream.cuda_stream)
            chain(main.cuda_stream)
            main.wait_stream(sidestream)
        gAB = cap(both)
        tB, tAB = timeit(gB), timeit(gAB)
        print("   side %-8s: alone %.3f ms | serial %.3f | two branches %.3f ms  (hidden %.0f%% of the side work)"
              % (force or "model", tB, tA + tB, tAB, 100.0 * (tA + tB - tAB) / max(tB, 1e-9)), flush=True)
os.environ.pop("VAG_GEMM_FORCE", None)
