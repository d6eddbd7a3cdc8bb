"""can the MFMA-bound products run BESIDE the latency-bound persistent recurrences?

Footprints first (profiles/r04_footprints.txt): dec_fwd / dec_bwd_persistent_kernel take 256 VGPRs at two waves per SIMD -- the whole
register file of every CU -- so nothing can share a CU with the decoder recurrences (1.5 of the step's 3.3 ms).  The encoder
kernels (192 / 181 VGPRs, 84 KB of LDS) leave 128-144 registers per SIMD lane and 74 KB: room for ONE more wave per SIMD, which the
8-wave bf16x6 blocks (2 waves per SIMD at 100-128 VGPRs, 60 KB) do not fit into.  What CAN be tried without a new kernel is the fork
itself: the decoder's weight-gradient products (a leaf of the backward pass, ~190 us) on a side stream beside [encoder backward
recurrence + its weight gradients] (phase 4 of vag_train_step, ~400 us), launched before or after it.

Usage (GPU box): python tools/exp_overlap_persist.py > profiles/r04_exp_overlap_persist.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from vagnmt_hip import _lib as L  # noqa: E402
from machine_translation_vision.losses import PairwiseRankingLoss  # noqa: E402
from vagnmt_hip.trainer import TrainStep  # noqa: E402

dev = torch.device("cuda:0")
c = bench.CFG2
m = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev)
vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), teacher_force_ratio=1.0, use_graph=False)
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
m.train()
side = torch.cuda.Stream()
L.set_option("persist_timing", 1)
R = c["B"] * c["Tt"]
H = c["H"]
# the decoder's weight-gradient products g_W += dY^T X (M, N, K = rows): W_hh1, [attn_h; W_hh2], W_ih2 W_c2h (folded), attn_e, W_ih1
SHAPES = [(3 * H, H, R), (2 * H + 3 * H, H, R), (3 * H, 2 * H, R), (2 * H, 2 * H, R), (3 * H, c["E"], R)]
bufs = []
for (M, N, K) in SHAPES:
    bufs.append((torch.randn(K, M, device=dev), torch.randn(K, N, device=dev), torch.zeros(M, N, device=dev)))


def leaf(stream):
    for (M, N, K), (A, B, Cm) in zip(SHAPES, bufs):
        L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(A), 1, M, L.ptr(B), N, 1, 1.0, L.ptr(Cm), N, None, 0, C.c_void_p(stream.cuda_stream))


def read_rec(kind):
    ms, n = C.c_double(0), C.c_int(0)
    L.lib().vag_recurrence_time(kind, C.byref(ms), C.byref(n))
    return (ms.value / max(n.value, 1)) * 1e3


def run(label, mode, reps=7):
    ts_, ks_ = [], []
    for _ in range(reps):
        ts.fp.grad.zero_()
        ts.backend.run(src, lt, tgt, im, True, 3)
        torch.cuda.synchronize()
        read_rec(2)
        main = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if mode == "leaf_only":
            leaf(main)
        elif mode == "phase4_only":
            ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
        elif mode == "serial":
            leaf(main)
            ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
        elif mode == "fork_leaf_first":
            side.wait_stream(main)
            leaf(side)
            ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
            main.wait_stream(side)
        elif mode == "fork_phase4_first":
            side.wait_stream(main)
            ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
            leaf(side)
            main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize()
        ts_.append(e0.elapsed_time(e1) * 1e3)
        ks_.append(read_rec(2))
    to = L.lib().vag_persistent_timeouts()
    ts_.sort(), ks_.sort()
    print("%-58s %8.1f us (min %7.1f)   enc_bwd_persistent_kernel %7.1f us   give-ups %d" % (label, ts_[len(ts_) // 2], ts_[0], ks_[len(ks_) // 2], to),
          flush=True)


print("# decoder weight-gradient products (leaf) beside phase 4 = [enc_bwd_persistent_kernel + encoder weight gradients]; configs[1], eager, median of 7")
run("leaf alone (5 products, one stream)", "leaf_only")
run("phase 4 alone", "phase4_only")
run("serial: leaf, then phase 4 (one stream)", "serial")
run("fork: leaf on a side stream, enqueued FIRST", "fork_leaf_first")
run("fork: phase 4 enqueued first, leaf on a side stream", "fork_phase4_first")
run("serial (again)", "serial")

# second question: the same leaf beside the chain of ~28 small dependent launches of the VSE / initial-state backward (B = 64 rows each,
# 8-128 workgroups of 256-512 threads; stand-ins here: skinny products 64 x 512 x 1024 and 64 x 1024 x 512 feeding each other)
xa = torch.randn(64, 1024, device=dev)
xb = torch.empty(64, 512, device=dev)
Wa = torch.randn(512, 1024, device=dev) / 32
Wb = torch.randn(1024, 512, device=dev) / 23


def chain(stream, n=14):
    s_ = C.c_void_p(stream.cuda_stream)
    for _ in range(n):
        L.call("vag_linear_fwd", 64, 512, 1024, L.ptr(xa), L.ptr(Wa), None, 1, L.ptr(xb), s_)
        L.call("vag_linear_fwd", 64, 1024, 512, L.ptr(xb), L.ptr(Wb), None, 1, L.ptr(xa), s_)


def run2(label, mode, reps=7):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        if mode == "chain":
            chain(main)
        elif mode == "serial":
            leaf(main)
            chain(main)
        elif mode == "fork_leaf_first":
            side.wait_stream(main)
            leaf(side)
            chain(main)
            main.wait_stream(side)
        elif mode == "fork_chain_first":
            side.wait_stream(main)
            chain(main)
            leaf(side)
            main.wait_stream(side)
        e1.record(main)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    out.sort()
    print("%-58s %8.1f us (min %7.1f)" % (label, out[len(out) // 2], out[0]), flush=True)


print("# the leaf beside a chain of 28 small dependent launches (eager)")
run2("chain alone (28 launches)", "chain")
run2("serial: leaf, then chain", "serial")
run2("fork: leaf on a side stream, enqueued first", "fork_leaf_first")
run2("fork: chain enqueued first, leaf on a side stream", "fork_chain_first")
