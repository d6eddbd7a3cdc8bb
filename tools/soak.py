"""Soak run of the configs[1] training loop as the reference drives it: N optimiser steps at teacher forcing ratio 0.8 (a python
coin per batch), ragged and full-length batches alternating, a beam-12 and a greedy decode of an eval batch every 200 steps,
check() (device-side skip counter, persistent-kernel give-ups, result-ring consistency) at the end.
Usage (GPU box): python tools/soak.py [steps] [H=256] [B=128] >> profiles/rNN_soak.txt"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from machine_translation_vision.losses import PairwiseRankingLoss
from vagnmt_hip.trainer import TrainStep
from vagnmt_hip import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
c = dict(bench.CFG2)
for kv in sys.argv[2:]:           # e.g. H=256 B=128: the round-6 shapes of the one-launch recurrences (H = 256; two passes of row tiles)
    k_, v_ = kv.split("=")
    c[k_] = int(v_)
print("# soak: %s" % {k_: c[k_] for k_ in ("B", "Ts", "Tt", "H")}, flush=True)
dev = torch.device("cuda:0")
model = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(model, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), lr=4e-4, weight_decay=1e-5,
               clip=1.0, teacher_force_ratio=0.8)
full = bench.make_batch(c, 0, dev)
rag = bench.make_batch(c, 1, dev, ragged=True)
lt = [torch.tensor(b[1], dtype=torch.int32, device=dev) for b in (full, rag)]
c4 = dict(c); c4["B"] = 16
ev = bench.make_batch(c4, 2, dev, ragged=True)
random.seed(7)
losses = []
t0 = time.perf_counter()
for i in range(N):
    b, l = (full, lt[0]) if i % 2 == 0 else (rag, lt[1])
    out = ts.step(b[0], l, b[2], b[3])
    if i % 100 == 0:
        losses.append(float(out[0]))
    if i % 200 == 199:
        model.eval()
        hb = model.beamsearch_decode(ev[0], ev[1], ev[3], 12, 80)
        hg = model.beamsearch_decode(ev[0], ev[1], ev[3], 1, 80)
        model.train()
        print("step %5d  loss %.4f  beam-12 mean length %.1f  greedy mean length %.1f" %
              (i + 1, losses[-1], sum(map(len, hb)) / 16.0, sum(map(len, hg)) / 16.0), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ts.check()
print("%d steps + %d decode pairs in %.1f s; loss %.3f -> %.3f; skipped steps %d, persistent give-ups %d, check() passed" %
      (N, N // 200, dt, losses[0], losses[-1], ts.skipped_steps(), L.lib().vag_persistent_timeouts()))
