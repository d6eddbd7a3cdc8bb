"""Experiment (round 6): optimiser-step time with the one-launch recurrence kernels (passes of row tiles for wide batches) against the
launch chains, at H = 512 / 256 and B = 16 .. 256.  python tools/exp_dec_passes.py > gpurun_out/exp_dec_passes.txt"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import _lib as L
from vagnmt_hip.trainer import TrainStep
from machine_translation_vision.losses import PairwiseRankingLoss

dev = torch.device("cuda:0")


def run(H, B, persistent, n=30):
    c = dict(bench.CFG2, H=H, B=B)
    L.set_option("persistent", persistent)
    m = bench.build_model(c, dev)
    vw = torch.ones(c["V"], device=dev); vw[0] = 0
    ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), teacher_force_ratio=1.0)
    src, lens, tgt, im = bench.make_batch(c, 0, dev)
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    for _ in range(5):
        ts.step(src, lt, tgt, im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(src, lt, tgt, im)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    ts.check()
    L.set_option("persistent", 1)
    return dt


for H in (512, 256):
    for B in (16, 64, 96, 128, 192, 256) if H == 512 else (16, 64, 128, 256, 384, 512):
        a, b = run(H, B, 1), run(H, B, 0)
        print("H=%d B=%3d Ts=Tt=40: persistent %.3f ms, chains %.3f ms  (x%.2f)  supported=%d" %
              (H, B, a, b, b / a, L.lib().vag_recurrence_supported(1, B, 40, 40, H)), flush=True)
