"""Workload for PMC passes over WHOLE optimiser steps: three eager steps (vag_train_step + clip/Adam + derived weights) of
configs[1] or configs[4], bracketed by marker launches (rng_advance_kernel), so that tools/pmc_step_summary.py can report HBM
traffic per step and per kernel.  Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/prof_step.py [cfg2|cfg5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip._lib import call, ptr, stream
from vagnmt_hip.trainer import TrainStep
from machine_translation_vision.losses import PairwiseRankingLoss
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
c = bench.CFG5 if cfg == "cfg5" else bench.CFG2
dev = torch.device("cuda:0")
m = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), lr=4e-4, weight_decay=1e-5, clip=1.0,
               teacher_force_ratio=1.0, use_graph=False, storage="f16" if cfg == "cfg5" else "f32")
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
marker = torch.zeros(2, dtype=torch.int64, device=dev)
for _ in range(2):
    ts.step(src, lt, tgt, im, teacher=True)
torch.cuda.synchronize()
call("vag_rng_advance", ptr(marker, torch.int64), stream())
for _ in range(3):
    ts.step(src, lt, tgt, im, teacher=True)
call("vag_rng_advance", ptr(marker, torch.int64), stream())
torch.cuda.synchronize()
ts.check()
