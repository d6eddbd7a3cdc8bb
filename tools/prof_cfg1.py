"""Workload for a kernel trace of BASELINE configs[0] (text-only, H = 256, B = 16): graph-replayed optimiser steps.
rocprofv3 --kernel-trace -d DIR -- python3 tools/prof_cfg1.py ; python tools/step_timeline.py DIR --full"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip.trainer import TrainStep
c = bench.CFG1
dev = torch.device("cuda:0")
m = bench.build_text_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), None, lr=4e-4, weight_decay=1e-5, clip=1.0, teacher_force_ratio=1.0)
src, lens, tgt = bench.make_text_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
for _ in range(30):
    ts.step(src, lt, tgt, None, teacher=True)
torch.cuda.synchronize()
ts.check()
