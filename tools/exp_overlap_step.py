"""Diagnostics for the overlapped step: host enqueue time of one vag_train_step call and GPU time per step, for a few
settings of the schedule (environment read when the library first builds its streams, so one setting per process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip.trainer import TrainStep

overlap = int(os.environ.get("OVERLAP", "1"))
dev = torch.device("cuda:0")
c = bench.CFG2
from machine_translation_vision.losses import PairwiseRankingLoss
model = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(model, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), teacher_force_ratio=1.0,
               use_graph=not overlap and not os.environ.get('EAGER'), overlap=bool(overlap))
src, lens, tgt, im = bench.make_batch(c, 0, dev)
batch = (src, torch.tensor(lens, dtype=torch.int32, device=dev), tgt, im)
model.train()
if os.environ.get("SIDE", "1") == "1":
    # the library's CU-masked streams are "blocking" streams (hipExtStreamCreateWithCUMask has no flags): they synchronise
    # implicitly with the NULL stream, so the step must not be driven from it
    torch.cuda.set_stream(torch.cuda.Stream())
for _ in range(5):
    ts.step(*batch, teacher=True)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(50):
    h0 = time.perf_counter()
    ts.step(*batch, teacher=True)
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("overlap=%d %s: %.3f ms/step GPU-inclusive, host enqueue median %.3f ms" %
      (overlap, {k: v for k, v in os.environ.items() if k.startswith("VAG_")}, (t1 - t0) / 50 * 1e3,
       sorted(host)[25] * 1e3), flush=True)
