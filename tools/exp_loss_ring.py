"""What the per-step copy of the loss words cost: TrainStep.step with the result ring (shipped) against the same driver whose
outputs() clones the static words as rounds 1-3 did (one eager copy launch between two graph replays).
Usage (GPU box): python tools/exp_loss_ring.py >> profiles/r04_exp_small_launches.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from machine_translation_vision.losses import PairwiseRankingLoss
from vagnmt_hip.trainer import TrainStep
c = bench.CFG2
dev = torch.device("cuda:0")
model = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(model, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), lr=4e-4, weight_decay=1e-5,
               clip=1.0, teacher_force_ratio=1.0)
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
be = ts.backend
ring = be.outputs


def clone_outputs():
    out = be.f.losses.clone()
    return out[0], out[1], out[2]


def run(n):
    for _ in range(10):
        ts.step(src, lt, tgt, im)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step(src, lt, tgt, im)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("configs[1] step, results handed out as views of the device-side ring vs a clone per step (ms per step, alternating):")
for rep in range(3):
    be.outputs = ring
    a = run(60)
    be.outputs = clone_outputs
    b = run(60)
    print("  ring %.4f   clone %.4f   (%+.1f us)" % (a, b, (b - a) * 1e3))
