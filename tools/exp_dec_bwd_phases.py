"""Phase timestamps of the persistent decoder BACKWARD kernel (workgroup 0): one fused training step at configs[1], eager.
Columns: A (d alpha shares + drain + arrive) | wait A | B (softmax bwd, dq, publish) | hidden side (dgh2 W_hh2) | wait B |
C (dq attn_h, gru_1 backward, publish) | wait C | D (dgh1 W_hh1, gru_2 backward of t-1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import numpy as np
import torch, bench
from vagnmt_hip import _lib as L
L.use_lab_build()          # the product library carries no stamp / debug hooks (csrc/Makefile: LAB=1)
from vagnmt_hip.trainer import TrainStep
from machine_translation_vision.losses import PairwiseRankingLoss
c = bench.CFG2
dev = torch.device("cuda:0")
Tt = c["Tt"]
m = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), use_graph=False)
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
for _ in range(3):
    ts.step(src, lt, tgt, im, teacher=True)
st = torch.zeros((Tt + 1) * 8, dtype=torch.int64, device=dev)
L.set_option("dec_bwd_stamps", st.data_ptr())
ts.step(src, lt, tgt, im, teacher=True)
torch.cuda.synchronize()
L.set_option("dec_bwd_stamps", 0)
raw = st.cpu().numpy().reshape(Tt + 1, 8).astype(np.float64) * 0.01
s = raw[:Tt]
pro = raw[Tt]
print("workgroup 0 (us): entry -> images in LDS %.2f, entry -> first stamp of step Tt-2 %.2f, entry -> exit %.2f" % (pro[1] - pro[0], s[Tt - 2][0] - pro[0], pro[2] - pro[0]))
rows = []
for t in range(Tt - 2, 0, -1):          # steps run Tt-1 .. 0; the next step after t is t-1
    a = s[t]
    nxt = s[t - 1][0]
    rows.append([a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5], a[7] - a[6], nxt - a[7]])
r = np.array(rows)
names = ["A shares", "wait A", "B ds+dq", "hidden", "wait B", "C gru_1", "wait C", "D gru_2"]
for n, mm, md in zip(names, r.mean(0), np.median(r, 0)):
    print("  %-9s mean %6.2f  median %6.2f" % (n, mm, md))
print("  total     %6.2f us per step" % r.sum(1).mean())
