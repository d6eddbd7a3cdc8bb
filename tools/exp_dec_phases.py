"""Where a step of the persistent decoder forward kernel spends its time: 100 MHz timestamps taken by workgroup 0 at its phase
boundaries (vag_set_option("dec_stamps", device address)).  Columns per step: wait for h2 | phase 1 work | wait h1 | phase 2 with the
score shares | drain of the atomics | arrive | wait scores | phase 4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import numpy as np
import torch, bench
from vagnmt_hip import _lib as L
L.use_lab_build()          # the product library carries no stamp / debug hooks (csrc/Makefile: LAB=1)
c = bench.CFG2
dev = torch.device("cuda:0")
Tt = c["Tt"]
st = torch.zeros((Tt + 4) * 8, dtype=torch.int64, device=dev)
L.set_option("dec_stamps", st.data_ptr())
fam = bench.measure_operators(c, dev)
torch.cuda.synchronize()
L.set_option("dec_stamps", 0)
raw = st.cpu().numpy().reshape(Tt + 4, 8).astype(np.float64) * 0.01        # us
s = raw[:Tt]
pro = raw[Tt]
print("prologue of workgroup 0 (us): weights -> registers %.2f, keys -> LDS %.2f, entry -> first step %.2f; last step's end - entry %.2f"
      % (pro[1] - pro[0], pro[2] - pro[1], s[0][0] - pro[0], s[Tt - 1][7] - pro[0]))
nt = (c["B"] + 15) // 16
t0 = raw[Tt + 1][:nt].min()
print("row tiles (us after the first entry): entry of workgroup 0 %s, entry of workgroup 63 %s, exit of workgroup 0 %s"
      % (np.round(raw[Tt + 1][:nt] - t0, 2).tolist(), np.round(raw[Tt + 3][:nt] - t0, 2).tolist(), np.round(raw[Tt + 2][:nt] - t0, 2).tolist()))
rows = []
for t in range(1, Tt - 1):
    a = s[t]
    nxt = s[t + 1][0]
    rows.append([a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5], a[7] - a[6], nxt - a[7]])
r = np.array(rows)
names = ["wait h2", "phase1", "wait h1", "phase2+scores", "atomic drain", "(arrive)", "wait sc", "phase4"]
print("per step (us), mean over steps 1..%d of workgroup 0:" % (Tt - 2))
for n, m, md in zip(names, r.mean(0), np.median(r, 0)):
    print("  %-8s mean %6.2f  median %6.2f" % (n, m, md))
print("  total    %6.2f" % r.sum(1).mean())
print("operator timings:", {k: round(v * 1e6, 1) for k, v in fam.items()})
