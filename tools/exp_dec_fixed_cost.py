"""Fixed cost of a persistent decoder forward launch: its duration (HIP events around the launch) against the number of time steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import numpy as np
import torch, bench
dev = torch.device("cuda:0")
xs, ys, es = [], [], []
for Tt in (2, 4, 8, 16, 24, 32, 40):
    c = dict(bench.CFG2, Tt=Tt)
    fam = bench.measure_operators(c, dev)
    xs.append(Tt); ys.append(fam["decoder_recurrence"] * 1e6); es.append(fam["encoder_fwd"] * 1e6)
    print("Tt = %2d: decoder recurrence launch %.1f us" % (Tt, ys[-1]), flush=True)
a, b = np.polyfit(xs, ys, 1)
print("fit: %.2f us per step + %.1f us per launch" % (a, b))
