"""Beam search on raw logits + log-sum-exp pieces (no normalising pass) against the log-softmax path: same hypotheses and scores,
time per step.  Usage (GPU box): python tools/exp_beam_raw_logits.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import numpy as np
import torch, bench
c = dict(bench.CFG2); c["B"] = 16
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=True)
res = {}
for raw in (True, False, True, False):
    m.decode_raw_logits = raw
    for _ in range(2):
        hyp = m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n):
        hyp = m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    steps = m.last_decode_steps
    res[raw] = ([[int(t) for t in h] for h in hyp], m.last_beam_scores.cpu().numpy().copy())
    print("raw logits %-5s: %.2f ms per batch, %d steps, %.1f us per step" % (raw, dt * 1e3, steps, dt / steps * 1e6), flush=True)
same = sum(a == b for a, b in zip(res[True][0], res[False][0]))
print("identical hypotheses: %d of 16; max score difference %.2e" % (same, float(np.abs(res[True][1] - res[False][1]).max())))
