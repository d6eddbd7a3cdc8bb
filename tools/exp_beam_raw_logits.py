"""Beam search (configs[3]): expansion on raw logits + log-sum-exp pieces against the log-softmax path, and the hoisted decoding
step (keys projected once per call, four launches, no context) against the five-launch step: same hypotheses and scores, time
per step.  Usage (GPU box): python tools/exp_beam_raw_logits.py >> profiles/r04_exp_beam.txt"""
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
for raw, hoist in ((True, True), (False, False), (True, False), (False, True), (True, True), (False, False)):
    m.decode_raw_logits, m.decode_hoisted = raw, hoist
    for _ in range(2):
        hyp = m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n):
        hyp = m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    steps = m.last_decode_steps
    res[(raw, hoist)] = ([[int(t) for t in h] for h in hyp], m.last_beam_scores.cpu().numpy().copy())
    print("raw logits %-5s hoisted step %-5s: %.2f ms per batch, %d steps, %.1f us per step" % (raw, hoist, dt * 1e3, steps, dt / steps * 1e6),
          flush=True)
ref = res[(False, False)]
for kk, v in res.items():
    same = sum(a == b for a, b in zip(v[0], ref[0]))
    print("%s vs (False, False): identical hypotheses %d of 16; max score difference %.2e" % (kk, same, float(np.abs(v[1] - ref[1]).max())))
