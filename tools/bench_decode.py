"""Config 4: beam-search decode (beam 12, eval batch 16, max_length 80) and greedy decode throughput, cfg2 model."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
c = dict(bench.CFG2); c["B"] = 16
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=True)
for k, L in ((1, 80), (12, 80)):
    for _ in range(2):
        m.beamsearch_decode(src, lens, im, k, L)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 5
    for _ in range(n):
        out = m.beamsearch_decode(src, lens, im, k, L)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("beam=%2d max_len=%d batch=16: %.1f ms per batch = %.0f sentences/s (%.3f ms per decode step), mean hyp len %.1f"
          % (k, L, dt * 1e3, 16 / dt, dt * 1e3 / L, sum(len(h) for h in out) / 16.0))
