"""Where a beam-12 decode call of an eval batch spends its wall time beside the steps themselves (host + device, synchronised
between the parts): prologue (encoder, VSE, initial state), per-call decode state (key projections, tables), step 0, the replayed
chunks, finish + copy back.  Usage (GPU box): python tools/exp_decode_call_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from machine_translation_vision.models import _seq2seq as S
c = dict(bench.CFG2); c["B"] = 16
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
with torch.no_grad():
    m.decoder.out.bias[3] += 2.0
src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=True)
for _ in range(3):
    m.beamsearch_decode(src, lens, im, 12, 80)
marks = []
def tick(name):
    torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
orig_state = m._decode_state
orig_beam = m._beam
def state(*a, **k):
    tick("prologue done / state begins"); r = orig_state(*a, **k); tick("decode state done"); return r
m._decode_state = state
tot = {}
N = 5
for _ in range(N):
    marks.clear()
    tick("start")
    m.beamsearch_decode(src, lens, im, 12, 80)
    tick("end")
    for (n0, t0), (n1, t1) in zip(marks, marks[1:]):
        tot[n1] = tot.get(n1, 0.0) + (t1 - t0)
for k, v in tot.items():
    print("%-34s %8.1f us" % (k, v / N * 1e6))
print("steps run:", m.last_decode_steps)
