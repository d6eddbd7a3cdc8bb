"""Where a beam-12 decode call of an eval batch spends its wall time (host + device, synchronised between the parts): prologue
(encoder, VSE, initial state), per-call decode state (key projections, tables), step 0, the replayed chunks, finish + copy back;
and the same call unsynchronised (what bench.py's extra.beam12_decode times).  The EOS bias makes hypotheses end early, as a
trained model's do: bench.py prices a step as call time / steps run, so the per-call part weighs more the fewer steps run.
Usage (GPU box): python tools/exp_decode_call_overhead.py [eos_bias ...] >> profiles/r05_exp_beam.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
c = dict(bench.CFG2); c["B"] = 16
dev = torch.device("cuda:0")
src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=True)
for bias in [float(x) for x in (sys.argv[1:] or ["0", "0.6", "2"])]:
    m = bench.build_model(c, dev).eval()
    with torch.no_grad():
        m.decoder.out.bias[3] += bias
    for _ in range(3):
        m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize()
    N = 8
    t0 = time.perf_counter()
    for _ in range(N):
        m.beamsearch_decode(src, lens, im, 12, 80)
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / N
    steps = m.last_decode_steps
    marks = []
    def tick(name):
        torch.cuda.synchronize(); marks.append((name, time.perf_counter()))
    orig_state, orig_prologue = m._decode_state, m._prologue
    def state(*a, **k):
        tick("prologue: encoder, VSE, h0"); r = orig_state(*a, **k); tick("decode state: keys, tables, copies"); return r
    m._decode_state = state
    tot = {}
    for _ in range(N):
        marks.clear()
        tick("start")
        m.beamsearch_decode(src, lens, im, 12, 80)
        tick("step 0 + chunks + finish + copy back")
        for (n0, t0_), (n1, t1) in zip(marks, marks[1:]):
            tot[n1] = tot.get(n1, 0.0) + (t1 - t0_)
    m._decode_state = orig_state
    print("EOS bias %+.1f: %d steps run; unsynchronised call %.0f us = %.1f us per step run" % (bias, steps, free * 1e6, free / steps * 1e6))
    for k, v in tot.items():
        print("    %-42s %8.1f us" % (k, v / N * 1e6))
    rest = tot["step 0 + chunks + finish + copy back"] / N
    print("    -> of the last part, beyond %d x 66 us of replayed steps: %.0f us" % (steps - 1, rest * 1e6 - (steps - 1) * 66))
