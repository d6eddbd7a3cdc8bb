"""Beam-12 decode step, fused attention + gru_2 launch (attn_row_gru_kernel) against the two launches it replaces, by source length."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
dev = torch.device("cuda:0")
for Ts in (12, 16, 24, 32, 40):
    c = dict(bench.CFG2, Ts=Ts, B=16)
    m = bench.build_model(c, dev).eval()
    src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=False)
    row = []
    for mode in (1, 0):
        L.set_option("attn_row", mode)
        m.__dict__.pop("_decode_cache", None)          # the captured decode graphs hold the other mode's launches
        for _ in range(3): m.beamsearch_decode(src, lens, im, 12, 80)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.beamsearch_decode(src, lens, im, 12, 80)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        row.append(dt * 1e6 / m.last_decode_steps)
    L.set_option("attn_row", 1)
    print("Ts=%d  fused %.1f us/step  separate %.1f us/step" % (Ts, row[0], row[1]), flush=True)
