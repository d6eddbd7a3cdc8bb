"""Where a step of the free-running form of the persistent decoder kernel spends its time (lab build: 100 MHz timestamps of
workgroup 0, vag_set_option("dec_stamps", address of (Tt + 1) x 16 words)).  A pass t of the kernel: wait h2[t-1] | products of
gru_1 and the head's W1 | head of step t-1: hidden layer + wait | logits rounds | candidates + wait | token | gru_1 cell,
phases 2 and 4 as in the teacher-forced form.
Usage (GPU box): python tools/exp_free_phases.py > profiles/r04_exp_free_phases.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, bench
from vagnmt_hip import _lib as L
L.use_lab_build()
from vagnmt_hip import ops
from test_gpu_edge_and_full import make
c = bench.CFG2
B, Ts, Tt, V = c["B"], c["Ts"], c["Tt"], c["V"]
m, src, tgt, im = make(300, V, 64, 256, 512, 48, B, Ts, 3, [Ts] * B, seed=1)
mg = m.cuda().eval()
dec = mg.decoder
g = torch.Generator().manual_seed(7)
enc = (torch.randn(B, Ts, 1024, generator=g) * 0.5).cuda()
mask = torch.ones(B, Ts).cuda()
h0 = (torch.randn(B, 512, generator=g) * 0.5).cuda()
st = torch.zeros((Tt + 1) * 16, dtype=torch.int64, device="cuda")
ldl = (V + 3) // 4 * 4
with torch.no_grad():
    pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
    for it in range(3):
        if it == 2:
            L.set_option("dec_stamps", st.data_ptr())
        tok = torch.zeros(Tt + 1, B, dtype=torch.int64, device="cuda"); tok[0] = 2
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), free_run=True, head=dec.head_params(),
                            p_out=0.0, rng=None, V=V, ldl=ldl)
        e1.record(); torch.cuda.synchronize()
        print("free-running forward operator (tables + kernel + contexts + gather): %.1f us" % (e0.elapsed_time(e1) * 1e3))
L.set_option("dec_stamps", 0)
s = st.cpu().numpy().reshape(Tt + 1, 16).astype(np.float64) * 0.01
names = ["wait h2", "gru_1 + W1 products", "head hidden + wait", "logits rounds", "candidates + wait", "token", "gru_1 cell + publish",
         "wait h1", "phase2+scores", "atomic drain", "(arrive)", "wait sc", "phase4"]
rows = []
for t in range(2, Tt - 1):
    a, nxt = s[t], s[t + 1][0]
    rows.append([a[1] - a[0], a[8] - a[1], a[9] - a[8], a[10] - a[9], a[11] - a[10], a[12] - a[11], a[2] - a[12], a[3] - a[2],
                 a[4] - a[3], a[5] - a[4], a[6] - a[5], a[7] - a[6], nxt - a[7]])
r = np.array(rows)
print("per step (us), mean / median over steps 2..%d of workgroup 0:" % (Tt - 2))
for n, mm, md in zip(names, r.mean(0), np.median(r, 0)):
    print("  %-22s %6.2f  %6.2f" % (n, mm, md))
print("  total                  %6.2f" % r.sum(1).mean())
