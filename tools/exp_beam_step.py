"""The beam expansion alone (beam_stage1 + beam_stage2: V11.py:279-313) on synthetic log-probabilities, by how many hypotheses
have ended (their continuations all tie at score - 1e5 except EOS, V11.py:291-294).  Usage (GPU box): python tools/exp_beam_step.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
from vagnmt_hip import _lib as L
from vagnmt_hip._lib import ptr, call, stream
B, k, V, H, ML = 16, 12, 9391, 512, 80
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
logp = torch.log_softmax(torch.randn(B * k, V, generator=g) * 2, -1).to(dev)
ldl = V
scratch = torch.empty(L.lib().vag_beam_scratch_bytes(B, k, V, ML), dtype=torch.uint8, device=dev)
h = torch.randn(B * k, H, device=dev); h2 = torch.empty_like(h)
n_alive = torch.zeros(1, dtype=torch.int32, device=dev)
for frac in (0.0, 0.25, 0.5, 0.9, 1.0):
    beam = torch.randint(4, V, (2 * ML, B, k), dtype=torch.int64, generator=g).to(dev)
    nfin = int(round(frac * k))
    beam[4, :, :nfin] = 3                        # previous words of step 5: the first nfin hypotheses of every sentence have ended
    nll = (-torch.rand(B, k, generator=g) * 20).to(dev)
    def step():
        call("vag_beam_step", ptr(logp), ldl, ptr(nll.clone()), ptr(beam, torch.int64), 5, ML, ptr(h), ptr(h2), B, k, V, H,
             ptr(n_alive, torch.int32), scratch.data_ptr(), stream())
    for _ in range(5): step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    N = 100
    nl = [nll.clone() for _ in range(N)]
    s.record()
    for i in range(N):
        call("vag_beam_step", ptr(logp), ldl, ptr(nl[i]), ptr(beam, torch.int64), 5, ML, ptr(h), ptr(h2), B, k, V, H,
             ptr(n_alive, torch.int32), scratch.data_ptr(), stream())
    e.record(); torch.cuda.synchronize()
    print("ended hypotheses per sentence %2d of %d: stage 1 + stage 2 = %.1f us per step" % (nfin, k, s.elapsed_time(e) / N * 1e3), flush=True)
