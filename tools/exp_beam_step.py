"""Isolated timing of one beam expansion (vag_beam_step at step di = 5: stage 1 + stage 2) at the configs[3] shape,
replayed from a graph of 50 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import _lib as L
from vagnmt_hip._lib import call, ptr
dev = torch.device("cuda:0")
B, k, V, H, ML = 16, 12, 9391, 512, 80
ldl = (V + 3) // 4 * 4
logp = torch.log_softmax(torch.randn(B * k, ldl, device=dev), -1).contiguous()
nll = torch.randn(B * k, device=dev)
beam = torch.randint(4, V, (2 * ML, B, k), device=dev, dtype=torch.int64)
h_in = torch.randn(B * k, H, device=dev); h_out = torch.empty_like(h_in)
n_alive = torch.zeros(1, dtype=torch.int32, device=dev)
scratch = torch.empty(L.lib().vag_beam_scratch_bytes(B, k, V, ML), dtype=torch.uint8, device=dev)
def one():
    for _ in range(50):
        call("vag_beam_step", ptr(logp), ldl, ptr(nll), ptr(beam, torch.int64), 5, ML, ptr(h_in), ptr(h_out), B, k, V, H,
             ptr(n_alive, torch.int32), scratch.data_ptr(), L.stream())
t = bench._time_graph(one, reps=10)
print("vag_beam_step (stage 1 + stage 2), B=16 k=12 V=9391: %.2f us per call" % (t / 50 * 1e6))
