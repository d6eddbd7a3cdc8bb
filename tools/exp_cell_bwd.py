"""Isolated timing of the backward cell kernel (vag_gru_cell_bwd, configs[1] shape) -- 100 launches per graph -- and a
check of its outputs against a run with the library's default path (VAG_SKINNY_FL toggles the full-line variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip._lib import call, ptr, stream
dev = torch.device("cuda:0")
B, H = 64, 512
torch.manual_seed(0)
dgh = torch.randn(B, 3 * H, device=dev); wt = torch.randn(H, 3 * H, device=dev) / 30
carry = torch.randn(B, H, device=dev); d_out = torch.randn(B, H, device=dev)
sv = torch.rand(4, B, H, device=dev) * 0.8 + 0.1; hp = torch.randn(B, H, device=dev)
dgi = torch.empty(B, 3 * H, device=dev); dgh_o = torch.empty(B, 3 * H, device=dev); cout = torch.empty(B, H, device=dev)
def cells():
    for _ in range(100):
        call("vag_gru_cell_bwd", ptr(dgh), ptr(wt), ptr(carry), ptr(d_out), ptr(sv), ptr(hp), B, H, ptr(dgi), ptr(dgh_o), ptr(cout), stream())
t = bench._time_graph(cells) / 100
# reference: dh = dgh @ wt^T + carry (then the cell backward): check dh_direct = dh * z for inactive-free rows
dh = dgh @ wt.t() + carry + d_out
ref = dh * sv[1]
print("FL=%s: %.2f us per launch, max |dh_direct - ref| = %.2e" % (os.environ.get("VAG_SKINNY_FL", "0"), t * 1e6, (cout - ref).abs().max().item()))
