"""configs[4], 2-byte storage: gradient error of every parameter against the fp32-storage run on the same GPU (which the parity
suite pins to the CPU oracle), with the one-plane products on 128 x 128 tiles (gemm_big=0) and on 256 x 256 tiles (gemm_big=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from vagnmt_hip import _lib as L
from machine_translation_vision.losses import PairwiseRankingLoss
from vagnmt_hip.trainer import TrainStep
c = bench.CFG5
dev = torch.device("cuda:0")
batch = bench.make_batch(c, 0, dev)
lt = torch.tensor(batch[1], dtype=torch.int32, device=dev)


def run(storage, state=None):
    m = bench.build_model(c, dev, dropout=False)
    if state is not None:
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(state[n])
    vw = torch.ones(c["V"], device=dev); vw[0] = 0
    ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), use_graph=False, storage=storage)
    if state is not None:
        ts.backend.after_optimizer()
    m.train()
    ts.backend.run(batch[0], lt, batch[2], batch[3], True, 7)
    torch.cuda.synchronize()
    out = [float(x) for x in ts.backend.outputs()]
    g = {n: p._vag_grad.detach().clone().cpu() for n, p in m.named_parameters()}
    st = {n: p.detach().clone() for n, p in m.named_parameters()}
    del ts, m
    torch.cuda.empty_cache()
    return out, g, st


l32, g32, state = run("f32")
res = {}
for big in (0, 1):
    L.set_option("gemm_big", big)
    l16, g16, _ = run("f16", state)
    res[big] = (l16, {n: (g16[n] - g32[n]).abs().max().item() / max(g32[n].abs().max().item(), 1e-3) for n in g32})
L.set_option("gemm_big", 1)
print("losses f32", l32, "| f16 128-tile", res[0][0], "| f16 256-tile", res[1][0])
for n in sorted(g32, key=lambda n: -max(res[0][1][n], res[1][1][n]))[:14]:
    print("%-44s rel err vs fp32 storage: 128-tile %.4f   256-tile %.4f" % (n, res[0][1][n], res[1][1][n]))
