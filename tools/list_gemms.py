"""Print every vag_gemm launch of one cfg2 training step (vag_set_option("gemm_debug")) with the cost model's estimate."""
import os, sys, re, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("LIST_GEMMS_CHILD") != "1":
    env = dict(os.environ, LIST_GEMMS_CHILD="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True).stderr
    lines = [l for l in out.splitlines() if l.startswith("[vag_gemm]")]
    # the second step's launches: split on the marker printed between steps
    agg = collections.OrderedDict()
    for l in lines[len(lines) // 2:]:
        agg[l] = agg.get(l, 0) + 1
    tot = 0.0
    for l, n in agg.items():
        us = float(re.search(r"model=([0-9.]+)", l).group(1))
        tot += us * n
        print("%2dx %s" % (n, l[11:]))
    print("launches %d, model total %.1f us" % (sum(agg.values()), tot))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip.trainer import TrainStep
from vagnmt_hip import _lib
_lib.use_lab_build()          # the product library carries no stamp / debug hooks (csrc/Makefile: LAB=1)
_lib.set_option("gemm_debug", 1)
from machine_translation_vision.losses import PairwiseRankingLoss
c = bench.CFG2
dev = torch.device("cuda:0")
m = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), use_graph=False)
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
for _ in range(2):
    ts.step(src, lt, tgt, im, teacher=True)
torch.cuda.synchronize()
