"""The epoch-shaped row of bench.py (extra.stream) under different graph-cache settings: why a bucketed Multi30K-shaped batch stream
runs far below the one-shape headline, and what fixes it.  Usage (GPU box): python tools/exp_stream.py >> profiles/r05_exp_stream.txt"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import trainer as T
dev = torch.device("cuda:0")
orig = T.TrainStep.__init__
def run(tag, **over):
    def init(self, *a, **k):
        k.update(over)
        orig(self, *a, **k)
    T.TrainStep.__init__ = init
    try:
        r = bench.measure_stream(bench.CFG2, dev, eval_batches=2)
    finally:
        T.TrainStep.__init__ = orig
    for e in ("epoch1", "epoch2"):
        v = r[e]
        print("%-34s %s: %6.0f pairs/s  %5.2f ms/step  shapes %3d  captures %3d  evictions %3d  eager %3d  replays %3d  host-in-step %.2f s of %.2f"
              % (tag, e, v["pairs_per_s"], v["ms_per_step"], v["distinct_shapes"], v["captures"], v["evictions"], v["eager_steps"],
                 v["replays"], v["host_seconds_inside_step_calls"], v["seconds"]), flush=True)
which = sys.argv[1:] or ["default", "big", "eager", "big8"]
if "default" in which: run("max_graphs=48 (round 4 default)", max_graphs=48)
if "big" in which: run("max_graphs=1024", max_graphs=1024)
if "eager" in which: run("no graphs (eager launches)", use_graph=False)
if "big8" in which: run("max_graphs=1024, pad_src=8", max_graphs=1024, pad_src=8)
