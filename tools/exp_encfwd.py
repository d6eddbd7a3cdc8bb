"""Experiment: time the stand-alone encoder forward operator (vag_bigru_seq_fwd through the autograd path) eagerly and
from a graph, and list its kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
c = bench.CFG2
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
with torch.no_grad():
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m._encode(src, lt, None)
        torch.cuda.synchronize(); print("eager encode %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    print("graph encode %.3f ms" % (bench._time_graph(lambda: m._encode(src, lt, None)) * 1e3), flush=True)
    print("graph encode again %.3f ms" % (bench._time_graph(lambda: m._encode(src, lt, None)) * 1e3), flush=True)
