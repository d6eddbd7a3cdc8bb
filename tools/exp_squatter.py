"""What happens to the persistent encoder-backward kernel (256 workgroups, one per CU, all must be resident) when a kernel
with a collective's footprint runs beside it?  (VERDICT r3 weak 8 / DESIGN section 6: in the data-parallel sequence graph B
runs while bucket 0's all-reduce is in flight.)

Phase 4 of vag_train_step (persistent encoder backward + its weight-gradient products) at configs[1] size, eager, timed
with HIP events on its stream; the squatter (tools/squatter.hip) on a side stream, launched BEFORE or AFTER phase 4 is
enqueued.  Reports phase-4 time, the encoder-backward kernel's own time (vag_recurrence_time) and the give-up count.
Usage (GPU box): python tools/exp_squatter.py > profiles/r04_exp_squatter.txt"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from vagnmt_hip import _lib as L  # noqa: E402

so = os.path.join(ROOT, "tools", "libsquatter.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "squatter.hip"), "-o", so])
SQ = C.CDLL(so)
SQ.squat.restype = C.c_int
SQ.squat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]

dev = torch.device("cuda:0")
c = bench.CFG2
from machine_translation_vision.losses import PairwiseRankingLoss  # noqa: E402
from vagnmt_hip.trainer import TrainStep  # noqa: E402
m = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev)
vw[0] = 0
ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), teacher_force_ratio=1.0,
               use_graph=False)
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
m.train()
side = torch.cuda.Stream()
sink = torch.zeros(4, device=dev)
L.set_option("persist_timing", 1)


def read_rec(kind):
    ms, n = C.c_double(0), C.c_int(0)
    L.lib().vag_recurrence_time(kind, C.byref(ms), C.byref(n))
    return (ms.value / max(n.value, 1)) * 1e3


def run(label, squat=None, when="before", reps=5):
    """squat = (workgroups, threads, lds bytes, microseconds)"""
    t4, k4, tq = [], [], []
    for _ in range(reps):
        ts.fp.grad.zero_()
        ts.backend.run(src, lt, tgt, im, True, 3)
        torch.cuda.synchronize()
        read_rec(2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if squat and when == "before":
            q0.record(side)
            SQ.squat(side.cuda_stream, squat[0], squat[1], squat[2], float(squat[3]), sink.data_ptr())
            q1.record(side)
        e0.record(main)
        ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
        e1.record(main)
        if squat and when == "after":
            q0.record(side)
            SQ.squat(side.cuda_stream, squat[0], squat[1], squat[2], float(squat[3]), sink.data_ptr())
            q1.record(side)
        torch.cuda.synchronize()
        if squat:
            tq.append(q0.elapsed_time(q1) * 1e3)
        t4.append(e0.elapsed_time(e1) * 1e3)
        k4.append(read_rec(2))
    to = L.lib().vag_persistent_timeouts()
    t4.sort(), k4.sort(), tq.sort()
    print("%-64s phase 4 %7.1f us (min %7.1f)   enc_bwd kernel %7.1f us   squatter itself %7.1f us   give-ups %d" %
          (label, t4[len(t4) // 2], t4[0], k4[len(k4) // 2], tq[len(tq) // 2] if tq else 0.0, to), flush=True)


print("# persistent encoder backward beside a squatting kernel; configs[1] (B=64, Ts=40, H=512), eager launches, median of 5")
run("alone")
for wgs, thr, lds, us in ((64, 256, 16384, 400), (64, 512, 16384, 400), (256, 256, 16384, 400), (64, 256, 98304, 400),
                          (64, 1024, 65536, 400), (32, 512, 32768, 2000)):
    for when in ("before", "after"):
        run("squatter %3d wg x %4d thr, %3d KB LDS, %4d us, launched %-6s" % (wgs, thr, lds // 1024, us, when),
            (wgs, thr, lds, us), when)
run("alone (again)")
# the alternative under data parallelism: the encoder backward as a launch chain (80 launches of 128-256 workgroups)
L.set_option("persistent", 0)
print("# launch chain instead of the persistent kernel (set_option persistent 0); 'enc_bwd kernel' is not timed there")
run("chain, alone")
for wgs, thr, lds, us in ((64, 256, 16384, 400), (64, 512, 16384, 400), (64, 1024, 65536, 400)):
    for when in ("before", "after"):
        run("chain, squatter %3d wg x %4d thr, %3d KB LDS, %4d us, launched %-6s" % (wgs, thr, lds // 1024, us, when),
            (wgs, thr, lds, us), when)
L.set_option("persistent", 1)
ts.fp.grad.zero_()
