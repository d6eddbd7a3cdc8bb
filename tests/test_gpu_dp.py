"""GPU: the data-parallel step driver on the fused HIP back end -- two processes sharing the one GPU of the test box
(gloo transport: RCCL refuses two ranks on one device; the driver's code path is the same, only the process-group
backend differs) against a single-process step on the mean of the two shards' gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu

DIMS = (120, 140, 64, 32, 48, 40)       # Vs, Vt, I, E, H, S
STEPS = 4


def _model(seed):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    Vs, Vt, I, E, H, S = DIMS
    torch.manual_seed(seed)
    return NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda()


def _batch(seed, B=6, Ts=9, Tt=7):
    Vs, Vt, I = DIMS[:3]
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(4, Vs, (B, Ts), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    return src.cuda(), [Ts] * B, tgt.cuda(), torch.randn(B, I, generator=g).abs().cuda()


def _criteria():
    from machine_translation_vision.losses import PairwiseRankingLoss
    vw = torch.ones(DIMS[1], device="cuda")
    vw[0] = 0
    return torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1)


def _worker(rank, world, port, q, use_graph, three=False, zero1=False):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vagnmt_hip.trainer import TrainStep
    m = _model(100 + rank)
    cm, cv = _criteria()
    ts = TrainStep(m, cm, cv, use_graph=use_graph, world_size=world, three_buckets=three, zero1=zero1)
    if zero1:
        lo, hi, size, r = ts._shard()
        assert r == rank and size * world == ts.fp.n_alloc and size % 64 == 0 and 0 <= lo <= hi <= ts.fp.n
    if three:
        assert len(ts.fp.buckets()) == 3
    losses = []
    for step in range(STEPS):
        out = ts.step(*_batch(1000 + 10 * step + rank), teacher=(step % 2 == 0))
        losses.append(float(out[0]))
    torch.cuda.synchronize()
    if zero1:
        # Adam's moments are current on the owner's shard only until gathered (what save_checkpoint does first)
        lo, hi, size, _ = ts._shard()
        own_m = ts.fp.m[lo:hi].clone()
        ts.gather_optimizer_state()
        assert torch.equal(ts.fp.m[lo:hi], own_m) and float(ts.fp.m.abs().max()) > 0.0
        other = ts.fp.m[:lo] if rank == world - 1 else ts.fp.m[hi:]
        assert float(other.abs().max()) > 0.0                   # the other rank's shard arrived
    q.put((rank, ts.fp.flat.cpu().numpy().copy(), losses, dict(ts.stats)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("use_graph,three,zero1", [(False, False, False), (True, False, False), (True, True, False),
                                                   (False, False, True), (True, False, True)])
def test_two_ranks_on_one_gpu_equal_single_process_mean_gradient(use_graph, three, zero1):
    """zero1: TrainStep(zero1=True) -- gradient by reduce-scatter (gloo: an all-reduce with the same sums), sum of squares / clip / Adam
    on each rank's shard of the flat buffers (vag_clip_adam_shard, two phases around an all-reduce of one double), parameters back by
    all-gather -- against the same single-process reference: SURVEY 8e option for train.py:46-49.
    three: the three-bucket cut (TrainStep(three_buckets=True): vag_train_step phases 1|16, 32, 4 with an all-reduce after each)
    against the same single-process reference -- the flat layout differs (vse_imagine.* / decoderini.* form a bucket of their own),
    so the comparison goes parameter by parameter."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000 + (1 if use_graph else 0) + (2 if three else 0) + (4 if zero1 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_graph, three, zero1)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, flat_a, losses_a, stats_a), (_, flat_b, losses_b, _) = res
    assert np.array_equal(flat_a, flat_b)
    assert losses_a != losses_b
    if use_graph:
        assert stats_a["captures"] >= (1 if zero1 else (3 if three else 2)) and stats_a["replays"] >= 2, stats_a      # every phase graph of a shape
    # single process: dropout is on in both runs (train mode) but the model here has p = 0 everywhere
    from vagnmt_hip.trainer import TrainStep
    m = _model(100)
    cm, cv = _criteria()
    ts = TrainStep(m, cm, cv, use_graph=False, three_buckets=three)      # (same flat layout as the ranks: the comparison is flat)
    ts.world = 2                                    # 1/world folded into clip+Adam, as on the ranks
    m.train()
    for step in range(STEPS):
        for rank in range(2):
            src, lens, tgt, im = _batch(1000 + 10 * step + rank)
            lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
            ts.backend.run(src, lt, tgt, im, step % 2 == 0, 7)          # gradients of both shards accumulate
            if rank == 0:
                assert abs(float(ts.backend.outputs()[0]) - losses_a[step]) <= 2e-4 * abs(losses_a[step])
        ts._optimizer()
    torch.cuda.synchronize()
    ref = ts.fp.flat.cpu().numpy()
    assert np.allclose(flat_a, ref, rtol=2e-4, atol=2e-6), np.abs(flat_a - ref).max()


def _rccl_worker(port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from vagnmt_hip.trainer import TrainStep
    res = {}
    for name, kw in (("phased", dict(process_group=dist.group.WORLD, force_phased=True)), ("single", {})):
        m = _model(100)
        cm, cv = _criteria()
        ts = TrainStep(m, cm, cv, use_graph=True, **kw)
        losses = [float(ts.step(*_batch(1000 + 10 * (s % 2)), teacher=True)[0]) for s in range(6)]
        torch.cuda.synchronize()
        res[name] = (ts.fp.flat.cpu().numpy().copy(), losses, dict(ts.stats))
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_phased_sequence_with_rccl_collectives_world1():
    """The data-parallel sequence -- phase graph, async all-reduce of the first bucket on RCCL's stream, encoder-phase graph,
    second all-reduce, optimiser -- on the real "nccl" (= RCCL) backend with one rank: what can be rehearsed of the
    multi-GPU path on a one-GPU box (graph capture next to RCCL's watchdog thread, stream ordering of the async work)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29700 + os.getpid() % 2000, q))
    p.start()
    res = q.get(timeout=500)
    p.join(120)
    assert p.exitcode == 0
    (fa, la, sa), (fb, lb, sb) = res["phased"], res["single"]
    assert sa["captures"] >= 2 and sa["replays"] >= 4, sa
    assert np.allclose(la, lb, rtol=2e-4), (la, lb)
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-6), np.abs(fa - fb).max()


def _vag_comm_worker(q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    from vagnmt_hip.trainer import TrainStep
    from vagnmt_hip.comm import Comm
    from vagnmt_hip import _lib
    res = {}
    comm = Comm(rank=0, world_size=1)
    res["size"] = int(_lib.lib().vag_comm_size(comm._h))
    for name, kw in (("phased", dict(comm=comm, force_phased=True)), ("single", {})):
        m = _model(100)
        cm, cv = _criteria()
        ts = TrainStep(m, cm, cv, use_graph=True, **kw)
        losses = [float(ts.step(*_batch(1000 + 10 * (s % 2)), teacher=True)[0]) for s in range(6)]
        torch.cuda.synchronize()
        res[name] = (ts.fp.flat.cpu().numpy().copy(), losses)
    comm.close()
    # a plain buffer: the sum over one rank leaves it unchanged
    comm = Comm(rank=0, world_size=1)
    x = torch.randn(1 << 20, device="cuda")
    y = x.clone()
    comm.all_reduce(x).wait()
    torch.cuda.synchronize()
    res["identity"] = bool(torch.equal(x, y))
    comm.close()
    q.put(res)


@pytest.mark.timeout(600)
def test_phased_sequence_with_vag_comm_world1():
    """The same data-parallel sequence with the exchange going through the C ABI's own RCCL communicator (vag_comm_init /
    vag_comm_allreduce on a side stream, include/vag_nmt.h) instead of torch.distributed: one rank, so the sum is the
    identity and the result must equal the single-process step; covers id creation, communicator binding, the event
    hand-offs between the step's stream and the exchange stream, and destruction.  In a process of its own, like the
    torch.distributed RCCL test above: RCCL and its helper threads stay out of the test runner."""
    if os.environ.get("VAG_TEST_COMM_INPROC") == "1":          # rehearsal of the in-process use (a trainer's own process)
        import queue
        q = queue.Queue()
        _vag_comm_worker(q)
        res = q.get()
    else:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        p = ctx.Process(target=_vag_comm_worker, args=(q,))
        p.start()
        res = q.get(timeout=500)
        p.join(120)
        assert p.exitcode == 0
    assert res["size"] == 1 and res["identity"]
    (fa, la), (fb, lb) = res["phased"], res["single"]
    assert np.allclose(la, lb, rtol=2e-4), (la, lb)
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-6), np.abs(fa - fb).max()


def test_zero1_at_world_1_is_the_replicated_step():
    """One rank: the shard is the whole buffer, no exchange -- vag_clip_adam_shard's two phases must do what vag_clip_adam_flat does
    (parameters, moments, step counter, reported norm to the run-to-run bound of the atomics), incl. a skipped (non-finite) step."""
    from vagnmt_hip.trainer import TrainStep
    res = {}
    for zero1 in (False, True):
        m = _model(100)
        cm, cv = _criteria()
        ts = TrainStep(m, cm, cv, use_graph=False, zero1=zero1)
        norms = []
        for s_ in range(4):
            if s_ == 2:
                ts.fp.grad[11] = float("inf")               # a void gradient: the step must be skipped on the device
            ts.step(*_batch(1000 + 10 * s_), teacher=True)
            norms.append(float(ts.grad_norm[0]))
        torch.cuda.synchronize()
        res[zero1] = (ts.fp.flat.clone(), ts.fp.m.clone(), ts.fp.v.clone(), int(ts.step_count.item()), norms, ts.skipped_steps(),
                      float(ts.fp.grad.abs().max()))
    a, b = res[False], res[True]
    # (the backward pass adds with atomics: two runs of the SAME driver differ in the last bits, so "equal" is the run-to-run bound)
    for x, y, what in ((a[0], b[0], "parameters"), (a[1], b[1], "exp_avg"), (a[2], b[2], "exp_avg_sq")):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-7), (what, float((x - y).abs().max()))
    assert a[3] == b[3] == 3 and a[5] == b[5] == 1 and a[6] == b[6] == 0.0
    assert np.isnan(a[4][2]) and np.isnan(b[4][2])
    assert np.allclose([a[4][0], a[4][1], a[4][3]], [b[4][0], b[4][1], b[4][3]], rtol=1e-5)
