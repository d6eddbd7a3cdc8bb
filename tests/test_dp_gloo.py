"""CPU, world_size 2 over gloo: the data-parallel bookkeeping of the REAL step driver (vagnmt_hip.trainer.TrainStep):
broadcast of the replica, flat gradient buffer split into the early / encoder buckets, one async all-reduce per bucket
issued between the backward phases, 1/world folded into clip+Adam, equal batch counts per rank.  The HIP kernels need a
GPU, so the compute back end is replaced by the CPU oracle through the driver's injection point (`backend=`); the
`-m gpu` twin in test_gpu_dp.py runs the same check on the fused HIP back end with two processes on one GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG

DIMS = dict(Vs=50, Vt=60, I=96, E=16, H=24, S=20)


class OracleBackend:
    """Stand-in for the HIP back end: gradients from the CPU oracle, written into the driver's flat gradient buffer in
    the same two phases (everything but the encoder, then the encoder), optimiser = the oracle's clip + Adam on the
    flat buffers.  Records what the driver asked for."""
    phased = True

    def __init__(self):
        self.calls = []
        self.state = {}

    def bind(self, ts):
        self.ts = ts

    def run(self, src, lengths, tgt, im, teacher, phases, reuse=False):
        from oracle import vag_oracle as O
        self.calls.append(("run", phases))
        ts = self.ts
        if phases & 3:
            leaves = {n: p.detach().clone().requires_grad_(True) for n, p in ts.fp.named}
            out = O.model_forward(leaves, src, [int(x) for x in lengths], tgt, im, teacher=teacher)
            out["loss"].backward()
            self._grads = {n: (v.grad if v.grad is not None else torch.zeros_like(v)) for n, v in leaves.items()}
            self._out = (out["loss"].detach(), out["loss_mt"].detach(), out["loss_vse"].detach())
        for n, p in ts.fp.named:
            late = n.startswith("encoder.")
            if (late and (phases & 4)) or (not late and (phases & 2)):
                p._vag_grad.add_(self._grads[n])

    def outputs(self):
        return self._out

    def optimizer(self):
        from oracle import vag_oracle as O
        self.calls.append(("opt", None))
        ts = self.ts
        grads = {n: p._vag_grad.detach().clone() / ts.world for n, p in ts.fp.named}
        total, cg = O.clip_grad_norm(grads, ts.clip)
        ts.grad_norm[0] = total
        with torch.no_grad():
            new = O.adam_step({n: p.detach() for n, p in ts.fp.named}, cg, self.state, lr=ts.lr, weight_decay=ts.wd)
            for n, p in ts.fp.named:
                p.copy_(new[n])
            ts.fp.grad.zero_()


def _batch(seed, B=3, Ts=5, Tt=4):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(4, DIMS["Vs"], (B, Ts), generator=g)
    tgt = torch.randint(4, DIMS["Vt"], (B, Tt), generator=g)
    tgt[:, -1] = 3
    im = torch.randn(B, DIMS["I"], generator=g).abs()
    return src, [Ts] * B, tgt, im


def _model(seed):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    torch.manual_seed(seed)
    d = DIMS
    return NMT_AttentionImagine_Seq2Seq_Beam_V11(d["Vs"], d["Vt"], d["I"], d["E"], d["E"], d["H"], d["S"], 0.99, tied_emb=True)


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vagnmt_hip.trainer import TrainStep
    m = _model(100 + rank)                            # replicas start DIFFERENT ...
    be = OracleBackend()
    ts = TrainStep(m, None, None, use_graph=False, world_size=world, backend=be)      # ... the driver broadcasts rank 0's
    be.bind(ts)
    flat0 = ts.fp.flat.numpy().copy()
    outs = []
    for step in range(2):
        outs.append(float(ts.step(*_batch(1234 + 10 * step + rank), teacher=True)[0]))
    q.put((rank, flat0, ts.fp.flat.numpy().copy(), be.calls, outs, float(ts.grad_norm[0]), ts.fp.early_end, ts.fp.n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trainstep_world2_equals_single_process_on_the_mean_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import numpy as np
    (_, f0a, f1a, calls_a, outs_a, gn_a, early, n), (_, f0b, f1b, calls_b, outs_b, gn_b, _, _) = res
    assert np.array_equal(f0a, f0b)                                   # identical replicas after the broadcast
    assert np.array_equal(f1a, f1b) and gn_a == gn_b                  # and after two optimiser steps
    assert outs_a != outs_b                                           # different shards
    # phases as the driver must issue them: fwd + decoder-side bwd, then the encoder's bwd, then the optimiser
    assert calls_a == [("run", 3), ("run", 4), ("opt", None)] * 2
    assert 0 < early < n
    # single-process reference: the same two steps on the mean of the two ranks' gradients (oracle throughout)
    from vagnmt_hip.trainer import TrainStep
    from oracle import vag_oracle as O
    m = _model(100)
    ts = TrainStep(m, None, None, use_graph=False)
    assert np.array_equal(ts.fp.flat.numpy(), f0a)
    state = {}
    for step in range(2):
        P = {n_: p.detach().clone() for n_, p in ts.fp.named}
        gsum = None
        for rank in range(2):
            src, lens, tgt, im = _batch(1234 + 10 * step + rank)
            leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            O.model_forward(leaves, src, lens, tgt, im, teacher=True)["loss"].backward()
            g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
            gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
        _, cg = O.clip_grad_norm({k: v / 2 for k, v in gsum.items()}, 1.0)
        new = O.adam_step(P, cg, state, lr=4e-4, weight_decay=1e-5)
        with torch.no_grad():
            for n_, p in ts.fp.named:
                p.copy_(new[n_])
    assert np.allclose(ts.fp.flat.numpy(), f1a, rtol=1e-5, atol=1e-7)


def test_flat_layout_puts_the_encoder_bucket_last():
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from vagnmt_hip.trainer import FlatParams
    for vse_separate in (False, True):
        m = _model(0)
        fp = FlatParams(m, vse_separate)
        (a0, a1), (b0, b1) = fp.buckets()
        assert a0 == 0 and a1 == b0 and b1 == fp.n
        for n, p in fp.named:
            o = fp.offsets[n]
            assert (o >= b0) == n.startswith("encoder."), n
        # segment table handed to vag_clip_adam_flat: contiguous, covers the buffer, weight decay only on non-bias names
        assert fp.seg_off[0] == 0 and fp.seg_off[-1] == fp.n and len(fp.groups) == len(fp.seg_off) - 1
        for (gname, names, wd, mult) in fp.groups:
            assert all(("bias" in x) != wd for x in names)
            assert mult == (0.5 if (vse_separate and "vse" in gname) else 1.0)


def test_dp_batch_stream_gives_every_rank_the_same_number_of_batches():
    """i % world == rank hands out unequal counts when the number of batches is not a multiple of world --
    the ranks with the extra batch would wait forever in the all-reduce."""
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    from vagnmt_hip.data import shard_batches
    from machine_translation_vision.samplers import BucketBatchSampler
    rs = np.random.RandomState(0)
    lengths = rs.randint(3, 12, size=131)
    for world in (2, 3, 8):
        per_rank = []
        for rank in range(world):
            np.random.seed(7)                                   # the common seed every rank must use
            batches = list(BucketBatchSampler(lengths, 16))
            per_rank.append(shard_batches(batches, rank, world))
        counts = [len(x) for x in per_rank]
        assert len(set(counts)) == 1 and counts[0] > 0, counts
        flat = [tuple(b) for r in per_rank for b in r]
        assert len(set(flat)) == len(flat)                      # no batch is given to two ranks
