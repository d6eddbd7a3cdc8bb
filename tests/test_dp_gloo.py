"""CPU, world_size 2 over gloo: the data-parallel bookkeeping of the step driver (flat buffers, broadcast of the
replica, sum all-reduce + 1/N scaling before clip+Adam).  The HIP kernels themselves need a GPU; here the update
applied to the averaged gradient is the oracle's clip+Adam."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip.trainer import FlatParams
    from oracle import vag_oracle as O
    torch.manual_seed(100 + rank)                     # replicas start DIFFERENT ...
    m = NMT_AttentionImagine_Seq2Seq_Beam_V11(50, 60, 96, 16, 16, 24, 20, 0.99, tied_emb=True)
    fp = FlatParams(m)
    dist.broadcast(fp.flat, src=0)                    # ... and are made identical (TrainStep.__init__)
    P0 = {n: p.detach().clone() for n, p in m.named_parameters()}
    # each rank gets its own shard -> its own gradient (oracle on CPU stands in for the HIP backward)
    g = torch.Generator().manual_seed(1234 + rank)
    src = torch.randint(4, 50, (3, 5), generator=g)
    tgt = torch.randint(4, 60, (3, 4), generator=g)
    tgt[:, -1] = 3
    im = torch.randn(3, 96, generator=g).abs()
    leaves = {n: p.detach().clone().requires_grad_(True) for n, p in P0.items()}
    O.model_forward(leaves, src, [5, 5, 5], tgt, im)["loss"].backward()
    fp.grad.zero_()
    for n, p in m.named_parameters():
        p._vag_grad.copy_(leaves[n].grad)
    local = fp.grad.clone()
    dist.all_reduce(fp.grad, op=dist.ReduceOp.SUM)    # TrainStep._allreduce
    # numpy copies: pickled by value, so the parent can read them after this process has exited
    q.put((rank, fp.flat.numpy().copy(), local.numpy().copy(), fp.grad.numpy().copy(),
           {n: p.grad.numpy().copy() for n, p in m.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_allreduce_of_flat_gradient_world2():
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
    (_, flat0, loc0, sum0, views0), (_, flat1, loc1, sum1, _) = res
    assert np.array_equal(flat0, flat1)                               # identical replicas after broadcast
    assert not np.allclose(loc0, loc1)                                # different shards, different gradients
    assert np.allclose(sum0, loc0 + loc1, atol=1e-6) and np.array_equal(sum0, sum1)
    # parameter .grad views alias the flat buffer: what the fused clip+Adam then scales by 1/world
    n0 = next(iter(views0))
    assert np.abs(views0[n0]).sum() > 0
