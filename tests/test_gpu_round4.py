"""GPU, round 4: (1) a void gradient is never applied -- a persistent recurrence kernel that gives up a wait, or a non-finite
gradient norm, makes the optimiser skip the step ON THE DEVICE and is reported (VERDICT r3 weak 2 / ADVICE r3 medium;
semantics kept for every applied step: train.py:44-49); (2) the data-parallel phased sequence with the persistent kernels
inside its phase graphs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu

DIMS = (300, 333, 64, 32, 512, 48)       # Vs, Vt, I, E, H, S: H = 512 / B = 64 is what the persistent kernels take


def _model(seed=0):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    Vs, Vt, I, E, H, S = DIMS
    torch.manual_seed(seed)
    return NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda()


def _batch(seed, B=64, Ts=12, Tt=5):
    Vs, Vt, I = DIMS[:3]
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(4, Vs, (B, Ts), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    return src.cuda(), [Ts] * B, tgt.cuda(), torch.randn(B, I, generator=g).abs().cuda()


def _driver(seed=0, **kw):
    from machine_translation_vision.losses import PairwiseRankingLoss
    from vagnmt_hip.trainer import TrainStep
    m = _model(seed)
    vw = torch.ones(DIMS[1], device="cuda")
    vw[0] = 0
    return m, TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), teacher_force_ratio=1.0, **kw)


def _state(ts):
    torch.cuda.synchronize()
    return (ts.fp.flat.clone(), ts.fp.m.clone(), ts.fp.v.clone(), int(ts.step_count.item()))


def test_persistent_kernels_are_what_this_shape_runs():
    from vagnmt_hip import _lib as L
    for kind in (0, 1):
        assert L.lib().vag_recurrence_supported(kind, 64, 12, 5, 512) == 1


@pytest.mark.parametrize("use_graph", [False, True])
def test_forced_give_up_skips_the_step_on_the_device_and_is_reported(use_graph):
    """Spin limit 1: every wait of the four persistent kernels gives up at once, their results are garbage.  The step must not
    touch parameters, moments or the step counter; it must leave the gradient buffer zeroed, report NaN as the norm, count
    one skipped step and make check() raise.  With the limit restored the next steps are healthy and change the weights."""
    from vagnmt_hip import _lib as L
    m, ts = _driver(use_graph=use_graph)
    for s in range(3):                                  # healthy steps (the third one replays the captured graph)
        ts.step(*_batch(10 + s), teacher=True)
    assert L.lib().vag_persistent_timeouts() == 0 and ts.skipped_steps() == 0
    before = _state(ts)
    L.set_option("persist_spin_limit", 1)
    try:
        m2, ts_bad = None, ts
        if use_graph:
            ts_bad._graphs.clear()                      # the limit is a kernel argument: captured graphs hold the old one
            ts_bad._seen.clear()
            ts.step(*_batch(20), teacher=True)          # eager visit of the shape
            n_eager = 1
        else:
            n_eager = 0
        ts.step(*_batch(21), teacher=True)              # graph capture + replay (or eager)
        torch.cuda.synchronize()
    finally:
        L.set_option("persist_spin_limit", 0)
    after = _state(ts)
    for a, b in zip(before[:3], after[:3]):
        assert torch.equal(a, b)
    assert before[3] == after[3]
    assert float(ts.fp.grad.abs().max()) == 0.0
    assert torch.isnan(ts.grad_norm).all()
    assert ts.skipped_steps() == 1 + n_eager
    with pytest.raises(L.VagError):
        ts.check()
    assert L.lib().vag_persistent_timeouts() == 0       # check() read and reset the count
    ts._graphs.clear()
    ts._seen.clear()
    for s in range(3):
        out = ts.step(*_batch(30 + s), teacher=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]) and torch.isfinite(ts.grad_norm).all()
    assert int(ts.step_count.item()) == before[3] + 3
    assert not torch.equal(ts.fp.flat, before[0])
    assert ts.skipped_steps() == 1 + n_eager
    ts.check()


def test_give_up_reaches_the_gradient_buffer_for_the_all_reduce():
    """Data parallelism: a give-up on ONE replica must make EVERY replica skip.  The last launch of the backward pass turns
    the give-up word into a non-finite entry of the flat gradient (the padding row of the encoder embedding, which no other
    kernel writes); the sum all-reduce spreads it and each replica's norm pass sees it."""
    from vagnmt_hip import _lib as L
    m, ts = _driver(use_graph=False)
    src, lens, tgt, im = _batch(40)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m.train()
    ts.backend.run(src, lt, tgt, im, True, 7)
    torch.cuda.synchronize()
    g00 = m.encoder.embedding.weight._vag_grad[0, 0]
    assert float(g00) == 0.0                            # healthy: nobody writes the padding row's gradient
    ts.fp.grad.zero_()
    L.set_option("persist_spin_limit", 1)
    try:
        ts.backend.run(src, lt, tgt, im, True, 3)       # the phases of the data-parallel sequence
        ts.backend.run(src, lt, tgt, im, True, 4, reuse=True)
        torch.cuda.synchronize()
    finally:
        L.set_option("persist_spin_limit", 0)
    assert torch.isinf(g00)
    before = _state(ts)
    # a replica that did NOT give up receives the entry through the all-reduce: emulate it by clearing the local word first
    assert L.lib().vag_persistent_timeouts() > 0
    ts._optimizer()                                     # (clears the word as well)
    after = _state(ts)
    assert torch.equal(before[0], after[0]) and before[3] == after[3] and ts.skipped_steps() == 1
    ts.fp.grad[5] = float("inf")                        # no give-up word now: the non-finite norm alone must skip
    ts._optimizer()
    torch.cuda.synchronize()
    assert torch.equal(before[0], ts.fp.flat) and ts.skipped_steps() == 2 and float(ts.fp.grad.abs().max()) == 0.0
    ts.fp.grad[7] = float("nan")
    ts._optimizer()
    torch.cuda.synchronize()
    assert torch.equal(before[0], ts.fp.flat) and ts.skipped_steps() == 3


def _rccl_worker(port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from vagnmt_hip import _lib as L
    res = {"supported": [int(L.lib().vag_recurrence_supported(k, 64, 12, 5, 512)) for k in (0, 1)]}
    for name, kw in (("phased", dict(process_group=dist.group.WORLD, force_phased=True)), ("single", {})):
        m, ts = _driver(seed=7, use_graph=True, **kw)
        losses = [float(ts.step(*_batch(1000 + 10 * (s % 2)), teacher=True)[0]) for s in range(6)]
        torch.cuda.synchronize()
        res[name] = (ts.fp.flat.cpu().numpy().copy(), losses, dict(ts.stats), ts.skipped_steps())
    res["timeouts"] = int(L.lib().vag_persistent_timeouts())
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_phased_sequence_runs_the_persistent_kernels_inside_its_graphs():
    """H = 512 / B = 64: graph A holds the persistent encoder forward, decoder forward and decoder backward, graph B the
    persistent encoder backward, which runs while bucket 0's all-reduce is in flight on RCCL's stream (world 1: what one
    GPU can rehearse).  No wait may give up, no step may be skipped, results equal the single-graph step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29800 + os.getpid() % 2000, q))
    p.start()
    res = q.get(timeout=500)
    p.join(120)
    assert p.exitcode == 0
    assert res["supported"] == [1, 1] and res["timeouts"] == 0
    (fa, la, sa, ka), (fb, lb, sb, kb) = res["phased"], res["single"]
    assert ka == 0 and kb == 0
    assert sa["captures"] >= 2 and sa["replays"] >= 4, sa
    assert np.allclose(la, lb, rtol=2e-4), (la, lb)
    # the persistent decoder adds score shares with fp32 atomics: two runs agree to rounding only, and six Adam steps turn a
    # rounding-level difference of a near-zero gradient entry into a fraction of one update (lr = 4e-4): bound the worst entry
    # by 5 % of one update and the mean by far less
    assert np.allclose(fa, fb, rtol=2e-4, atol=2e-5), np.abs(fa - fb).max()
    assert float(np.abs(fa - fb).mean()) <= 2e-7


@pytest.mark.parametrize("M,N,K", [(192, 9391, 256), (16, 9391, 256), (64, 9391, 256), (97, 4100, 256), (137, 5000, 512), (256, 4096, 768),
                                   (100, 40000, 256), (133, 4097, 320)])
def test_tall_skinny_vocabulary_product_is_fp32_grade(M, N, K):
    """The per-step vocabulary head of decoding / free-running steps (NMT_Decoder.py:143 at one time step) on the tall-skinny
    bf16x6 kernel: against a float64 product, with the error bound of an fp32 dot product of length K (a few ulp of
    sum |a||w|); ragged M (not a multiple of 32), N (not a multiple of 64), K in two LDS chunks."""
    from vagnmt_hip import _lib as L
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * 0.3).cuda()
    b = torch.randn(N, generator=g).cuda()
    ldy = (N + 3) // 4 * 4
    y = torch.full((M, ldy), float("nan"), device="cuda")
    L.call("vag_linear_fwd", M, N, K, L.ptr(x), L.ptr(W), L.ptr(b), 0, L.ptr(y), L.stream())
    torch.cuda.synchronize()
    # vag_linear_fwd writes a dense (M, N) result: read it back with that stride
    got = y.view(-1)[: M * N].view(M, N).double().cpu()
    want = x.double().cpu() @ W.double().cpu().t() + b.double().cpu()
    bound = (x.abs().double().cpu() @ W.abs().double().cpu().t() + b.abs().double().cpu()) * 2.0 ** -21
    assert torch.isfinite(got).all()
    assert bool(((got - want).abs() <= bound).all()), float(((got - want).abs() / bound).max())


def test_step_results_stay_valid_without_a_copy_launch():
    """TrainStep.step hands out views of the step's slot in the device-side result ring (vag_step_cfg.loss_ring): they must
    still hold that step's losses after later steps have run (eager first visits, a capture, replays, both shapes of a
    mixed run), and the host's and the device's execution counts must agree (check())."""
    m, ts = _driver(seed=5)
    kept, seen = [], []
    for i in range(9):
        b = _batch(20 + i % 3, Ts=12 if i % 2 == 0 else 8)
        out = ts.step(*b, teacher=(i % 4 != 3))
        kept.append(out)
        seen.append([float(x) for x in out])        # read at once: the reference's loop does loss.item() here (train.py)
    torch.cuda.synchronize()
    later = [[float(x) for x in out] for out in kept]
    assert later == seen
    assert len({tuple(s) for s in seen}) > 1          # (the steps did produce different losses)
    assert kept[0][0].data_ptr() != kept[1][0].data_ptr()
    ts.check()
    assert ts.backend.f.executed == 9


@pytest.mark.parametrize("B,H,storage", [(130, 512, "f32"), (256, 1024, "f16"), (136, 1024, "f32")])
def test_wide_batch_launch_chain_kernels_match_the_round3_kernels(B, H, storage):
    """Round 4 kernels of the per-step launch chains at wide batches (what configs[4] runs): the score / d-alpha reductions with
    the row-constant operands in registers, their riding products and the query product on 32 x 64 / 32 x 32 tiles.  Against the
    round-2/3 kernels (option attn_dot_reg = 0 keeps them for the reductions and their riding products) on the same step:
    loss and every gradient; B not a multiple of 32, fp32 and 2-byte storage."""
    from machine_translation_vision.losses import PairwiseRankingLoss
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    Vs, Vt, I, E, S, Ts, Tt = 120, 150, 64, 64, 48, 16, 6
    res = []
    for reg in (1, 0):
        assert L.lib().vag_set_option(b"attn_dot_reg", reg) == 0
        try:
            torch.manual_seed(3)
            m = NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda()
            vw = torch.ones(Vt, device="cuda")
            vw[0] = 0
            ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), use_graph=False, pad_src=1,
                           storage=storage)
            m.eval()
            g = torch.Generator().manual_seed(9)
            src = torch.randint(4, Vs, (B, Ts), generator=g).cuda()
            tgt = torch.randint(4, Vt, (B, Tt), generator=g)
            tgt[:, -1] = 3
            im = torch.randn(B, I, generator=g).abs().cuda()
            lt = torch.full((B,), Ts, dtype=torch.int32, device="cuda")
            ts.backend.run(src, lt, tgt.cuda(), im, True, 7)
            torch.cuda.synchronize()
            res.append(([float(x) for x in ts.backend.outputs()],
                        {n: p._vag_grad.detach().cpu().clone() for n, p in m.named_parameters()}))
        finally:
            L.lib().vag_set_option(b"attn_dot_reg", 1)
    (la, ga), (lb, gb) = res
    tol = 2e-5 if storage == "f32" else 2e-3
    assert np.allclose(la, lb, rtol=tol, atol=tol), (la, lb)
    for n in ga:
        err = (ga[n] - gb[n]).abs().max().item()
        assert err <= (3e-5 if storage == "f32" else 5e-3) * max(gb[n].abs().max().item(), 1e-3), (n, err)


def test_beam_search_on_raw_logits_equals_the_log_softmax_path():
    """Beam search whose expansion kernel normalises RAW logits with the log-sum-exp pieces the vocabulary product's epilogue
    leaves (vag_head_logits_step / vag_beam_step_logits_dev; B k = 192 rows, V = 4100: the shape class of configs[3]) against
    the path that materialises log-probabilities first: same hypotheses, same scores; finished hypotheses included."""
    from test_gpu_edge_and_full import make
    from vagnmt_hip import ops
    B, k, Ts, V, L = 16, 12, 9, 4100, 14
    lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=torch.Generator().manual_seed(2))], reverse=True)
    lens[0] = Ts
    m, src, _, im = make(80, V, 64, 256, 64, 48, B, Ts, 3, lens, seed=4)
    with torch.no_grad():
        m.decoder.out.bias[3] += 6.0                  # some hypotheses finish early
    mg = m.cuda().eval()
    assert ops.head_logits_parts_count(mg.decoder.head_params(), B * k, 256, V) == (V + 63) // 64
    res = {}
    for raw in (True, False):
        mg.decode_raw_logits = raw
        hyp = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), k, L)]
        res[raw] = (hyp, mg.last_beam_scores.cpu().numpy().copy())
    assert res[True][0] == res[False][0]
    assert np.allclose(res[True][1], res[False][1], rtol=1e-5, atol=1e-5)
    assert any(len(h) < L - 1 for h in res[True][0])


def test_cached_decoding_tables_follow_the_weights():
    """What decoding derives from the weights alone (stacked decoder matrices, per-token tables) is kept between decode calls.
    It must be rebuilt after an optimiser step of the step driver (which writes the parameters through the flat buffer, unseen by
    torch's version counters) and after an in-place write from outside: decoding with the kept buffers == decoding with the
    caches dropped."""
    m, ts = _driver(seed=8, lr=0.05)
    src, lens, tgt, im = _batch(31, B=16, Ts=9, Tt=5)

    def decode(k):
        m.eval()
        hyp = [[int(t) for t in h] for h in m.beamsearch_decode(src, lens, im, k, 12)]
        sc = m.last_beam_scores.cpu().numpy().copy() if k > 1 else None
        return hyp, sc

    def fresh(k):
        m.__dict__.pop("_decode_wcache", None)
        m.__dict__.pop("_decode_cache", None)
        return decode(k)

    first = decode(3)
    for _ in range(3):
        ts.step(src, lens, tgt, im, teacher=True)            # large learning rate: the weights really move
    got, want = decode(3), fresh(3)
    # (not bitwise: the tables come out of split-K products whose atomics land in a run-dependent order)
    assert got[0] == want[0] and np.allclose(got[1], want[1], rtol=0, atol=1e-5)
    assert not np.allclose(first[1], got[1], rtol=0, atol=1e-2)       # (the steps did change what decoding sees, grossly)
    with torch.no_grad():
        m.decoder.embedding.weight.mul_(1.5)                  # tied output / input embedding: tables and head both depend on it
    stale = got[1]
    got, want = decode(3), fresh(3)
    assert got[0] == want[0] and np.allclose(got[1], want[1], rtol=0, atol=1e-5)
    assert not np.allclose(stale, got[1], rtol=0, atol=1e-2)
    gg, gw = decode(1), fresh(1)                              # the graphed greedy path (E = 32: not the one-launch form)
    assert gg[0] == gw[0]
