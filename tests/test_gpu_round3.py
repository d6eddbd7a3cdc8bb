"""GPU, round 3: the persistent recurrence kernels (persist.hip) against the per-step launch chains they replace."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(H, Vs=300, Vt=333, I=64, E=32, S=48, seed=0):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    torch.manual_seed(seed)
    return NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda().eval()


def _batch(B, Ts, Tt, Vs=300, Vt=333, I=64, seed=1, full=False):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=g)], reverse=True)
    lens[0] = Ts
    if full:
        lens = [Ts] * B
    src = torch.zeros(B, Ts, dtype=torch.long)
    for b, L in enumerate(lens):
        src[b, :L] = torch.randint(4, Vs, (L,), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    im = torch.randn(B, I, generator=g).abs()
    return src.cuda(), lens, tgt.cuda(), im.cuda()


@pytest.mark.parametrize("H,B,Ts", [(256, 64, 9), (256, 20, 5), (512, 64, 40), (512, 37, 7), (256, 1, 3), (1024, 16, 6)])
def test_persistent_encoder_equals_launch_chain(H, B, Ts):
    """Forward: encoder states (zeros past each row's length) and, through the saved gates / hidden states, every
    gradient -- the persistent kernel (W_hh as bf16x3 planes in registers, six products) against the chain of per-step
    launches (exact f32-input MFMA).  Ragged lengths, batches that are not a multiple of the 16-row tile."""
    from vagnmt_hip import _lib as L
    from machine_translation_vision.losses import PairwiseRankingLoss
    src, lens, tgt, im = _batch(B, Ts, 4)
    vw = torch.ones(333, device="cuda")
    vw[0] = 0
    crit = torch.nn.NLLLoss(weight=vw, reduction="none")
    res = {}
    for mode in (0, 1):
        L.set_option("persistent", mode)
        try:
            m = _model(H)
            enc, mask = m.encoder(src, lens)
            loss, _, _ = m(src, lens, tgt, im, 1.0, criterion_mt=crit, criterion_vse=PairwiseRankingLoss(0.1))
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (enc.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
        finally:
            L.set_option("persistent", 1)
    e0, l0, g0 = res[0]
    e1, l1, g1 = res[1]
    assert L.lib().vag_persistent_timeouts() == 0
    assert torch.isfinite(e1).all()
    assert (e0 - e1).abs().max().item() <= 2e-6, (e0 - e1).abs().max().item()
    for b, Lb in enumerate(lens):
        if Lb < Ts:
            assert float(e1[Lb:, b].abs().max()) == 0.0
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0))
    for n in g0:
        scale = max(g0[n].abs().max().item(), 1e-3)
        assert (g0[n] - g1[n]).abs().max().item() <= 2e-5 * scale, n


@pytest.mark.parametrize("B,Ts,Tt", [(64, 40, 40), (64, 12, 5), (37, 9, 7), (5, 7, 3), (16, 33, 4), (64, 50, 6)])
def test_persistent_decoder_equals_launch_chain(B, Ts, Tt):
    """Teacher-forced decoder forward in one launch (weights in registers, keys in LDS, four exchanges per step) against
    the 4-launches-per-step chain: per-step outputs through the loss, the saved attention weights through every gradient
    (the backward operators read what the forward saved).  H = 512 is the only hidden size the persistent decoder takes."""
    from vagnmt_hip import _lib as L
    from machine_translation_vision.losses import PairwiseRankingLoss
    src, lens, tgt, im = _batch(B, Ts, Tt, seed=3)
    vw = torch.ones(333, device="cuda")
    vw[0] = 0
    crit = torch.nn.NLLLoss(weight=vw, reduction="none")
    res = {}
    for mode in (0, 1):
        L.set_option("persistent", mode)
        try:
            m = _model(512, seed=2)
            loss, loss_mt, _ = m(src, lens, tgt, im, 1.0, criterion_mt=crit, criterion_vse=PairwiseRankingLoss(0.1))
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (float(loss), float(loss_mt), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
        finally:
            L.set_option("persistent", 1)
    (l0, m0, g0), (l1, m1, g1) = res[0], res[1]
    assert L.lib().vag_persistent_timeouts() == 0
    assert np.isfinite(l1)
    assert abs(l0 - l1) <= 2e-6 * max(1.0, abs(l0)), (l0, l1)
    assert abs(m0 - m1) <= 2e-6 * max(1.0, abs(m0)), (m0, m1)
    for n in g0:
        scale = max(g0[n].abs().max().item(), 1e-3)
        assert (g0[n] - g1[n]).abs().max().item() <= 2e-5 * scale, n


def test_wide_fp16_encoder_kernel_equals_the_launch_chain():
    """2-byte storage mode at configs[4] widths (H = 1024, B = 256, ragged lengths): the one-launch encoder forward
    (enc_fwd_wide16_kernel: fp16 weight slice in registers, states exchanged as fp16) against the chain of per-step launches
    (fp32 states x fp16 weights).  The only difference is the fp16 rounding of the state that enters a step's product
    (2^-12 relative): encoder states within 2e-3 absolute (|h| <= 1), losses within 1e-3."""
    import ctypes as C
    from test_gpu_round2 import _fp16_case
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    m_of, (src, lens, tgt, im), cm, cv = _fp16_case("wide")
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = {}
    try:
        for persistent in (0, 1):
            L.set_option("persistent", persistent)
            m = m_of()
            ts = TrainStep(m, cm, cv, use_graph=False, storage="f16", pad_src=1)
            m.eval()
            ts.backend.run(src, lt, tgt, im, True, 7)
            torch.cuda.synchronize()
            f = ts.backend.f
            B, Ts = src.shape
            c = f.cfg(B, Ts, tgt.shape[1], True, False)
            off = L.lib().vag_step_ws_offset(C.byref(c), 0)
            enc = f.ws[off:off + B * Ts * 2 * 1024].detach().clone()
            out[persistent] = (enc, [float(x) for x in ts.backend.outputs()], ts.fp.grad.detach().clone())
            assert L.lib().vag_persistent_timeouts() == 0
    finally:
        L.set_option("persistent", 1)
    (e0, l0, g0), (e1, l1, g1) = out[0], out[1]
    assert torch.isfinite(e1).all()
    assert (e0 - e1).abs().max().item() <= 2e-3, (e0 - e1).abs().max().item()
    assert (e0 - e1).abs().max().item() > 0.0           # the kernel really ran (fp16-rounded exchange differs in the last bits)
    assert np.allclose(l0, l1, rtol=1e-3, atol=1e-4), (l0, l1)
    assert (g0 - g1).abs().max().item() <= 1e-2 * g0.abs().max().item()


@pytest.mark.parametrize("V", [10243, 40000])
def test_one_pass_log_softmax_for_large_vocabularies(V):
    """lse_nll_kernel<0> (vocabularies above 10240 words: one pass of 16-byte loads with a running (max, sum) pair per
    thread) through the decode step's head (layers/NMT_Decoder.py:137-143): log-probabilities and arg-max against torch,
    incl. a vocabulary that is not a multiple of 4 and a row whose maximum sits in the last, partial group."""
    from vagnmt_hip import ops
    torch.manual_seed(V)
    N, E, H = 5, 64, 32
    dev = "cuda"
    h2, c, e = torch.randn(N, H, device=dev), torch.randn(N, 2 * H, device=dev), torch.randn(N, E, device=dev)
    w1, w2, w3 = torch.randn(E, H, device=dev) * 0.2, torch.randn(E, 2 * H, device=dev) * 0.2, torch.randn(E, E, device=dev) * 0.2
    b1, b2, b3 = torch.randn(E, device=dev) * 0.1, torch.randn(E, device=dev) * 0.1, torch.randn(E, device=dev) * 0.1
    ow, ob = torch.randn(V, E, device=dev) * 0.3, torch.randn(V, device=dev) * 0.1
    ob[V - 1] += 50.0 if V % 4 else 0.0            # a clear maximum in the partial last group of four
    head = (w1, b1, w2, b2, w3, b3, ow, ob)
    logp, am = ops.head_logp_step(h2, c, e, head, want_argmax=True)
    t = torch.tanh(h2 @ w1.t() + b1 + c @ w2.t() + b2 + e @ w3.t() + b3)
    ref = torch.log_softmax((t.double() @ ow.double().t() + ob.double()), dim=-1)
    got = logp[:, :V].double()
    assert (got - ref).abs().max().item() <= 2e-4, (got - ref).abs().max().item()
    assert torch.equal(am.cpu(), ref.argmax(-1).cpu())
    assert abs(float(torch.logsumexp(got, -1).abs().max())) <= 1e-4          # rows are normalised


def test_recurrence_event_timing_reports_every_persistent_launch():
    """vag_set_option("persist_timing") + vag_recurrence_time (include/vag_nmt.h): an eager optimiser step at the benched
    shape records one launch of each of the four recurrence kernels with a plausible duration, a graph replay records
    nothing (events cannot be read out of a replay), and reading resets the counters."""
    import ctypes as C
    import bench
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    from machine_translation_vision.losses import PairwiseRankingLoss
    c = bench.CFG2
    dev = torch.device("cuda:0")
    m = bench.build_model(c, dev)
    vw = torch.ones(c["V"], device=dev)
    vw[0] = 0
    ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), use_graph=False)
    src, lens, tgt, im = bench.make_batch(c, 0, dev)
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)

    def read(kind):
        ms, n = C.c_double(0.0), C.c_int(0)
        L.call("vag_recurrence_time", kind, C.byref(ms), C.byref(n))
        return ms.value, n.value
    L.set_option("persist_timing", 1)
    try:
        for k in range(4):
            read(k)
        ts.step(src, lt, tgt, im, teacher=True)
        torch.cuda.synchronize()
        got = [read(k) for k in range(4)]
        for ms, n in got:
            assert n == 1 and 0.05 < ms < 5.0, got          # 0.13 .. 0.9 ms on an MI355X
        assert got[3][0] > got[0][0]                           # the decoder backward is the longest, the encoder forward the shortest
        assert all(read(k) == (0.0, 0) for k in range(4))      # reading resets
    finally:
        L.set_option("persist_timing", 0)
    ts.step(src, lt, tgt, im, teacher=True)
    torch.cuda.synchronize()
    assert all(read(k)[1] == 0 for k in range(4))              # switched off: nothing is recorded
    ts.check()


def test_bf16_dlogits_chunks_match_fp32_dlogits_in_the_two_byte_mode():
    """2-byte storage mode with the output head in row chunks (configs[4]'s regime) at H = 1024 / B = 256: d(logits) of a chunk
    written and read as bf16 (head_bf16_dlogits = 1: ce_bwd_colsum's out16, the bf16-stored A operand of the one-plane
    products) against the same step with fp32 d(logits) in place (whose products, at this reduced vocabulary, run on the exact
    f32 kernel): gradients within the mode's 1e-2 of each tensor's largest entry (measured 2e-3: one bf16 rounding of
    d(logits)); losses are untouched."""
    from test_gpu_round2 import _fp16_case
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    m_of, (src, lens, tgt, im), cm, cv = _fp16_case("wide")
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    B = src.shape[0]
    out = {}
    try:
        L.set_option("head_chunk", B)                       # one time step (256 rows) per chunk, five chunks
        for flag in (0, 1):
            L.set_option("head_bf16_dlogits", flag)
            m = m_of()
            ts = TrainStep(m, cm, cv, use_graph=False, storage="f16", pad_src=1)
            m.eval()
            ts.backend.run(src, lt, tgt, im, True, 7)
            torch.cuda.synchronize()
            out[flag] = ([float(x) for x in ts.backend.outputs()],
                         {n: p._vag_grad.detach().clone() for n, p in m.named_parameters()})
    finally:
        L.set_option("head_chunk", -1)
        L.set_option("head_bf16_dlogits", 1)
    (l0, g0), (l1, g1) = out[0], out[1]
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
    differs = False
    for n in g0:
        scale = max(g0[n].abs().max().item(), 1e-8)
        err = (g0[n] - g1[n]).abs().max().item()
        assert err <= 1e-2 * scale, (n, err, scale)
        differs = differs or err > 0.0
    assert differs                                           # the bf16 path really ran (sums in another order)
