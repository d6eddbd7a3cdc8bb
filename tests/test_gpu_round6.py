"""GPU: round-6 additions -- the launcher of bench.py with real ranks on the test box's card, per-tensor relative-L2 gradient
parity, the wider persistent recurrence kernels against the launch chains they replace and against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_self_launch_two_ranks_smoke():
    """`python bench.py --gpus 2` (no launcher around it) on a one-GPU box: VAG_DP_SMOKE=1 puts both ranks on cuda:0 over gloo with the
    launch-chain recurrences (two processes cannot both keep a persistent grid resident).  The line must say n_gpus 2 and carry the
    data-parallel block; SURVEY 8e, BASELINE.json configs[2]."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["VAG_DP_SMOKE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-operators", "--no-cpu-baseline", "--no-extras", "--single-window"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3
    assert d["dp"]["backend"] == "gloo" and len(d["dp"]["per_rank_ms_per_step"]) == 2
    assert np.isfinite(d["final_loss"]) and d["value"] > 0


# ---------------------------------------------------------------------------------------------- give-up accounting
R4_DIMS = (300, 333, 64, 32, 512, 48)       # Vs, Vt, I, E, H, S: H = 512 / B = 64 is a shape the persistent kernels take


def _r4_model(seed=0):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    Vs, Vt, I, E, H, S = R4_DIMS
    torch.manual_seed(seed)
    return NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda()


def _r4_batch(seed, B=64, Ts=12, Tt=5):
    Vs, Vt, I = R4_DIMS[:3]
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(4, Vs, (B, Ts), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    return src.cuda(), [Ts] * B, tgt.cuda(), torch.randn(B, I, generator=g).abs().cuda()


def _r4_driver(seed=0, **kw):
    from machine_translation_vision.losses import PairwiseRankingLoss
    from vagnmt_hip.trainer import TrainStep
    m = _r4_model(seed)
    vw = torch.ones(R4_DIMS[1], device="cuda")
    vw[0] = 0
    return m, TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), teacher_force_ratio=1.0, **kw)


def test_give_up_in_autograd_backward_thread_voids_the_step_of_its_driver():
    """Per-operator path (TrainStep(fused=False): model(...) + loss.backward() + the flat optimiser).  autograd runs the backward
    operators on its own worker thread; their persistent launches must still report to the DRIVER's guard pair, so the optimiser
    skips the step and check() raises -- not to the process-wide pair nobody reads (train.py:44-49 semantics for applied steps)."""
    from vagnmt_hip import _lib as L
    m, ts = _r4_driver(use_graph=False, fused=False)
    assert type(ts.backend).__name__ == "_AutogradBackend"
    for s in range(2):
        ts.step(*_r4_batch(10 + s), teacher=True)
    torch.cuda.synchronize()
    assert ts.skipped_steps() == 0
    L.lib().vag_persistent_timeouts()
    before = (ts.fp.flat.clone(), int(ts.step_count.item()))
    L.set_option("persist_spin_limit", 1)
    try:
        ts.step(*_r4_batch(21), teacher=True)
        torch.cuda.synchronize()
    finally:
        L.set_option("persist_spin_limit", 0)
    assert torch.equal(before[0], ts.fp.flat) and int(ts.step_count.item()) == before[1]
    assert ts.skipped_steps() == 1
    guard = ts._scratch.view(torch.int32)[ts.GUARD_OFFSET // 4 + 1]
    assert int(guard) > 0                                    # the give-ups of forward AND backward launches were counted here
    with pytest.raises(L.VagError):
        ts.check()
    ts.step(*_r4_batch(22), teacher=True)                    # healthy again
    torch.cuda.synchronize()
    assert int(ts.step_count.item()) == before[1] + 1 and not torch.equal(before[0], ts.fp.flat)
    ts.check()


def test_unguarded_step_consumes_the_process_wide_flag():
    """vag_train_step with cfg.guard == NULL (FusedStep used without a TrainStep) reports give-ups to the process-wide pair and turns
    them into a non-finite gradient entry.  That step must also CLEAR the pair: before, one transient give-up anywhere in the
    process poisoned every later unguarded step for good."""
    from vagnmt_hip import _lib as L
    m, ts = _r4_driver(use_graph=False)
    f = ts.backend.f
    f.guard = None                                            # the bare C-API form
    src, lens, tgt, im = _r4_batch(40)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m.train()
    g00 = m.encoder.embedding.weight._vag_grad[0, 0]
    L.set_option("persist_spin_limit", 1)
    try:
        ts.backend.run(src, lt, tgt, im, True, 7)
        torch.cuda.synchronize()
    finally:
        L.set_option("persist_spin_limit", 0)
    assert torch.isinf(g00)                                   # void step: injected
    ts.fp.grad.zero_()
    ts.backend.run(src, lt, tgt, im, True, 7)                 # a healthy step right after, nobody polled the count in between
    torch.cuda.synchronize()
    assert float(g00) == 0.0 and torch.isfinite(ts.fp.grad).all()
    assert L.lib().vag_persistent_timeouts() > 0              # the count is still there for whoever polls it
    assert L.lib().vag_persistent_timeouts() == 0


# ---------------------------------------------------------------- wider persistent recurrences
@pytest.mark.parametrize("H,B,Ts,Tt", [(256, 16, 40, 40), (256, 64, 12, 5), (256, 37, 9, 7), (256, 5, 7, 3), (256, 128, 8, 4),
                                        (256, 230, 6, 3), (256, 256, 5, 3), (512, 128, 9, 4), (512, 115, 7, 5), (512, 120, 5, 3)])
def test_wider_persistent_decoder_equals_launch_chain(H, B, Ts, Tt):
    """The one-launch decoder recurrences (forward and backward) and the one-launch encoder backward at H = 256 (BASELINE configs[0]:
    32 workgroups per row tile instead of 64) and for batches wider than one launch holds (B > 64 at H = 512, > 128 at H = 256:
    passes of row tiles through the same kernel) against the per-step launch chains: losses to 2e-6, every gradient to 2e-5 of its
    tensor's largest entry, no wait gave up (layers/NMT_Decoder.py:109-145 x models/...V11.py:138-146, layers/Encoder.py:58)."""
    from test_gpu_round3 import _model, _batch
    from vagnmt_hip import _lib as L
    from machine_translation_vision.losses import PairwiseRankingLoss
    assert L.lib().vag_recurrence_supported(1, B, Ts, Tt, H) == 1          # the shape takes the persistent decoder now
    src, lens, tgt, im = _batch(B, Ts, Tt, seed=3)
    vw = torch.ones(333, device="cuda")
    vw[0] = 0
    crit = torch.nn.NLLLoss(weight=vw, reduction="none")
    res = {}
    L.lib().vag_persistent_timeouts()
    for mode in (0, 1):
        L.set_option("persistent", mode)
        try:
            m = _model(H, seed=2)
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
    differs = False
    for n in g0:
        scale = max(g0[n].abs().max().item(), 1e-3)
        err = (g0[n] - g1[n]).abs().max().item()
        assert err <= 2e-5 * scale, (n, err, scale)
        differs = differs or err > 0.0
    assert differs                          # (the two paths sum in different orders: identical bits would mean the same kernels ran)


def test_persistent_decoder_eligibility_edges():
    from vagnmt_hip import _lib as L
    sup = L.lib().vag_recurrence_supported
    assert sup(1, 64, 40, 40, 512) == 1 and sup(1, 16, 40, 40, 256) == 1
    # two passes of row tiles at most (64 rows each at H = 512, 128 at H = 256), the second at least three quarters full
    assert sup(1, 128, 40, 40, 512) == 1 and sup(1, 129, 40, 40, 512) == 0 and sup(1, 96, 40, 40, 512) == 0 and sup(1, 97, 40, 40, 512) == 1
    assert sup(1, 256, 40, 40, 256) == 1 and sup(1, 257, 40, 40, 256) == 0 and sup(1, 200, 40, 40, 256) == 0
    assert sup(1, 64, 40, 40, 1024) == 0 and sup(1, 64, 40, 40, 128) == 0       # (configs[4]'s width: launch chains)
    assert sup(1, 64, 600, 40, 512) == 0                                        # keys of a row tile must fit the LDS


def test_public_beamsearch_with_the_reference_argument_layout():
    """models/...V11.py:233 / NMT_Seq2Seq_Beam_V2.py:173: `beamsearch(encoder_outputs (Ts,B,2H), context_mask (Ts,B), decoder_input (B,1),
    decoder_hidden (1,B,H), beam_size, max_length)` is a public method of the reference's models.  Called the way beamsearch_decode
    calls it there (:199-226), it must return the golden hypotheses of the reference run."""
    from conftest import load_golden
    from test_gpu_golden import build
    for name in ("mm_dot_tied_s0_f32", "text_tied_s0_f32"):
        meta, P, z = load_golden(name)
        m = build(meta, P)
        src = torch.from_numpy(z["src"]).cuda()
        lens = meta["lengths"]
        with torch.no_grad():
            if meta["kind"] == "mm":
                enc, mask, _, h0 = m._prologue(src, lens, torch.from_numpy(z["im"]).cuda(), None, None)
            else:
                enc, mask, h0 = m._prologue(src, lens, None)
        B = src.shape[0]
        assert m._validate_args(src, None, 17) == (B, 17)
        for k, want in meta["decode"].items():
            k = int(k)
            if k == 1:
                continue
            got = m.beamsearch(enc.transpose(0, 1), mask.transpose(0, 1), torch.full((B, 1), 2, dtype=torch.int64, device="cuda"),
                               h0.unsqueeze(0), k, meta["max_len"])
            assert [[int(t) for t in h] for h in got] == want, (name, k)
        with pytest.raises(NotImplementedError):
            m.beamsearch(enc.transpose(0, 1), mask.transpose(0, 1), None, h0.unsqueeze(0), 2, 5, avoid_unk=True)


def test_shim_adopts_restored_and_pre_stepped_optimizer_state():
    """ADVICE r5 (vag-nmt_amd/train.py): (a) optimizer.load_state_dict() on an optimiser that already has a fused driver replaces the
    installed views of the flat moment buffers -- the next fused step must continue from the RESTORED moments and step count, not from
    the stale flat buffers; (b) an optimiser that has stepped by itself before the first fused call is adopted (its moments copied into
    the flat buffers) instead of being demoted to the unfused path for the whole run.  Reference: two torch.optim.Adam runs on the
    per-operator path with the same sequence of calls (train.py:38-51)."""
    import copy
    from conftest import load_golden
    from test_gpu_golden import build, criteria
    from test_gpu_round5 import _shim, _reference_optimizer, _literal
    T = _shim()
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    cm, cv = criteria(meta)
    src, tgt, im = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda(), torch.from_numpy(z["im"]).cuda()
    lens = meta["lengths"]
    batch = (src, lens, tgt, im)
    # (a) three fused steps, snapshot, two more, restore the snapshot (weights + optimiser), two more: equals five literal steps
    # with the same restore in the middle
    ma, mb = build(meta, P), build(meta, P)
    oa, ob = _reference_optimizer(ma), _reference_optimizer(mb)

    def fused():
        return T.train_imagine_beam(src, tgt, im, lens, ma, oa, cm, cv, meta["loss_w"], 1.0, clip=1.0)
    for _ in range(3):
        fused(); _literal(mb, ob, cm, cv, batch, 1.0, True)
    snap_a = (copy.deepcopy(oa.state_dict()), {n: p.detach().clone() for n, p in ma.named_parameters()})
    snap_b = (copy.deepcopy(ob.state_dict()), {n: p.detach().clone() for n, p in mb.named_parameters()})
    for _ in range(2):
        fused(); _literal(mb, ob, cm, cv, batch, 1.0, True)
    for (sd, pw), m_, o_ in ((snap_a, ma, oa), (snap_b, mb, ob)):
        with torch.no_grad():
            for n, p in m_.named_parameters():
                p.copy_(pw[n])
        o_.load_state_dict(sd)
    d = oa._vag_driver
    assert not d.views_intact()                                # load_state_dict put fresh tensors into optimizer.state
    for i in range(2):
        got = fused(); want = _literal(mb, ob, cm, cv, batch, 1.0, True)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-5), (i, got, want)
    assert d.views_intact() and int(d.ts.step_count.item()) == 5 and type(d.ts.backend).__name__ == "_FusedBackend"
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert (pa - pb).abs().max().item() <= 3e-5 * max(1.0, pb.abs().max().item()), n
    # (b) two literal torch.optim.Adam steps first, then the shim: fused from its first call on, continuing that state
    mc, md = build(meta, P), build(meta, P)
    oc, od = _reference_optimizer(mc), _reference_optimizer(md)
    for _ in range(2):
        _literal(mc, oc, cm, cv, batch, 1.0, True); _literal(md, od, cm, cv, batch, 1.0, True)
    for i in range(3):
        got = T.train_imagine_beam(src, tgt, im, lens, mc, oc, cm, cv, meta["loss_w"], 1.0, clip=1.0)
        want = _literal(md, od, cm, cv, batch, 1.0, True)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-5), (i, got, want)
    dc = oc._vag_driver
    assert type(dc.ts.backend).__name__ == "_FusedBackend" and int(dc.ts.step_count.item()) == 5
    for (n, pa), (_, pb) in zip(mc.named_parameters(), md.named_parameters()):
        assert (pa - pb).abs().max().item() <= 3e-5 * max(1.0, pb.abs().max().item()), n


@pytest.mark.parametrize("H,B,Ts", [(512, 128, 6), (512, 115, 9), (1024, 64, 5), (256, 256, 4)])
def test_persistent_encoder_in_passes_equals_launch_chain(H, B, Ts):
    """The encoder's one-launch recurrences for a batch wider than one launch holds (2 x B/16 x H/16 workgroups > 256 CUs): two
    passes of row tiles through the same kernels, against the per-step launch chain (layers/Encoder.py:55-60)."""
    import test_gpu_round3 as R3
    from vagnmt_hip import _lib as L
    assert L.lib().vag_recurrence_supported(0, B, Ts, 1, H) == 1
    R3.test_persistent_encoder_equals_launch_chain(H, B, Ts)


def test_persistent_encoder_eligibility_edges():
    from vagnmt_hip import _lib as L
    sup = L.lib().vag_recurrence_supported
    assert sup(0, 64, 40, 1, 512) == 1 and sup(0, 128, 40, 1, 512) == 1 and sup(0, 129, 40, 1, 512) == 0 and sup(0, 96, 40, 1, 512) == 0
    assert sup(0, 256, 40, 1, 256) == 1 and sup(0, 32, 40, 1, 1024) == 1 and sup(0, 64, 40, 1, 1024) == 1 and sup(0, 65, 40, 1, 1024) == 0


@pytest.mark.timeout(1200)
def test_bench_multi_rank_falls_back_when_persistent_waits_give_up():
    """Two ranks on ONE card with the persistent kernels left on (VAG_DP_SMOKE=2) cannot both keep their 256-workgroup grids resident:
    bounded waits give up (forced here: spin limit 1 -- the one-GPU box time-slices the two processes and never starves a grid by itself), the optimiser skips those steps on every replica -- and a line
    timed over skipped steps would be worthless.  bench.py must notice (TrainStep.check on every rank, agreed by an all-reduce), take
    the persistent kernels back in stages, re-time and say so in dp.fallback: what keeps a first multi-GPU session from ending
    without a number should a collective's kernels starve the encoder backward's grid."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["VAG_DP_SMOKE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-operators", "--no-cpu-baseline", "--no-extras", "--single-window", "--opt", "persist_spin_limit=1"],
                       env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 2 and np.isfinite(d["final_loss"])
    assert d["dp"]["fallback"] and "launch chain" in d["dp"]["fallback"], d["dp"]
    assert d["dp"]["persistent_kernels"] is False


def test_split_k_through_slabs_equals_split_k_through_atomics():
    """Inside vag_train_step the k-slices of a 128 x 128 output tile park their accumulators in slabs of the step's workspace and the
    last arriver sums them in slice order (gemm.hip: GemmArgs::slab) instead of every slice adding into the output with one atomic
    per element: same sums to fp32 rounding, at configs[1] size where the planner really cuts products into 2-12 slices.  The slab
    region and the tickets are part of the workspace (vag_step_ws_floats grows with them)."""
    import bench
    from vagnmt_hip import _lib as L
    from test_gpu_benched_path import _driver
    c = bench.CFG2
    dev = torch.device("cuda", 0)
    batch = bench.make_batch(c, 0, dev)
    src, lens, tgt, im = batch
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    res = {}
    try:
        for flag in (0, 1):
            L.set_option("gemm_slabs", flag)
            m, ts = _driver(c, dropout=False, use_graph=False)
            m.train()
            ts.backend.run(src, lt, tgt, im, True, 7)
            torch.cuda.synchronize()
            res[flag] = ([float(x) for x in ts.backend.outputs()], {n: p._vag_grad.detach().clone() for n, p in m.named_parameters()})
            import ctypes as C
            f = ts.backend.f
            need = int(L.lib().vag_step_ws_floats(C.byref(f.cfg(c["B"], src.shape[1], c["Tt"], True, False))))
            tick = f.ws.view(torch.int32)[need - 16384:need]       # the tickets are the last region of the workspace
            assert int(tick.abs().max()) == 0                      # every tile's last block put its ticket back
            del m, ts
    finally:
        L.set_option("gemm_slabs", 1)
    (l0, g0), (l1, g1) = res[0], res[1]
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
    differs = False
    for n in g0:
        scale = max(g0[n].abs().max().item(), 1e-6)
        err = (g0[n] - g1[n]).abs().max().item()
        assert err <= 1e-5 * scale, (n, err, scale)
        differs = differs or err > 0.0
    assert differs                                                 # (another summation order: the slab path really ran)


def test_wide_fp16_encoder_applies_the_context_dropout_itself():
    """2-byte storage mode at configs[4] widths, TRAIN mode with dropout on: enc_fwd_wide16_kernel multiplies the encoder states by the
    counter-based context-dropout mask as it writes them (Encoder.py:63-64; round 6: a separate pass before) -- the same mask the
    launch chain's separate pass applies and the backward kernels recompute.  Encoder states against the launch chain (fp16 exchange:
    2e-3 absolute), exact zeros in the same places, losses and gradients within the mode's tolerances."""
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
            m.encoder.dropout_ctx = 0.5                            # the context dropout on (the fixture builds every dropout at 0)
            ts = TrainStep(m, cm, cv, use_graph=False, storage="f16", pad_src=1)
            m.train()
            ts.backend.run(src, lt, tgt, im, True, 7)
            torch.cuda.synchronize()
            f = ts.backend.f
            B, Ts = src.shape
            c = f.cfg(B, Ts, tgt.shape[1], True, True)
            assert abs(c.p_ctx - 0.5) < 1e-6                       # the step really runs with the context dropout
            off = L.lib().vag_step_ws_offset(C.byref(c), 0)
            enc = f.ws[off:off + B * Ts * 2 * 1024].detach().clone()
            out[persistent] = (enc, [float(x) for x in ts.backend.outputs()], ts.fp.grad.detach().clone())
            assert L.lib().vag_persistent_timeouts() == 0
    finally:
        L.set_option("persistent", 1)
    (e0, l0, g0), (e1, l1, g1) = out[0], out[1]
    assert torch.isfinite(e1).all()
    live = e0.view(src.shape[0], src.shape[1], -1)[0, 0] != 0       # a full-length row: about half of its 2H entries are dropped
    assert 0.35 < float(live.float().mean()) < 0.65
    assert torch.equal(e0 == 0, e1 == 0)                            # the same mask
    assert (e0 - e1).abs().max().item() <= 4e-3                     # (kept entries are scaled by 1 / (1 - p) = 2)
    assert np.allclose(l0, l1, rtol=1e-3, atol=1e-4), (l0, l1)
    assert (g0 - g1).abs().max().item() <= 1e-2 * g0.abs().max().item()


@pytest.mark.parametrize("H,B", [(512, 128), (256, 256)])
def test_two_pass_recurrences_match_the_oracle(H, B):
    """A batch two launches wide (B = 128 at H = 512, 256 at H = 256: two passes of row tiles through every persistent recurrence
    kernel) through the fused step against the CPU oracle itself, not only against the launch chains: losses 1e-4, every gradient
    3e-4 of its largest entry and 2e-4 relative L2 (models/...V11.py:82-168)."""
    from test_gpu_benched_path import _oracle, _check
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from machine_translation_vision.losses import PairwiseRankingLoss
    from vagnmt_hip.trainer import TrainStep
    from vagnmt_hip import _lib as L
    Vs, Vt, I, E, S, Ts, Tt = 300, 333, 64, 32, 48, 9, 6
    assert L.lib().vag_recurrence_supported(1, B, Ts, Tt, H) == 1 and L.lib().vag_recurrence_supported(0, B, Ts, 1, H) == 1
    torch.manual_seed(5)
    m = NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, tied_emb=True).cuda()
    g = torch.Generator().manual_seed(6)
    lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=g)], reverse=True)
    lens[0] = Ts
    src = torch.zeros(B, Ts, dtype=torch.long)
    for b, n in enumerate(lens):
        src[b, :n] = torch.randint(4, Vs, (n,), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    im = torch.randn(B, I, generator=g).abs()
    vw = torch.ones(Vt, device="cuda")
    vw[0] = 0
    ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), use_graph=False, pad_src=1)
    m.eval()
    batch = (src.cuda(), lens, tgt.cuda(), im.cuda())
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    L.lib().vag_persistent_timeouts()
    ts.backend.run(batch[0], lt, batch[2], batch[3], True, 7)
    torch.cuda.synchronize()
    assert L.lib().vag_persistent_timeouts() == 0
    losses = [float(x) for x in ts.backend.outputs()]
    grads = {n: p._vag_grad.detach().clone() for n, p in m.named_parameters()}
    want_l, want_g = _oracle(m, batch)
    _check("two passes H=%d B=%d" % (H, B), losses, grads, want_l, want_g)
