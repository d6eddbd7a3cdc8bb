"""GPU, round 5.

(1) The reference's trainer, unchanged, on the fused step: ``train_imagine_beam`` / ``train_nmt`` of the shadow module
    ``train`` (vag-nmt_amd/train.py; reference train.py:36-51, :19-32, called from nmt_multimodal_beam_DE.py:394) with the
    reference's own optimiser objects (``optim.Adam`` over the :303-332 groups, ``ReduceLROnPlateau`` :335):
    four steps at configs[1] size against the oracle, and at fixture size against the literal sequence on torch.optim.Adam.
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_golden import build, criteria, close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

LOSS_TOL, GRAD_TOL = 1e-4, 3e-4


def _shim():
    import train                      # vag-nmt_amd/train.py (conftest puts vag-nmt_amd on sys.path)
    assert os.path.dirname(os.path.abspath(train.__file__)).endswith("vag-nmt_amd"), train.__file__
    return train


def _reference_optimizer(model, lr=4e-4, wd=1e-5, vse_separate=False):
    """nmt_multimodal_beam_DE.py:303-332, as the script builds it."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if not vse_separate:
        groups = [{"params": [p for n, p in named if "bias" not in n], "weight_decay": wd},
                  {"params": [p for n, p in named if "bias" in n]}]
    else:
        groups = [{"params": [p for n, p in named if "bias" not in n and "vse_imagine" not in n], "weight_decay": wd},
                  {"params": [p for n, p in named if "bias" in n and "vse_imagine" not in n]},
                  {"params": [p for n, p in named if "bias" not in n and "vse_imagine" in n], "weight_decay": wd, "lr": lr / 2},
                  {"params": [p for n, p in named if "bias" in n and "vse_imagine" in n], "lr": lr / 2}]
    return torch.optim.Adam(groups, lr=lr)


def test_reference_trainer_steps_at_cfg2_match_oracle():
    """train_imagine_beam through the shim = the benched fused step: losses 1e-4, clip norm 3e-4, parameters after each of
    three Adam steps within the element-wise bound of test_gpu_benched_path (iii); dropout on, the kernels' masks handed to
    the oracle."""
    import bench
    from machine_translation_vision.losses import PairwiseRankingLoss
    from oracle import vag_oracle as O
    from test_gpu_benched_path import _masks
    T = _shim()
    c = bench.CFG2
    dev = torch.device("cuda", 0)
    m = bench.build_model(c, dev, dropout=True)
    vw = torch.ones(c["V"], device=dev)
    vw[0] = 0
    cm, cv = torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1)
    opt = _reference_optimizer(m)
    src, lens, tgt, im = bench.make_batch(c, 0, dev)
    P = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    state, acc = {}, {}
    for i in range(4):                       # (the shim captures a shape on its third visit: eager, eager, capture + replay, replay)
        got = T.train_imagine_beam(src, tgt, im, lens, m, opt, cm, cv, 0.99, 1.0, clip=1.0)
        assert len(got) == 3 and all(type(x) is float for x in got), got
        ts = opt._vag_driver.ts
        masks = _masks(m, c)
        o, grads, total, P, state = O.train_step(P, src.cpu(), lens, tgt.cpu(), im.cpu(), teacher=True, state=state,
                                                 masks=masks, hoist=True)
        for k, g in zip(("loss", "loss_mt", "loss_vse"), got):
            assert abs(g - float(o[k])) <= LOSS_TOL * max(1.0, abs(float(o[k]))), (i, k, g, float(o[k]))
        assert abs(float(ts.grad_norm[0]) - float(total)) <= 3e-4 * float(total), (i, float(ts.grad_norm[0]), float(total))
        bc2 = 1.0 - 0.999 ** (i + 1)
        for n, p in m.named_parameters():
            gmax = float(grads[n].abs().max()) * min(1.0, 1.0 / (float(total) + 1e-6))
            vhat = (state[n][1] / bc2).sqrt()
            acc[n] = acc.get(n, 0.0) + 4e-4 * torch.clamp(2 * GRAD_TOL * gmax / (vhat + 1e-8), max=1.0)
            err = (p.detach().cpu() - P[n]).abs()
            assert not bool((err > 2e-5 + acc[n]).any()), (i, n, float(err.max()))
            assert float(err.mean()) <= 2e-5, (i, n, float(err.mean()))
    ts = opt._vag_driver.ts
    assert ts.stats["captures"] == 1 and ts.stats["replays"] >= 2 and int(ts.step_count.item()) == 4, ts.stats
    assert type(ts.backend).__name__ == "_FusedBackend"
    sd = opt.state_dict()                                    # Adam's state under torch's names: step count and moments
    assert all(float(st["step"]) == 4.0 for st in sd["state"].values())
    p0 = opt.param_groups[0]["params"][0]
    assert opt.state[p0]["exp_avg"].data_ptr() >= ts.fp.m.data_ptr()
    ts.check()


def _literal(m, opt, cm, cv, batch, clip, mm):
    """train.py:38-51 / :21-32 on the per-operator path with the caller's torch.optim.Adam."""
    src, lens, tgt, im = batch
    m.train()
    opt.zero_grad()
    if mm:
        loss, loss_mt, loss_vse = m(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)
    else:
        loss = m(src, lens, tgt, 1.0, criterion=cm)
        loss_mt = loss_vse = loss
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
    opt.step()
    return loss.item(), loss_mt.item(), loss_vse.item()


@pytest.mark.parametrize("name,vse_separate", [("mm_dot_tied_mid_f32", False), ("mm_dot_tied_mid_f32", True),
                                               ("mm_mlp_untied_s1_f32", True), ("text_tied_s0_f32", False)])
def test_shim_equals_the_literal_sequence_with_torch_adam(name, vse_separate):
    """Six steps, a ReduceLROnPlateau-style cut of every group's rate after step 3 (nmt_multimodal_beam_DE.py:335,469 multiplies
    each group's lr by 0.2), the lr/2 groups of :316-329, a change of ``clip`` on the way: the shim (fused step, graphs) against the same
    calls on the per-operator path + torch.optim.Adam."""
    T = _shim()
    meta, P, z = load_golden(name)
    mm = meta["kind"] == "mm"
    cm, cv = criteria(meta)
    src, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda()
    im = torch.from_numpy(z["im"]).cuda() if mm else None
    batch = (src, meta["lengths"], tgt, im)
    ma, mb = build(meta, P), build(meta, P)
    oa, ob = _reference_optimizer(ma, vse_separate=vse_separate), _reference_optimizer(mb, vse_separate=vse_separate)
    sched_a = torch.optim.lr_scheduler.ReduceLROnPlateau(oa, factor=0.2, patience=0)
    sched_b = torch.optim.lr_scheduler.ReduceLROnPlateau(ob, factor=0.2, patience=0)
    if not mm:
        T.CLIP = 1.0
    for i in range(6):
        clip = 1.0 if i < 5 else 0.5
        if mm:
            got = T.train_imagine_beam(src, tgt, im, meta["lengths"], ma, oa, cm, cv, meta["loss_w"], 1.0, clip=clip)
        else:
            got = (T.train_nmt(src, tgt, meta["lengths"], ma, cm, oa, 1.0),) * 3
            clip = 1.0
        want = _literal(mb, ob, cm, cv if mm else None, batch, clip, mm)
        assert np.allclose(got, want, rtol=2e-4, atol=2e-5), (i, got, want)
        if i == 2:                      # two "bad" epochs in a row: the scheduler cuts every group's rate
            for s in (sched_a, sched_b):
                s.step(1.0)
                s.step(2.0)
            assert abs(oa.param_groups[0]["lr"] - 8e-5) < 1e-12
    d = oa._vag_driver
    assert type(d.ts.backend).__name__ == "_FusedBackend" and int(d.ts.step_count.item()) == 6
    assert abs(d.ts.lr - 8e-5) < 1e-12
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        close(pa, pb.detach().cpu().numpy(), 3e-5, "parameters after six steps: " + n)
    # torch's view of the optimiser state is the driver's
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert len(sa) == len(sb)
    for k in sb:
        assert float(sa[k]["step"]) == float(sb[k]["step"]) == 6.0
        close(sa[k]["exp_avg"], sb[k]["exp_avg"].cpu().numpy(), 2e-4, "exp_avg %s" % k)


def test_shim_serves_other_criteria_on_the_per_operator_path_and_keeps_one_state():
    """A criterion the fused step does not implement (label smoothing by hand) goes through the literal sequence; on an optimiser
    that already has a fused driver torch.optim.Adam then steps on the driver's own moment buffers and step count."""
    T = _shim()
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    cm, cv = criteria(meta)
    src, tgt, im = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda(), torch.from_numpy(z["im"]).cuda()
    lens = meta["lengths"]

    class Other(torch.nn.Module):
        def forward(self, logp, target):
            return torch.nn.functional.nll_loss(logp, target, reduction="none") * 0.9 - 0.1 * logp.mean(dim=1)

    ma, mb = build(meta, P), build(meta, P)
    oa, ob = _reference_optimizer(ma), _reference_optimizer(mb)
    for i in range(4):
        crit = cm if i != 2 else Other()
        got = T.train_imagine_beam(src, tgt, im, lens, ma, oa, crit, cv, meta["loss_w"], 1.0, clip=1.0)
        mb.train()
        ob.zero_grad()
        loss, loss_mt, loss_vse = mb(src, lens, tgt, im, 1.0, criterion_mt=crit, criterion_vse=cv)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(mb.parameters(), 1.0)
        ob.step()
        assert np.allclose(got, (loss.item(), loss_mt.item(), loss_vse.item()), rtol=2e-4, atol=2e-5), (i, got)
    assert int(oa._vag_driver.ts.step_count.item()) == 4
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        close(pa, pb.detach().cpu().numpy(), 3e-5, "parameters: " + n)
    # decoding after shim steps sees the stepped weights (tables keyed on the weights version)
    ma.eval()
    mb.eval()
    ta = ma.beamsearch_decode(src, lens, im, 3, 10)
    tb = mb.beamsearch_decode(src, lens, im, 3, 10)
    assert [list(map(int, x)) for x in ta] == [list(map(int, x)) for x in tb]


# ---------------------------------------------------------------------------------------------------------------------------
# (2) BASELINE.json configs[0] at its own size: the text-only model (models/NMT_Seq2Seq_Beam_V2.py:58-113), H = 256, E = 256,
#     B = 16, T = 40.  At H = 256 the encoder takes the persistent kernels and the decoder the launch chain: a kernel selection
#     no fixture-size test runs.
# ---------------------------------------------------------------------------------------------------------------------------
def _text_oracle(m, batch, teacher):
    from oracle import vag_oracle as O
    src, lens, tgt = batch
    leaves = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in m.named_parameters()}
    out = O.model_forward(leaves, src.cpu(), lens, tgt.cpu(), None, teacher=teacher, hoist=True)
    out["loss"].backward()
    return float(out["loss"]), {n: (v.grad if v.grad is not None else torch.zeros_like(v)) for n, v in leaves.items()}


@pytest.mark.parametrize("teacher", [True, False])
@pytest.mark.parametrize("ragged", [False, True])
def test_cfg1_text_only_at_config_size_matches_oracle(teacher, ragged):
    """Loss 1e-4, every gradient 3e-4 of its tensor's largest entry, through BOTH host paths: the module API
    (model(...), loss.backward(): what train.py:25-28 runs) and the fused step driver (eager and graph replay)."""
    import bench
    from vagnmt_hip.trainer import TrainStep
    c = bench.CFG1
    dev = torch.device("cuda", 0)
    m = bench.build_text_model(c, dev, dropout=False)
    src, lens, tgt = bench.make_text_batch(c, 0, dev)
    if ragged:
        g = torch.Generator().manual_seed(5)
        lens = sorted(torch.clamp((torch.randn(c["B"], generator=g) * 5 + 15).round().long(), 4, c["Ts"]).tolist(), reverse=True)
        lens[0] = c["Ts"]
        for b, L in enumerate(lens):
            src[b, L:] = 0
        tgt[3, 20:] = 0
        tgt[3, 19] = 3
    vw = torch.ones(c["V"], device=dev)
    vw[0] = 0
    crit = torch.nn.NLLLoss(weight=vw, reduction="none")
    want_l, want_g = _text_oracle(m, (src, lens, tgt), teacher)
    # module API
    m.train()
    m.zero_grad(set_to_none=True)
    loss = m(src, lens, tgt, 1.0 if teacher else 0.0, criterion=crit)
    loss.backward()
    assert abs(loss.item() - want_l) <= LOSS_TOL * max(1.0, abs(want_l)), ("module api", loss.item(), want_l)
    for n, p in m.named_parameters():
        g_ = p.grad if p.grad is not None else torch.zeros_like(p)
        err = (g_.cpu() - want_g[n]).abs().max().item()
        assert err <= GRAD_TOL * max(want_g[n].abs().max().item(), 1e-3), ("module api", n, err)
    # fused step driver: eager, capture + replay, replay
    ts = TrainStep(m, crit, None, teacher_force_ratio=1.0)
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    for visit in range(3):
        ts.fp.grad.zero_()
        ts.backend.run(src, lt, tgt, None, teacher, 7)
        got = float(ts.backend.outputs()[0])
        assert abs(got - want_l) <= LOSS_TOL * max(1.0, abs(want_l)), ("fused", visit, got, want_l)
        for n, p in m.named_parameters():
            err = (p._vag_grad.cpu() - want_g[n]).abs().max().item()
            assert err <= GRAD_TOL * max(want_g[n].abs().max().item(), 1e-3), ("fused", visit, n, err)
    assert ts.stats["captures"] == 1
    ts.check()


def test_cfg1_greedy_and_beam_decode_at_config_size_match_oracle():
    import bench
    from oracle import vag_oracle as O
    c = bench.CFG1
    dev = torch.device("cuda", 0)
    m = bench.build_text_model(c, dev, dropout=False).eval()
    src, lens, _ = bench.make_text_batch(c, 0, dev)
    P = {n: p.detach().cpu() for n, p in m.named_parameters()}
    got = m.beamsearch_decode(src, lens, 1, 30)
    want = O.greedy_decode(P, src.cpu(), lens, None, max_length=30)
    assert [list(map(int, x)) for x in got] == [list(map(int, x)) for x in want]
    got = m.beamsearch_decode(src, lens, 4, 20)
    want = O.beam_search(P, src.cpu(), lens, None, beam_size=4, max_length=20)
    assert [list(map(int, x)) for x in got] == [list(map(int, x)) for x in want]


# ---------------------------------------------------------------------------------------------------------------------------
# (3) Run-to-run spread of the paths that use fp32 atomics (the persistent decoder's score / d-alpha shares, split-K products,
#     embedding scatters): reproducible to rounding, and bounded here.
# ---------------------------------------------------------------------------------------------------------------------------
def test_run_to_run_spread_of_one_cfg2_step_is_bounded():
    """Five replays of ONE configs[1] forward + backward (dropout off, same weights, same batch): loss spread <= 1e-6
    relative, every gradient tensor's spread <= 3e-5 of its largest entry (observed on MI355X: 0 on the losses, 7.6e-6 .. 1.1e-5 on
    the worst tensor, decoder.attn.attn_h.weight, which sums the d-alpha shares' rounding over B x Tt x Ts terms)."""
    import bench
    from test_gpu_benched_path import _driver, _run_phases
    c = bench.CFG2
    m, ts = _driver(c, dropout=False)
    batch = bench.make_batch(c, 0, torch.device("cuda", 0))
    m.train()
    runs = _run_phases(ts, batch, 6)[1:]                     # replays (and the capturing visit)
    losses = np.array([r[0] for r in runs])
    assert (losses.max(0) - losses.min(0) <= 1e-6 * np.maximum(1.0, np.abs(losses).max(0))).all(), losses
    worst = (0.0, None)
    for n in runs[0][1]:
        st = torch.stack([r[1][n] for r in runs])
        spread = float((st.max(0).values - st.min(0).values).max())
        rel = spread / max(float(st.abs().max()), 1e-12)
        if rel > worst[0]:
            worst = (rel, n)
        assert rel <= 3e-5, (n, spread, float(st.abs().max()))
    print("run-to-run spread: worst gradient tensor %s at %.2e of its largest entry; losses %s"
          % (worst[1], worst[0], (losses.max(0) - losses.min(0)).tolist()))
    ts.check()


# ---------------------------------------------------------------------------------------------------------------------------
# (4) Persistent-kernel give-up state per driver (vag_step_cfg.guard, VAG_ADAM_SCRATCH_GUARD_OFFSET): two drivers on one device.
# ---------------------------------------------------------------------------------------------------------------------------
def test_two_drivers_on_one_device_do_not_void_each_others_steps():
    """Driver A's recurrences are forced to give up (spin limit 1) between two of driver B's steps, and a greedy decode of a third
    model gives up as well: A skips its step (weights untouched, check() raises), B's steps are all applied and B's check() is
    clean -- before round 5 the give-up word was process-wide and B's next step would have been skipped too."""
    from test_gpu_round4 import _driver, _batch, _state
    from vagnmt_hip import _lib as L
    ma, a = _driver(seed=3, use_graph=False)
    mb, b = _driver(seed=4, use_graph=False)
    for s in range(2):
        a.step(*_batch(50 + s), teacher=True)
        b.step(*_batch(60 + s), teacher=True)
    a.check()
    b.check()
    a0, b0 = _state(a), _state(b)
    L.set_option("persist_spin_limit", 1)
    try:
        a.step(*_batch(52), teacher=True)                   # void: every wait gives up
        src, lens, _, im = _batch(70)
        mb.eval()
        mb.beamsearch_decode(src, lens, im, 1, 6)           # an unguarded launch gives up too (its tokens are garbage)
        torch.cuda.synchronize()
    finally:
        L.set_option("persist_spin_limit", 0)
    out = b.step(*_batch(62), teacher=True)                 # B's next step: healthy, must be APPLIED
    torch.cuda.synchronize()
    a1, b1 = _state(a), _state(b)
    assert torch.equal(a0[0], a1[0]) and a0[3] == a1[3] and a.skipped_steps() == 1
    assert b1[3] == b0[3] + 1 and not torch.equal(b0[0], b1[0]) and b.skipped_steps() == 0
    assert torch.isfinite(out[0]) and torch.isfinite(b.grad_norm).all()
    b.check()                                               # nothing of B gave up
    assert b.process_timeouts > 0                           # ... although the process-wide diagnostic count saw A's and the decoder's
    with pytest.raises(L.VagError):
        a.check()
    a.step(*_batch(53), teacher=True)                       # A recovers
    torch.cuda.synchronize()
    assert int(a.step_count.item()) == a0[3] + 1
    a.check()


# ---------------------------------------------------------------------------------------------------------------------------
# (5) Marked-word hand-offs (persist.hip: tag1 / ld_rows_tagged) under skewed starts: the producer signals WITHOUT draining its
#     stores, a consumer re-loads words whose mark is missing.  A kernel squatting on a quarter of the CUs when the persistent decoder
#     forward is launched makes some of its workgroups start up to ~0.2 ms late (profiles/r04_exp_squatter.txt): early workgroups
#     then poll and load while late ones have not stored anything -- every run must still equal the undisturbed one.
# ---------------------------------------------------------------------------------------------------------------------------
def test_marked_handoffs_survive_skewed_workgroup_starts():
    import ctypes as C
    import subprocess
    import bench
    from vagnmt_hip import _lib as L, ops
    so = os.path.join(ROOT, "tools", "libsquatter.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "squatter.hip"), "-o", so])
    SQ = C.CDLL(so)
    SQ.squat.restype = C.c_int
    SQ.squat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    c = bench.CFG2
    dev = torch.device("cuda", 0)
    m = bench.build_model(c, dev, dropout=False).eval()
    src, lens, tgt, im = bench.make_batch(c, 0, dev, ragged=True)
    lt = torch.tensor(lens, dtype=torch.int32, device=dev)
    assert L.lib().vag_recurrence_supported(1, c["B"], c["Ts"], c["Tt"], c["H"]) == 1
    with torch.no_grad():
        enc, mask = m._encode(src, lt, None)
        _, ctx = m.vse_imagine.forward_bm(im, enc, mask, None)
        h0 = ops.DecInit.apply(enc, mask, ctx, m.decoderini.weight, m.decoderini.bias, 0.5)
        pe = ops.KeysProj.apply(enc, m.decoder.attn.attn_e.weight)
        sos = torch.full((1, c["B"]), 2, dtype=torch.int64, device=dev)
        tok = torch.cat([sos, tgt.t()], 0).contiguous()
        dec = m.decoder

        def run():
            h2, cc, e = ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), V=c["V"])
            return h2.clone(), cc.clone()
        ref_h2, ref_c = run()
        torch.cuda.synchronize()
        assert L.lib().vag_persistent_timeouts() == 0
        side = torch.cuda.Stream()
        sink = torch.zeros(4, device=dev)
        worst = 0.0
        for i in range(12):
            wgs, us = (64, 150.0) if i % 2 == 0 else (160, 60.0)
            side.wait_stream(torch.cuda.current_stream())
            assert SQ.squat(side.cuda_stream, wgs, 256, 16 * 1024, us, sink.data_ptr()) == 0       # resident first
            h2, cc = run()
            torch.cuda.synchronize()
            # (the score shares are summed with fp32 atomics: equal to rounding, not bitwise)
            worst = max(worst, float((h2 - ref_h2).abs().max()), float((cc - ref_c).abs().max()))
            assert torch.isfinite(h2).all()
        assert worst <= 2e-5, worst
        assert L.lib().vag_persistent_timeouts() == 0


# ---------------------------------------------------------------------------------------------------------------------------
# (6) The reference's checkpointing after shim steps and an evaluation pass: torch.save(model) / torch.load
#     (nmt_multimodal_beam_DE.py:491-520, :534).
# ---------------------------------------------------------------------------------------------------------------------------
def test_whole_module_checkpoint_after_shim_steps_and_decoding():
    import io
    T = _shim()
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    cm, cv = criteria(meta)
    src, tgt, im = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda(), torch.from_numpy(z["im"]).cuda()
    lens = meta["lengths"]
    m = build(meta, P)
    opt = _reference_optimizer(m)
    for _ in range(3):
        T.train_imagine_beam(src, tgt, im, lens, m, opt, cm, cv, meta["loss_w"], 1.0, clip=1.0)
    m.eval()
    dev = m(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)          # validation loss, as :431
    want_beam = m.beamsearch_decode(src, lens, im, 3, 12)                       # (captures decode graphs on the module)
    want_greedy = m.beamsearch_decode(src, lens, im, 1, 12)
    buf = io.BytesIO()
    torch.save(m, buf)                                                          # :492
    assert buf.tell() < 3 * 4 * sum(p.numel() for p in m.parameters())          # values once: no gradient buffer, no caches
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)                                    # :534
    m2.eval()
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2) and not hasattr(p2, "_vag_grad")
    got = m2(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)
    assert abs(float(got[0]) - float(dev[0])) <= 1e-6 * max(1.0, abs(float(dev[0])))
    assert [list(map(int, x)) for x in m2.beamsearch_decode(src, lens, im, 3, 12)] == [list(map(int, x)) for x in want_beam]
    assert [list(map(int, x)) for x in m2.beamsearch_decode(src, lens, im, 1, 12)] == [list(map(int, x)) for x in want_greedy]
    # the loaded module trains through the module API with ordinary .grad tensors
    m2.train()
    loss, _, _ = m2(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m2.parameters())
    # ... and the live model goes on training on the fused step
    T.train_imagine_beam(src, tgt, im, lens, m, opt, cm, cv, meta["loss_w"], 1.0, clip=1.0)
    assert int(opt._vag_driver.ts.step_count.item()) == 4


# ---------------------------------------------------------------------------------------------------------------------------
# (7) The launches merged late in round 5 (one-launch dot attention with the mean-pool rider, loss reduction riding in the head's
#     backward) against the separate launches they replace, over the register kernel's position counts (Ts <= 16, 32, 48, 64) and
#     past them (the generic row kernel).
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Ts,H", [(9, 512), (16, 512), (17, 512), (33, 512), (48, 512), (52, 512), (64, 512), (70, 512),
                                  (12, 256), (40, 256), (64, 256)])
def test_merged_visual_attention_launches_equal_the_separate_ones(Ts, H):
    import bench
    from vagnmt_hip import _lib as L
    from test_gpu_benched_path import _driver, _run_phases
    c = dict(bench.CFG2, Ts=Ts, Tt=7, B=24, H=H)       # H = 256: context width 512, the register kernel's other instantiation
    dev = torch.device("cuda", 0)
    m, ts = _driver(c, dropout=False, use_graph=False)
    batch = bench.make_batch(c, 3, dev, ragged=True)
    m.train()
    got = {}
    try:
        for mode in (1, 0):
            L.set_option("attn_row", mode)
            L.set_option("loss_ride", mode)
            got[mode] = _run_phases(ts, batch, 1)[0]
    finally:
        L.set_option("attn_row", 1)
        L.set_option("loss_ride", 1)
    (l1, g1), (l0, g0) = got[1], got[0]
    assert np.allclose(l1, l0, rtol=2e-6, atol=1e-6), (l1, l0)
    for n in g0:
        ref = float(g0[n].abs().max())
        err = float((g1[n] - g0[n]).abs().max())
        assert err <= 2e-5 * max(ref, 1e-3), (n, err, ref)
    ts.check()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("Ts", [11, 32])
def test_fused_decoding_step_attention_matches_the_two_launches_and_the_oracle(Ts):
    """attn_row_gru_kernel (scores + softmax + projected context + gru_2 + W2 c of a beam step in one launch; H = 512, Ts <= 32)
    against the separate launches (option attn_row = 0: same winners, scores to rounding) and against the oracle's beam search."""
    from oracle import vag_oracle as O
    from vagnmt_hip import _lib as L
    from test_gpu_round2 import make
    lens = sorted([max(3, Ts - 2 * i) for i in range(8)], reverse=True)
    lens[0] = Ts
    m, src, tgt, im = make(8507, 9391, 2048, 256, 512, 512, 8, Ts, 6, lens, seed=33)
    with torch.no_grad():
        m.decoder.out.bias[3] += 2.0
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, want_scores = O.beam_search(P, src, lens, im, beam_size=12, max_length=30, return_scores=True)
    mg = m.cuda().eval()
    res = {}
    try:
        for mode in (1, 0):
            L.set_option("attn_row", mode)
            mg.__dict__.pop("_decode_cache", None)
            got = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 12, 30)]
            res[mode] = (got, mg.last_beam_scores.cpu().numpy().copy())
    finally:
        L.set_option("attn_row", 1)
        mg.__dict__.pop("_decode_cache", None)
    assert np.allclose(res[1][1], res[0][1], rtol=1e-5, atol=1e-5)
    assert sum(a == b for a, b in zip(res[1][0], res[0][0])) >= 7
    assert np.allclose(res[1][1], want_scores.numpy(), rtol=2e-4, atol=2e-4), np.abs(res[1][1] - want_scores.numpy()).max()
    assert sum(a == b for a, b in zip(res[1][0], want)) >= 7
