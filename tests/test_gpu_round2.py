"""GPU: parity cases added in round 2 -- Bahdanau attention weights against the golden per-step alpha (module API and the
fused sequence), the fused training step against the per-operator autograd path, the bf16x6 product against the exact-f32
kernel, the four-group optimiser against torch.optim.Adam, the bounded graph cache, checkpoint resume equivalence and the
beam search at the size BASELINE configs[3] names."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_golden import F32_CASES, build, close, criteria
from test_gpu_edge_and_full import make

pytestmark = pytest.mark.gpu


def _inputs(meta, z):
    src, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda()
    im = torch.from_numpy(z["im"]).cuda() if meta["kind"] == "mm" else None
    return src, meta["lengths"], tgt, im


# ------------------------------------------------------------------------------------------------------------------
# a4: Bahdanau attention weights (layers/NMT_Decoder.py:27-51) against the reference's own per-step alpha
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", F32_CASES)
def test_bahdanau_alpha_module_api(name):
    """m.decoder.attn(h1, enc, mask) -> vag_bahdanau_attn_fwd, fed the reference's h1 of every teacher-forced step
    (recomputed by the oracle, which is pinned to the same fixtures)."""
    from oracle import vag_oracle as O
    meta, P, z = load_golden(name)
    m = build(meta, P)
    src, lens, tgt, im = _inputs(meta, z)
    enc_ref, mask_ref = torch.from_numpy(z["enc"]), torch.from_numpy(z["mask"])
    Pf = {k: v.float() for k, v in P.items()}
    # decoder initial state and the h1 sequence on the CPU
    if meta["kind"] == "mm":
        _, _, _, ctx = O.vse_forward(Pf, im.cpu(), enc_ref, mask_ref, method=meta["attn"])
        h = O.decoder_init(Pf, enc_ref, mask_ref, ctx, meta["init_split"])
    else:
        h = O.decoder_init(Pf, enc_ref, mask_ref, None, 0.0)
    tok = torch.full((src.shape[0],), 2, dtype=torch.long)
    with torch.no_grad():
        enc, mask = m.encoder(src, lens)
        for di in range(tgt.shape[1]):
            _, h, aux = O.decoder_step(Pf, tok, h, enc_ref, mask_ref, tied=meta["tied"])
            close(aux["alpha"], z["alpha_steps"][di], 2e-6, "oracle alpha %d" % di)
            a = m.decoder.attn(aux["h1"].cuda().unsqueeze(0), enc, ctx_mask=mask)
            assert a.shape == (src.shape[0], 1, src.shape[1])
            close(a[:, 0, :], z["alpha_steps"][di], 1e-4, "alpha step %d" % di)
            tok = tgt[:, di].cpu()


@pytest.mark.parametrize("name", F32_CASES)
def test_bahdanau_alpha_saved_by_fused_sequence(name):
    """The alpha the training path itself computes and keeps for backward (vag_cgru_attn_decode_seq_fwd workspace)."""
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    meta, P, z = load_golden(name)
    m = build(meta, P)
    cm, cv = criteria(meta)
    ts = TrainStep(m, cm, cv if meta["kind"] == "mm" else None, use_graph=False, pad_src=1)
    src, lens, tgt, im = _inputs(meta, z)
    m.eval()        # no dropout: the fixtures are eval-mode
    be = ts.backend
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    # drive the backend directly so that model.train() (TrainStep.step) does not switch dropout on
    be.run(src, lt, tgt, im, True, 1)
    B, Ts = src.shape
    Tt = tgt.shape[1]
    f = be.f
    c = f.cfg(B, Ts, Tt, True, False)
    lib = L.lib()
    off = lib.vag_step_ws_offset(C.byref(c), 4) + lib.vag_cgru_ws_offset(B, Ts, Tt, f.Et, f.H, 0)
    alpha = f.ws[off:off + Tt * B * Ts].view(Tt, B, Ts)
    close(alpha, z["alpha_steps"], 1e-4, "saved alpha")
    close(f.losses[0], z["teacher/loss"], what="loss")
    if meta["kind"] == "mm":
        av = f.ws[lib.vag_step_ws_offset(C.byref(c), 1):][:B * Ts].view(B, Ts)
        close(av, z["alpha_vse"], 1e-4, "alpha_vse")


# ------------------------------------------------------------------------------------------------------------------
# fused step == per-operator autograd path
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mm_dot_tied_s0_f32", "mm_mlp_untied_s1_f32", "text_tied_s0_f32", "mm_dot_tied_mid_f32"])
@pytest.mark.parametrize("teacher", [True, False])
def test_fused_step_gradients_equal_autograd_path(name, teacher):
    from vagnmt_hip.trainer import TrainStep
    meta, P, z = load_golden(name)
    cm, cv = criteria(meta)
    cv = cv if meta["kind"] == "mm" else None
    src, lens, tgt, im = _inputs(meta, z)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    grads, losses = [], []
    for fused in (True, False):
        m = build(meta, P)
        ts = TrainStep(m, cm, cv, use_graph=False, fused=fused, pad_src=1)
        m.eval()
        ts.backend.run(src, lt, tgt, im, teacher, 7)
        losses.append([float(x) for x in ts.backend.outputs()])
        grads.append({n: p._vag_grad.detach().clone() for n, p in m.named_parameters()})
    assert np.allclose(losses[0], losses[1], rtol=1e-5, atol=1e-6), losses
    if teacher:
        close(torch.tensor(losses[0][0]), z["teacher/loss"], what="fused loss")
    else:
        close(torch.tensor(losses[0][0]), z["free/loss"], what="fused free-running loss")
    for n in grads[0]:
        ref = grads[1][n]
        err = (grads[0][n] - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1e-3), (n, err)
        if teacher:
            close(grads[0][n], z["G/" + n], 2e-4, "fused grad " + n)


def test_fused_step_source_padding_is_exact():
    """pad_src rounds the source length up with masked positions: same loss and gradients as the unpadded batch."""
    from vagnmt_hip.trainer import TrainStep
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    cm, cv = criteria(meta)
    src, lens, tgt, im = _inputs(meta, z)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    res = []
    for pad in (1, 8):
        m = build(meta, P)
        ts = TrainStep(m, cm, cv, use_graph=False, pad_src=pad)
        m.eval()
        ts.backend.run(src, lt, tgt, im, True, 7)
        res.append(([float(x) for x in ts.backend.outputs()], ts.fp.grad.clone()))
    assert np.allclose(res[0][0], res[1][0], rtol=2e-6, atol=1e-6)
    assert (res[0][1] - res[1][1]).abs().max().item() <= 2e-5 * res[0][1].abs().max().item()


# ------------------------------------------------------------------------------------------------------------------
# bf16x6 products against the exact-f32 kernel (include/vag_nmt.h: vag_gemm_f32)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("shape", [(2560, 1024, 1024, True, True), (1536, 512, 2560, False, False), (640, 9391, 256, True, True),
                                   (2560, 256, 9391, True, False), (1000, 333, 777, False, True), (200, 130, 100, True, True)])
def test_bf16x6_error_is_bounded_by_the_exact_f32_kernel(shape, wide):
    from vagnmt_hip import _lib as L
    M, N, K, a_kc, b_kc = shape
    rs = np.random.RandomState(7)
    A = rs.randn(M, K).astype(np.float32)
    Bm = rs.randn(K, N).astype(np.float32)
    if wide:        # 2^12 dynamic range inside every operand
        A *= np.exp2(rs.uniform(-6, 6, size=A.shape)).astype(np.float32)
        Bm *= np.exp2(rs.uniform(-6, 6, size=Bm.shape)).astype(np.float32)
    At = torch.from_numpy(A if a_kc else np.ascontiguousarray(A.T)).cuda()
    Bt = torch.from_numpy(np.ascontiguousarray(Bm.T) if b_kc else Bm).cuda()
    sam, sak = (K, 1) if a_kc else (1, M)
    sbk, sbn = (1, K) if b_kc else (N, 1)
    ref = A.astype(np.float64) @ Bm.astype(np.float64)
    # per-element error scale: sum_k |a||b| (what any fp32 summation order is judged against)
    mag = np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    errs = {}
    for mode in ("bf16x6", "f32"):
        L.set_option("gemm_f32mfma", 1 if mode == "f32" else 0)
        try:
            Ct = torch.zeros(M, N, device="cuda")
            L.call("vag_gemm_f32", M, N, K, 1.0, L.ptr(At), sam, sak, L.ptr(Bt), sbk, sbn, 0.0, L.ptr(Ct), N, None, 0,
                   L.stream())
            got = Ct.cpu().numpy().astype(np.float64)
        finally:
            L.set_option("gemm_f32mfma", 0)
        errs[mode] = float((np.abs(got - ref) / mag).max())
    # fp32-grade: a few ulp of the magnitude sum, and no worse than twice the f32-input MFMA kernel on the same data
    assert errs["f32"] < 4e-6 and errs["bf16x6"] < 4e-6, errs
    assert errs["bf16x6"] <= 2.0 * errs["f32"] + 2.0 ** -24, errs


# ------------------------------------------------------------------------------------------------------------------
# optimiser: the reference's four param groups (nmt_multimodal_beam_DE.py:316-329) on vag_clip_adam_flat
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("vse_separate", [False, True])
def test_clip_adam_segments_equal_torch_adam_groups(vse_separate):
    from vagnmt_hip.trainer import TrainStep, param_groups
    meta, P, z = load_golden("mm_dot_tied_s0_f32")
    cm, cv = criteria(meta)
    src, lens, tgt, im = _inputs(meta, z)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    m = build(meta, P)
    lr, wd, clip = 4e-4, 1e-5, 0.05          # a clip small enough to bite
    ts = TrainStep(m, cm, cv, lr=lr, weight_decay=wd, clip=clip, vse_separate=vse_separate, use_graph=False)
    m.eval()
    # torch reference on CPU copies
    ref = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in m.named_parameters()}
    named = list(ref.items())
    groups = [{"params": [ref[n] for n in names], "weight_decay": wd if decay else 0.0, "lr": lr * mult}
              for _, names, decay, mult in param_groups(named, vse_separate)]
    opt = torch.optim.Adam(groups, lr=lr)
    for it, cur_lr in enumerate((lr, lr * 0.2)):          # second step after a ReduceLROnPlateau cut (set_lr)
        ts.set_lr(cur_lr)
        for g_, (_, _, _, mult) in zip(opt.param_groups, param_groups(named, vse_separate)):
            g_["lr"] = cur_lr * mult
        ts.backend.run(src, lt, tgt, im, True, 7)
        for n, p in m.named_parameters():
            ref[n].grad = p._vag_grad.detach().cpu().clone()
        total = torch.nn.utils.clip_grad_norm_([ref[n] for n in ref], clip)
        opt.step()
        ts._optimizer()
        assert abs(float(ts.grad_norm[0]) - float(total)) <= 1e-5 * float(total)
        assert float(ts.fp.grad.abs().max()) == 0.0            # left zeroed for the next step
        for n, p in m.named_parameters():
            err = (p.detach().cpu() - ref[n].detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, ref[n].abs().max().item()), (it, n, err)
    assert int(ts.step_count.item()) == 2


# ------------------------------------------------------------------------------------------------------------------
# bounded graph cache (samplers/bucket.py:59-60,93: any batch size 1..B, any (Ts, Tt))
# ------------------------------------------------------------------------------------------------------------------
def test_graph_cache_is_bounded_and_replays_match_eager():
    from machine_translation_vision.losses import PairwiseRankingLoss
    from vagnmt_hip.trainer import TrainStep
    Vs, Vt, I, E, H, S = 120, 140, 64, 32, 48, 40
    rnd = random.Random(3)
    shapes = []
    for _ in range(200):
        B = rnd.choice([16, 16, 16, 7, 1, 12])
        Ts, Tt = rnd.randint(3, 22), rnd.randint(2, 14)
        shapes.append((B, Ts, Tt, rnd.random() < 0.8))
    vw = torch.ones(Vt, device="cuda")
    vw[0] = 0
    cm = torch.nn.NLLLoss(weight=vw, reduction="none")
    runs = []
    for use_graph in (True, False):
        m, _, _, _ = make(Vs, Vt, I, E, H, S, 2, 3, 3, [3, 3], seed=5)
        m = m.cuda()
        ts = TrainStep(m, cm, PairwiseRankingLoss(0.1), use_graph=use_graph, max_graphs=8, pad_src=4)
        g = torch.Generator().manual_seed(9)
        losses = []
        mem = []
        for i, (B, Ts, Tt, teacher) in enumerate(shapes):
            lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=g)], reverse=True)
            lens[0] = Ts
            src = torch.zeros(B, Ts, dtype=torch.long)
            for b, L in enumerate(lens):
                src[b, :L] = torch.randint(4, Vs, (L,), generator=g)
            tgt = torch.randint(4, Vt, (B, Tt), generator=g)
            tgt[:, -1] = 3
            im = torch.randn(B, I, generator=g).abs()
            out = ts.step(src.cuda(), lens, tgt.cuda(), im.cuda(), teacher=teacher)
            losses.append(float(out[0]))
            if i in (120, 199):
                torch.cuda.synchronize()
                mem.append(torch.cuda.memory_allocated())
        runs.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}, mem, dict(ts.stats), len(ts._graphs)))
    (lg, pg, memg, stats, ngraphs), (le, pe, _, _, _) = runs
    assert ngraphs <= 8 and stats["evictions"] > 0 and stats["captures"] > 8 and stats["replays"] > 0, stats
    assert memg[1] <= memg[0] * 1.02 + (1 << 20), memg            # no per-shape memory pools pile up
    assert np.allclose(lg, le, rtol=5e-4, atol=1e-5), max(abs(a - b) for a, b in zip(lg, le))
    for n in pg:
        assert (pg[n] - pe[n]).abs().max().item() <= 1e-3 * max(1.0, pe[n].abs().max().item()), n


# ------------------------------------------------------------------------------------------------------------------
# checkpoint: N steps -> save -> load into a fresh model/driver -> M steps == N+M uninterrupted steps
# ------------------------------------------------------------------------------------------------------------------
def test_checkpoint_resume_equivalence(tmp_path):
    from machine_translation_vision.losses import PairwiseRankingLoss
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip.checkpoint import load_checkpoint, save_checkpoint
    from vagnmt_hip.trainer import TrainStep
    Vs, Vt, I, E, H, S, B, Ts, Tt = 90, 100, 64, 32, 48, 40, 6, 9, 7

    def new_model(seed):
        torch.manual_seed(seed)
        return NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, dropout_ctx=0.5, dropout_emb=0.3,
                                                     dropout_out=0.5, tied_emb=True).cuda()
    vw = torch.ones(Vt, device="cuda")
    vw[0] = 0
    cm = torch.nn.NLLLoss(weight=vw, reduction="none")
    g = torch.Generator().manual_seed(1)
    batches = []
    for _ in range(5):
        src = torch.randint(4, Vs, (B, Ts), generator=g)
        tgt = torch.randint(4, Vt, (B, Tt), generator=g)
        tgt[:, -1] = 3
        batches.append((src.cuda(), [Ts] * B, tgt.cuda(), torch.randn(B, I, generator=g).abs().cuda()))
    coins = [True, False, True, True, False]
    # uninterrupted
    torch.manual_seed(11)           # the dropout seed is drawn from torch's generator on first use
    m_a = new_model(0)
    ts_a = TrainStep(m_a, cm, PairwiseRankingLoss(0.1), use_graph=True)
    la = [float(ts_a.step(*b, teacher=c)[0]) for b, c in zip(batches, coins)]
    # 3 steps, save, resume in fresh objects (different init, different dropout seed before loading)
    torch.manual_seed(11)
    m_b = new_model(0)
    ts_b = TrainStep(m_b, cm, PairwiseRankingLoss(0.1), use_graph=True)
    lb = [float(ts_b.step(*b, teacher=c)[0]) for b, c in zip(batches[:3], coins[:3])]
    path = str(tmp_path / "resume.pt")
    save_checkpoint(path, m_b, ts_b)
    torch.manual_seed(99)
    m_c = new_model(123)
    ts_c = TrainStep(m_c, cm, PairwiseRankingLoss(0.1), use_graph=True)
    ts_c.step(*batches[0], teacher=True)          # the resumed driver already holds captured graphs and an rng
    ts_c.step(*batches[0], teacher=True)
    load_checkpoint(path, m_c, ts_c)
    lc = [float(ts_c.step(*b, teacher=c)[0]) for b, c in zip(batches[3:], coins[3:])]
    assert np.allclose(la[:3], lb, rtol=1e-5)
    assert np.allclose(la[3:], lc, rtol=2e-5), (la[3:], lc)
    for (n, pa), (_, pc) in zip(m_a.named_parameters(), m_c.named_parameters()):
        assert (pa - pc).abs().max().item() <= 2e-5 * max(1.0, pa.abs().max().item()), n
    assert torch.equal(m_a._vag_rng, m_c._vag_rng)
    assert int(ts_c.step_count) == 5 and torch.allclose(ts_a.fp.m, ts_c.fp.m, rtol=1e-3, atol=1e-7)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at its real size: V=9391, H=512, eval batch 16, beam 12, max_length 80
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_beam12_at_config_size_matches_oracle():
    from oracle import vag_oracle as O
    lens = [40, 33, 30, 27, 25, 22, 20, 18, 17, 15, 13, 11, 9, 7, 5, 3]
    m, src, tgt, im = make(8507, 9391, 2048, 256, 512, 512, 16, 40, 8, lens, seed=21)
    with torch.no_grad():
        m.decoder.out.bias[3] += 2.0          # let some hypotheses finish inside the 80 steps (finished-beam rules)
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, want_scores = O.beam_search(P, src, lens, im, beam_size=12, max_length=80, return_scores=True)
    want_g = O.greedy_decode(P, src, lens, im, max_length=80)
    mg = m.cuda().eval()
    res = {}
    for graph in (False, True):
        mg.decode_graph = graph
        got = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 12, 80)]
        res[graph] = (got, mg.last_beam_scores.cpu().numpy().copy(),
                      [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 1, 80)])
    # graph replay against launch-by-launch: same kernels, but the B*k = 192-row vocabulary product takes a split-K path
    # whose fp32 atomics land in a run-dependent order, so scores agree to rounding rather than bit for bit at this size
    assert np.allclose(res[True][1], res[False][1], rtol=1e-5, atol=1e-5)
    assert sum(a == b for a, b in zip(res[True][0], res[False][0])) >= 15
    assert sum(a == b for a, b in zip(res[True][2], res[False][2])) >= 15
    got, scores, got_g = res[True]
    # 112 692 candidates per sentence and step: a 1e-6 rounding difference may swap two near-tied low-ranked beams, so the
    # selection is compared through the normalised score of the winner (tight) and the token lists (all but at most one)
    assert np.allclose(scores, want_scores.numpy(), rtol=2e-4, atol=2e-4), np.abs(scores - want_scores.numpy()).max()
    assert sum(a == b for a, b in zip(got, want)) >= 15, [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert sum(a == b for a, b in zip(got_g, want_g)) >= 15


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: 2-byte ("fp16") storage of what the recurrences stream per time step, fp32 accumulation
# ------------------------------------------------------------------------------------------------------------------
def _fp16_case(name):
    if name == "mid":
        meta, P, z = load_golden("mm_dot_tied_mid_f32")            # H = 64, B = 16, Ts = Tt = 12, ragged
        m_of = lambda: build(meta, P)                               # noqa: E731
        src, lens, tgt, im = _inputs(meta, z)
        cm, cv = criteria(meta)
        return m_of, (src, lens, tgt, im), cm, cv
    # "wide": configs[4] widths (H=1024, B=256, 2048-d features) at reduced length / vocabulary -- the fp16 twin of
    # test_edge_shapes[wide]
    lens = sorted([int(x) for x in torch.randint(1, 7, (256,), generator=torch.Generator().manual_seed(9))], reverse=True)
    lens[0] = 6
    m0, src, tgt, im = make(700, 1003, 2048, 256, 1024, 512, 256, 6, 5, lens, seed=7)
    state = {k: v.clone() for k, v in m0.state_dict().items()}

    def m_of():
        m, _, _, _ = make(700, 1003, 2048, 256, 1024, 512, 256, 6, 5, lens, seed=7)
        m.load_state_dict(state)
        return m.cuda()
    from machine_translation_vision.losses import PairwiseRankingLoss
    vw = torch.ones(1003, device="cuda")
    vw[0] = 0
    return m_of, (src.cuda(), lens, tgt.cuda(), im.cuda()), torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1)


@pytest.mark.parametrize("name", ["mid", "wide"])
def test_fp16_storage_step_against_fp32(name):
    """fp16 storage changes what is READ per time step (weights, keys), not the arithmetic: losses within 2e-3 of the fp32
    path (itself within 1e-4 of the reference), gradients within 1e-2 of each tensor's largest entry; and the mode is
    really on (the keys in the workspace are fp16 bit patterns, the loss differs from the fp32 one)."""
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    m_of, (src, lens, tgt, im), cm, cv = _fp16_case(name)
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = {}
    for storage in ("f32", "f16"):
        m = m_of()
        ts = TrainStep(m, cm, cv, use_graph=False, storage=storage, pad_src=1)
        m.eval()
        ts.backend.run(src, lt, tgt, im, True, 7)
        losses = [float(x) for x in ts.backend.outputs()]
        grads = {n: p._vag_grad.detach().clone() for n, p in m.named_parameters()}
        f = ts.backend.f
        B, Ts = src.shape
        c = f.cfg(B, Ts, tgt.shape[1], True, False)
        # pe sits right after enc, mask and the encoder workspace: read its first words as stored
        off_enc = L.lib().vag_step_ws_offset(C.byref(c), 0)
        out[storage] = (losses, grads, int(c.storage), off_enc)
        if storage == "f16":
            ts._optimizer()          # the optimiser refreshes the fp16 weight copies: a second step must still work
            ts.backend.run(src, lt, tgt, im, True, 7)
            assert np.isfinite([float(x) for x in ts.backend.outputs()]).all()
    (l32, g32, s32, _), (l16, g16, s16, _) = out["f32"], out["f16"]
    assert s32 == 0 and s16 == 1
    assert np.allclose(l16, l32, rtol=2e-3, atol=2e-4), (l16, l32)
    assert l16 != l32                                   # not silently the fp32 path
    for n in g32:
        scale = max(g32[n].abs().max().item(), 1e-6)
        err = (g16[n] - g32[n]).abs().max().item()
        assert err <= 1e-2 * scale, (n, err, scale)


def test_fp16_storage_trains_and_free_running_steps_fall_back_to_fp32_storage():
    from vagnmt_hip.trainer import TrainStep
    m_of, (src, lens, tgt, im), cm, cv = _fp16_case("mid")
    m = m_of()
    ts = TrainStep(m, cm, cv, use_graph=True, storage="f16")
    losses = [float(ts.step(src, lens, tgt, im, teacher=(i % 3 != 2))[0]) for i in range(9)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses


# ------------------------------------------------------------------------------------------------------------------
# output head in row chunks: the (Tt*B, V) logits are never formed as a whole (forward per chunk, backward recomputes)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("storage", ["f32", "f16"])
@pytest.mark.parametrize("chunk_steps", [1, 5])           # 5 does not divide Tt = 12: a ragged last chunk
@pytest.mark.parametrize("mode", ["fused", "recompute", "split_phases"])
def test_chunked_head_equals_whole_sequence_head(storage, chunk_steps, mode):
    """fused: forward+backward in one call -- each chunk is finished (d(logits) and its three products) in the forward.
    recompute / split_phases: the backward builds each chunk's logits again (forced by the switch, or because the
    backward is a separate call)."""
    from vagnmt_hip.trainer import TrainStep
    from vagnmt_hip import _lib as L
    m_of, (src, lens, tgt, im), cm, cv = _fp16_case("mid")
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    B = src.shape[0]
    out = []
    for chunk in (0, chunk_steps * B):
        L.set_option("head_chunk", chunk)
        L.set_option("head_fuse", 0 if mode == "recompute" else 1)
        m = m_of()
        ts = TrainStep(m, cm, cv, use_graph=False, storage=storage, pad_src=1)
        m.train()                                   # dropout on: the recomputation must see the same masks
        if mode == "split_phases" and chunk:
            ts.backend.run(src, lt, tgt, im, True, 1)
            ts.backend.run(src, lt, tgt, im, True, 6)
        else:
            ts.backend.run(src, lt, tgt, im, True, 7)
        out.append(([float(x) for x in ts.backend.outputs()], ts.fp.grad.detach().clone()))
    L.set_option("head_chunk", -1)
    L.set_option("head_fuse", 1)
    (l0, g0), (l1, g1) = out
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), (l0, l1)
    # same products on the same operands chunk by chunk; only the summation order of the weight-gradient sums over rows
    # (and the split-K atomics) differs
    scale = g0.abs().max().item()
    assert (g0 - g1).abs().max().item() <= 2e-5 * scale, ((g0 - g1).abs().max().item(), scale)
    assert g0.abs().sum().item() > 0


# ------------------------------------------------------------------------------------------------------------------
# beam expansion at the kernel level: the selection (lane-maxima bound + rank counting, radix search under heavy ties)
# against a numpy restatement of models/...V11.py:279-313 with the total order (score desc, flat index asc)
# ------------------------------------------------------------------------------------------------------------------
def _beam_step_reference(logp, nll, prev_tok, di, k_in, k, V):
    B = logp.shape[0] // k_in
    words = np.zeros((B, k), dtype=np.int64)
    parents = np.zeros((B, k), dtype=np.int64)
    scores = np.zeros((B, k), dtype=np.float32)
    for b in range(B):
        cand = np.empty(k_in * V, dtype=np.float32)
        for j in range(k_in):
            n = b * k_in + j
            lp = logp[n, :V].copy()
            base = np.float32(0.0)
            if di > 0:
                pt = prev_tok[n]
                if pt == 3:
                    lp[:] = np.float32(-1e5)
                    lp[3] = 0.0
                else:
                    lp[pt] = np.float32(-1e5)
                base = nll[n]
            cand[j * V:(j + 1) * V] = base + lp
        order = np.lexsort((np.arange(cand.size), -cand.astype(np.float64)))[:k]
        words[b], parents[b], scores[b] = order % V, order // V, cand[order]
    return words, parents, scores


@pytest.mark.parametrize("case", ["random", "quantised", "constant", "small_vocab"])
def test_beam_expansion_selection_matches_total_order(case):
    from vagnmt_hip import _lib as L
    from vagnmt_hip._lib import call, ptr
    B, k, V, H, ML = (5, 12, 9391, 32, 10) if case != "small_vocab" else (3, 7, 37, 8, 6)
    ldl = (V + 3) // 4 * 4
    g = torch.Generator().manual_seed(3)
    for di in (0, 2):
        k_in = 1 if di == 0 else k
        N = B * k_in
        x = torch.randn(N, ldl, generator=g)
        if case == "quantised":
            x = (x * 2).round() / 2             # thousands of exact ties on every value
        elif case == "constant":
            x = torch.zeros(N, ldl)             # everything ties: winners are the smallest flat indices
        logp = x.cuda().contiguous()
        nll = (torch.randn(B * k, generator=g).round() if case != "random" else torch.randn(B * k, generator=g)).cuda()
        beam = torch.zeros(2 * ML, B, k, dtype=torch.int64)
        if di > 0:
            beam[di - 1] = torch.randint(0, V, (B, k), generator=g)
            beam[di - 1, 0, 0] = 3              # a finished hypothesis
            beam[di - 1, 1, :] = 3              # a sentence whose hypotheses have all finished
        beam = beam.cuda()
        h_in = torch.randn(N, H, generator=g).cuda()
        h_out = torch.zeros(B * k, H, device="cuda")
        n_alive = torch.zeros(1, dtype=torch.int32, device="cuda")
        scratch = torch.empty(L.lib().vag_beam_scratch_bytes(B, k, V, ML), dtype=torch.uint8, device="cuda")
        nll_in = nll.clone()
        call("vag_beam_step", ptr(logp), ldl, ptr(nll), ptr(beam, torch.int64), di, ML, ptr(h_in), ptr(h_out), B, k, V, H,
             ptr(n_alive, torch.int32), scratch.data_ptr(), L.stream())
        torch.cuda.synchronize()
        prev = beam[di - 1].reshape(-1).cpu().numpy() if di > 0 else None
        w, p, s = _beam_step_reference(logp.cpu().numpy(), nll_in.cpu().numpy(), prev, di, k_in, k, V)
        assert np.array_equal(beam[di].cpu().numpy(), w), (case, di)
        assert np.array_equal(beam[ML + di].cpu().numpy(), p), (case, di)
        assert np.array_equal(nll.cpu().numpy().reshape(B, k), s), (case, di)
        want_h = h_in.cpu().numpy().reshape(B, k_in, H)[np.arange(B)[:, None], p]
        assert np.array_equal(h_out.cpu().numpy().reshape(B, k, H), want_h), (case, di)
        assert int(n_alive.item()) == int((w != 3).sum()), (case, di)
