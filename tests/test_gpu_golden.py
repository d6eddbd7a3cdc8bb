"""GPU parity of the drop-in models against the golden vectors produced by the reference itself
(tests/golden/, see oracle/make_golden.py) -- fp32, tolerance 1e-4 as stated in BASELINE.json."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

F32_CASES = ["mm_dot_tied_s0_f32", "mm_mlp_untied_s1_f32", "text_tied_s0_f32", "mm_dot_tied_mid_f32",
             "mm_dot_full_len_f32"]
TOL = 1e-4


def build(meta, P):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11, NMT_Seq2Seq_Beam_V2
    Vs, Vt, I, E, H, S, B, Ts, Tt = meta["dims"]
    if meta["kind"] == "mm":
        m = NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, meta["loss_w"], attn_model=meta["attn"],
                                                  tied_emb=meta["tied"], init_split=meta["init_split"])
    else:
        m = NMT_Seq2Seq_Beam_V2(Vs, Vt, E, E, H, tied_emb=meta["tied"])
    missing, unexpected = m.load_state_dict(P, strict=False)
    assert not unexpected and set(missing) <= {"decoder.out.weight"}, (missing, unexpected)
    return m.cuda().eval()


def criteria(meta):
    from machine_translation_vision.losses import PairwiseRankingLoss, ImageRetrievalRankingLoss
    Vt = meta["dims"][1]
    vw = torch.ones(Vt)
    vw[0] = 0
    cm = torch.nn.NLLLoss(weight=vw.cuda(), reduction="none")
    cv = (PairwiseRankingLoss if meta["vse_loss"] == "pairwise" else ImageRetrievalRankingLoss)(margin=meta["margin"])
    return cm, cv


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), "%s: max abs err %.3e (scale %.3e)" % (what, err, np.abs(b).max())


def run_forward(m, meta, z, tfr):
    src, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda()
    cm, cv = criteria(meta)
    if meta["kind"] == "mm":
        return m(src, meta["lengths"], tgt, torch.from_numpy(z["im"]).cuda(), tfr, criterion_mt=cm, criterion_vse=cv)
    loss = m(src, meta["lengths"], tgt, tfr, criterion=cm)
    return loss, loss, torch.zeros(())


@pytest.mark.parametrize("name", F32_CASES)
def test_losses_teacher_and_free(name):
    meta, P, z = load_golden(name)
    m = build(meta, P)
    with torch.no_grad():
        l, lm, lv = run_forward(m, meta, z, 1.0)
        close(l, z["teacher/loss"], what="loss")
        close(lm, z["teacher/loss_mt"], what="loss_mt")
        if meta["kind"] == "mm":
            close(lv, z["teacher/loss_vse"], what="loss_vse")
        l, lm, lv = run_forward(m, meta, z, 0.0)
        close(l, z["free/loss"], what="free loss")
        close(lm, z["free/loss_mt"], what="free loss_mt")


@pytest.mark.parametrize("name", F32_CASES)
def test_intermediates(name):
    meta, P, z = load_golden(name)
    m = build(meta, P)
    src = torch.from_numpy(z["src"]).cuda()
    with torch.no_grad():
        enc, mask = m.encoder(src, meta["lengths"])           # layer API: (Ts,B,2H), (Ts,B)
        close(enc, z["enc"], what="enc")
        close(mask, z["mask"], 0, "mask")
        if meta["kind"] == "mm":
            im = torch.from_numpy(z["im"]).cuda()
            ie, te = m.embed_sent_im_test(src, meta["lengths"], im)
            close(ie, z["im_emb"], what="im_emb")
            close(te, z["txt_emb"], what="txt_emb")
            aw = m.get_imagine_attention_test(src, meta["lengths"], im)
            close(aw[:, 0, :], z["alpha_vse"], what="alpha_vse")
        # per-step layer API of the decoder (teacher-forced inputs), against the reference's step outputs
        tgt = torch.from_numpy(z["tgt"]).cuda()
        B = src.shape[0]
        if meta["kind"] == "mm":
            _, ctx = m.vse_imagine(im, enc, context_mask=mask)
            from vagnmt_hip import ops
            h = ops.DecInit.apply(enc.transpose(0, 1).contiguous(), mask.t().contiguous(), ctx, m.decoderini.weight,
                                  m.decoderini.bias, m.init_split)
        else:
            from vagnmt_hip import ops
            h = ops.DecInit.apply(enc.transpose(0, 1).contiguous(), mask.t().contiguous(), None, m.decoderini.weight,
                                  m.decoderini.bias, 0.0)
        h = h.unsqueeze(0)
        tok = torch.full((B,), 2, dtype=torch.long, device="cuda")
        for di in range(tgt.shape[1]):
            logp, h = m.decoder(tok, h, enc, ctx_mask=mask)
            # north_star: logits within 1e-4 of the reference.  Observed on MI355X (round 5, all five fixtures, every step):
            # max |logp - reference| 9.5e-7 .. 1.4e-6 at max |logp| ~7; the bound is 2e-5 ABSOLUTE (was 2e-4 x max|logp| ~ 1.4e-3)
            err = float((logp.detach().cpu().double() - torch.from_numpy(z["logp_steps"][di]).double()).abs().max())
            assert err <= 2e-5, "logp step %d: max abs err %.3e" % (di, err)
            close(h[0], z["h2_steps"][di], what="h2 step %d" % di)
            tok = tgt[:, di]


@pytest.mark.parametrize("name", F32_CASES)
def test_gradients(name):
    meta, P, z = load_golden(name)
    m = build(meta, P)
    loss, _, _ = run_forward(m, meta, z, 1.0)
    loss.backward()
    tot = 0.0
    for n, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        close(g, z["G/" + n], 2e-4, "grad " + n)
        tot += float((g.double() ** 2).sum())
    close(np.sqrt(tot), z["grad_norm"], 2e-4, "grad_norm")


@pytest.mark.parametrize("name", F32_CASES)
def test_decode(name):
    meta, P, z = load_golden(name)
    m = build(meta, P)
    src = torch.from_numpy(z["src"]).cuda()
    for k, want in meta["decode"].items():
        k = int(k)
        if meta["kind"] == "mm":
            got = m.beamsearch_decode(src, meta["lengths"], torch.from_numpy(z["im"]).cuda(), k, meta["max_len"])
        else:
            got = m.beamsearch_decode(src, meta["lengths"], k, meta["max_len"])
        got = [[int(t) for t in h] for h in got]
        assert got == want, (name, k, got, want)


def test_ranking_losses_and_l2norm():
    import os
    from conftest import GOLDEN
    from machine_translation_vision.losses import PairwiseRankingLoss, ImageRetrievalRankingLoss
    from machine_translation_vision.utils.utils import l2norm
    z = dict(np.load(os.path.join(GOLDEN, "losses.npz")))
    for key in [k for k in z if k.endswith("/im") and k.startswith("B")]:
        pre = key[:-3]
        mg = float(pre.split("_m")[1])
        im, s = torch.from_numpy(z[pre + "/im"]).cuda(), torch.from_numpy(z[pre + "/s"]).cuda()
        close(PairwiseRankingLoss(mg)(im, s), z[pre + "/pairwise"], what=pre)
        close(ImageRetrievalRankingLoss(mg)(im, s), z[pre + "/imageretrieval"], what=pre)
    close(l2norm(torch.from_numpy(z["l2norm/x"]).cuda()), z["l2norm/y"], 1e-6, "l2norm")


def test_retrieval_metrics_t2i_i2t():
    import os
    from conftest import GOLDEN
    from machine_translation_vision.utils import im_retrieval_eval as RE
    z = dict(np.load(os.path.join(GOLDEN, "losses.npz")))
    for N in (7, 100):
        im, cap = torch.from_numpy(z["retr%d/im" % N]).cuda(), torch.from_numpy(z["retr%d/cap" % N]).cuda()
        assert list(RE.t2i(im, cap)) == list(z["retr%d/t2i" % N])      # integer-exact statistics
        assert list(RE.i2t(im, cap)) == list(z["retr%d/i2t" % N])
    # a larger case against the oracle (N = 1014 like the Multi30K validation set)
    from oracle import vag_oracle as O
    g = torch.Generator().manual_seed(9)
    im = torch.nn.functional.normalize(torch.randn(1014, 512, generator=g))
    cap = torch.nn.functional.normalize(im + 0.08 * torch.randn(1014, 512, generator=g))
    assert RE.t2i(im.cuda(), cap.cuda()) == O.t2i(im, cap)
    assert RE.i2t(im.cuda(), cap.cuda()) == O.i2t(im, cap)


def test_device_resident_batch_pipeline():
    """vagnmt_hip.data: batches assembled on the device equal the batches the reference's own generator produced
    (preprocessing.py:308-384; tests/golden/batches.npz from oracle/make_golden.py:run_batches)."""
    import os
    from conftest import GOLDEN
    from vagnmt_hip.data import DeviceCorpus, data_generator_tl_mtv
    z = dict(np.load(os.path.join(GOLDEN, "batches.npz")))
    N = z["x"].shape[0]
    pairs = [[[int(v) for v in z["x"][i, :z["x_len"][i]]], [int(v) for v in z["y"][i, :z["y_len"][i]]]] for i in range(N)]
    feats = z["feats"]
    corpus = DeviceCorpus(pairs, feats, torch.device("cuda:0"))
    for bs in (16, 5):
        np.random.seed(11)
        got = list(data_generator_tl_mtv(corpus, bs))
        assert len(got) == int(z["bs%d/n" % bs][0]) and sum(g[0].shape[0] for g in got) == N
        for k, g in enumerate(got):
            assert np.array_equal(g[0].cpu().numpy(), z["bs%d/%d/x" % (bs, k)])
            assert np.array_equal(g[1].cpu().numpy(), z["bs%d/%d/y" % (bs, k)])
            assert np.array_equal(g[2].cpu().numpy(), z["bs%d/%d/im" % (bs, k)])
            assert g[3] == list(z["bs%d/%d/xl" % (bs, k)]) and g[4] == list(z["bs%d/%d/yl" % (bs, k)])
            assert g[3] == sorted(g[3], reverse=True) and len(set(g[4])) == 1
    np.random.seed(11)
    got = list(data_generator_tl_mtv(corpus, 16))
    # data-parallel sharding: a common seed, disjoint batches, the same number of batches on every rank
    np.random.seed(11)
    state = np.random.get_state()[1].copy()
    r0 = list(data_generator_tl_mtv(corpus, 16, rank=0, world_size=2, seed=11))
    r1 = list(data_generator_tl_mtv(corpus, 16, rank=1, world_size=2, seed=11))
    assert np.array_equal(np.random.get_state()[1], state)          # the global generator is left as it was
    assert len(r0) == len(r1) == len(got) // 2 and torch.equal(r0[0][0], got[0][0]) and torch.equal(r1[0][0], got[1][0])
    with pytest.raises(ValueError):
        list(data_generator_tl_mtv(corpus, 16, rank=0, world_size=2))
