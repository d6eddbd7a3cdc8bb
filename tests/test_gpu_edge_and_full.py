"""GPU parity against the CPU oracle on shapes the golden fixtures do not cover: the full BASELINE cfg2 size, batch of
one, single-token sequences, odd vocabulary sizes, ragged lengths with length-1 rows; plus size-independent properties
at full size (attention rows sum to one, zero rows past each length, batch-row independence)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(Vs, Vt, I, E, H, S, B, Ts, Tt, lens, seed=0, attn="dot", tied=True, kind="mm"):
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11, NMT_Seq2Seq_Beam_V2
    torch.manual_seed(seed)
    if kind == "mm":
        m = NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, attn_model=attn, tied_emb=tied)
    else:
        m = NMT_Seq2Seq_Beam_V2(Vs, Vt, E, E, H, tied_emb=tied)
    g = torch.Generator().manual_seed(seed + 1)
    src = torch.zeros(B, Ts, dtype=torch.long)
    for b, L in enumerate(lens):
        src[b, :L] = torch.randint(4, Vs, (L,), generator=g)
    tgt = torch.randint(4, Vt, (B, Tt), generator=g)
    tgt[:, -1] = 3
    if B > 1 and Tt > 2:
        tgt[-1, 1] = 3
        tgt[-1, 2:] = 0
    im = torch.randn(B, I, generator=g).abs()
    return m, src, tgt, im


def run_both(m, src, lens, tgt, im, teacher=True, tol=1e-4, gtol=3e-4, check_grads=True, kind="mm"):
    from machine_translation_vision.losses import PairwiseRankingLoss
    from oracle import vag_oracle as O
    Vt = m.tgt_size
    vw = torch.ones(Vt)
    vw[0] = 0
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    leaves = {n: p.clone().requires_grad_(True) for n, p in P.items()}
    out = O.model_forward(leaves, src, lens, tgt, im if kind == "mm" else None, teacher=teacher, attn=m.attn_model
                          if kind == "mm" else "dot")
    out["loss"].backward()
    mg = m.cuda().eval()
    crit = torch.nn.NLLLoss(weight=vw.cuda(), reduction="none")
    tfr = 1.0 if teacher else 0.0
    if kind == "mm":
        loss, loss_mt, loss_vse = mg(src.cuda(), lens, tgt.cuda(), im.cuda(), tfr, criterion_mt=crit,
                                     criterion_vse=PairwiseRankingLoss(0.1))
        assert abs(float(loss_vse) - float(out["loss_vse"])) <= tol * max(1.0, abs(float(out["loss_vse"])))
    else:
        loss = loss_mt = mg(src.cuda(), lens, tgt.cuda(), tfr, criterion=crit)
    assert abs(float(loss) - float(out["loss"])) <= tol * max(1.0, abs(float(out["loss"]))), (float(loss), float(out["loss"]))
    assert abs(float(loss_mt) - float(out["loss_mt"])) <= tol * max(1.0, abs(float(out["loss_mt"])))
    if check_grads:
        loss.backward()
        for n, p in mg.named_parameters():
            ref = leaves[n].grad
            ref = ref if ref is not None else torch.zeros_like(P[n])
            err = (p.grad.cpu() - ref).abs().max().item()
            assert err <= gtol * max(ref.abs().max().item(), 1e-3), (n, err, ref.abs().max().item())
    return mg


@pytest.mark.parametrize("teacher", [True, False])
def test_cfg2_full_size_matches_oracle(teacher):
    """BASELINE.json configs[1]: B=64, Ts=Tt=40, E=256, H=512, S=512, I=2048, Vs=8507, V=9391 (eval mode)."""
    lens = [40] * 64
    m, src, tgt, im = make(8507, 9391, 2048, 256, 512, 512, 64, 40, 40, lens, seed=3)
    run_both(m, src, lens, tgt, im, teacher=teacher, check_grads=teacher)


def test_cfg2_ragged_lengths_and_properties():
    lens = sorted([int(x) for x in torch.randint(1, 41, (64,), generator=torch.Generator().manual_seed(5))], reverse=True)
    lens[0] = 40
    lens[-1] = 1
    m, src, tgt, im = make(8507, 9391, 2048, 256, 512, 512, 64, 40, 40, lens, seed=4)
    mg = run_both(m, src, lens, tgt, im, teacher=True, check_grads=False)
    with torch.no_grad():
        enc, mask = mg.encoder(src.cuda(), lens)                      # (Ts,B,2H)
        for b, L in enumerate(lens):
            assert float(enc[L:, b].abs().max()) == 0.0 if L < 40 else True     # zeros past each row's length
            assert float(mask[:L, b].min()) == 1.0 and (L == 40 or float(mask[L:, b].max()) == 0.0)
        aw = mg.get_imagine_attention_test(src.cuda(), lens, im.cuda())[:, 0, :]
        assert torch.allclose(aw.sum(1), torch.ones(64, device="cuda"), atol=1e-5)
        for b, L in enumerate(lens):
            if L < 40:
                assert float(aw[b, L:].abs().max()) == 0.0          # masked positions get exactly zero weight
        # batch-row independence: reversing the batch reverses the outputs (lengths need not stay sorted for our kernels)
        perm = torch.arange(63, -1, -1)
        enc2, _ = mg.encoder(src[perm].cuda(), [lens[i] for i in perm.tolist()])
        assert torch.allclose(enc2, enc[:, perm.cuda()], atol=1e-6)


@pytest.mark.parametrize("case", ["b1", "t1", "odd", "text_b1", "wide", "bigvocab"])
def test_edge_shapes(case):
    if case == "b1":        # a bucket's last batch can hold a single sentence (samplers/bucket.py:59-60,93)
        lens = [5]
        m, src, tgt, im = make(40, 45, 64, 16, 24, 20, 1, 5, 4, lens)
        run_both(m, src, lens, tgt, im)
    elif case == "t1":      # single-token source and target
        lens = [1, 1, 1]
        m, src, tgt, im = make(40, 45, 64, 16, 24, 20, 3, 1, 1, lens)
        run_both(m, src, lens, tgt, im)
    elif case == "odd":     # vocabulary sizes that are not multiples of 4, untied, mlp attention, free running
        lens = [9, 7, 7, 2, 1]
        m, src, tgt, im = make(51, 67, 100, 20, 28, 24, 5, 9, 7, lens, attn="mlp", tied=False)
        run_both(m, src, lens, tgt, im)
        m2, _, _, _ = make(51, 67, 100, 20, 28, 24, 5, 9, 7, lens, attn="mlp", tied=False)
        run_both(m2, src, lens, tgt, im, teacher=False, check_grads=False)
    elif case == "bigvocab":  # V > 10240: log-softmax rows no longer fit the register-resident path (head.hip, NV = 0)
        lens = [5, 3, 2]
        m, src, tgt, im = make(60, 10301, 64, 16, 24, 20, 3, 5, 4, lens, seed=5)
        run_both(m, src, lens, tgt, im)
        m2, _, _, _ = make(60, 10301, 64, 16, 24, 20, 3, 5, 4, lens, seed=5)
        run_both(m2, src, lens, tgt, im, teacher=False, check_grads=False)
    elif case == "wide":    # BASELINE configs[4] widths (H=1024, B=256, 2048-d features) at reduced length / vocabulary, fp32
        lens = sorted([int(x) for x in torch.randint(1, 7, (256,), generator=torch.Generator().manual_seed(9))], reverse=True)
        lens[0] = 6
        m, src, tgt, im = make(700, 1003, 2048, 256, 1024, 512, 256, 6, 5, lens, seed=7)
        run_both(m, src, lens, tgt, im)
    else:
        lens = [6]
        m, src, tgt, im = make(40, 45, 64, 16, 24, 20, 1, 6, 5, lens, kind="text")
        run_both(m, src, lens, tgt, None, kind="text")


def test_decode_matches_oracle_at_eval_batch():
    """test_multimodal.py setting at reduced width: eval batch 16, beam 12 and greedy, ragged sources."""
    from oracle import vag_oracle as O
    lens = [12, 11, 11, 9, 9, 8, 7, 7, 6, 5, 5, 4, 3, 2, 2, 1]
    m, src, tgt, im = make(300, 333, 256, 32, 64, 48, 16, 12, 8, lens, seed=11)
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    want_g = O.greedy_decode(P, src, lens, im, max_length=20)
    want_b, want_scores = O.beam_search(P, src, lens, im, beam_size=12, max_length=20, return_scores=True)
    mg = m.cuda().eval()
    got_g = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 1, 20)]
    got_b = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 12, 20)]
    assert got_g == want_g
    # topk(sorted=False) leaves the order of equal-score beams unspecified: compare the normalised scores too
    assert np.allclose(mg.last_beam_scores.cpu().numpy(), want_scores.numpy(), rtol=1e-4, atol=1e-4)
    assert got_b == want_b


def test_decode_graph_replay_equals_launch_by_launch():
    """The captured decode graphs (device-side step counter, source padded to 8 positions) against the same kernels
    launched step by step: identical hypotheses and scores, also when the cached graph is reused for another batch of a
    different true length, when max_length is not a multiple of the chunk, and on an early stop."""
    lens_a = [12, 11, 11, 9, 9, 8, 7, 7, 6, 5, 5, 4, 3, 2, 2, 1]
    lens_b = [10, 9, 9, 9, 8, 8, 7, 7, 6, 5, 5, 4, 3, 2, 2, 1]
    m, src_a, _, im_a = make(300, 333, 256, 32, 64, 48, 16, 12, 8, lens_a, seed=11)
    _, src_b, _, im_b = make(300, 333, 256, 32, 64, 48, 16, 10, 8, lens_b, seed=12)
    mg = m.cuda().eval()
    for max_len in (20, 13, 1):
        for k in (1, 3, 12):
            for src, lens, im in ((src_a, lens_a, im_a), (src_b, lens_b, im_b), (src_a, lens_a, im_a)):
                mg.decode_graph = False
                want = mg.beamsearch_decode(src.cuda(), lens, im.cuda(), k, max_len)
                want_s = mg.last_beam_scores.cpu().numpy() if k > 1 else None
                mg.decode_graph = True
                got = mg.beamsearch_decode(src.cuda(), lens, im.cuda(), k, max_len)
                assert [[int(t) for t in h] for h in got] == [[int(t) for t in h] for h in want], (max_len, k)
                if k > 1:
                    assert np.array_equal(mg.last_beam_scores.cpu().numpy(), want_s)
    # early stop: make EOS overwhelmingly likely so that every hypothesis finishes within the first chunk
    with torch.no_grad():
        mg.decoder.out.bias[3] += 50.0
    mg.decode_graph = False
    want = mg.beamsearch_decode(src_a.cuda(), lens_a, im_a.cuda(), 5, 40)
    mg.decode_graph = True
    got = mg.beamsearch_decode(src_a.cuda(), lens_a, im_a.cuda(), 5, 40)
    assert [[int(t) for t in h] for h in got] == [[int(t) for t in h] for h in want]
    assert all(len(h) == 0 for h in got)
