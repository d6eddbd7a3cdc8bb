"""GPU, round 4: the free-running decoder (models/...V11.py:148-160: training without teacher forcing; :207-226: greedy
decoding) as ONE launch -- the persistent recurrence kernel forms the head's hidden layer, the logits of its vocabulary
tiles and the arg-max itself and feeds the token back (persist.hip, dec_fwd_persistent_kernel<true>).  Checked against the
chain of per-step launches it replaces (same tokens, same saved tensors to fp32 rounding, same loss and gradients) and
against the CPU oracle; ragged batches (B not a multiple of 16), vocabulary sizes that are not a multiple of 16, output
dropout on."""
import numpy as np
import pytest
import torch

from test_gpu_edge_and_full import make, run_both

pytestmark = pytest.mark.gpu


def _opt(name, v):
    from vagnmt_hip import _lib as L
    assert L.lib().vag_set_option(name.encode(), int(v)) == 0


@pytest.fixture(autouse=True)
def _restore():
    yield
    _opt("free_persistent", 1)


def test_shapes_the_one_launch_form_takes():
    from vagnmt_hip import _lib as L
    ok = L.lib().vag_cgru_free_supported
    assert ok(64, 40, 40, 256, 512, 9391) == 1 and ok(16, 40, 80, 256, 512, 9391) == 1 and ok(1, 1, 1, 256, 512, 16) == 1
    assert ok(65, 40, 40, 256, 512, 9391) == 0          # five row tiles: more workgroups than CUs
    assert ok(64, 60, 40, 256, 512, 9391) == 0          # keys do not fit the LDS
    assert ok(64, 40, 40, 128, 512, 9391) == 0 and ok(64, 40, 40, 256, 256, 9391) == 0
    _opt("free_persistent", 0)
    assert ok(64, 40, 40, 256, 512, 9391) == 0


@pytest.mark.parametrize("B,Ts,Tt,Vt,p_out", [(64, 40, 7, 9391, 0.0), (37, 23, 9, 1003, 0.3), (5, 3, 4, 50, 0.0), (16, 43, 3, 4096, 0.5),
                                              (3, 2, 1, 40, 0.0), (33, 17, 2, 20011, 0.0)])
def test_operator_equals_the_launch_chain(B, Ts, Tt, Vt, p_out):
    """vag_cgru_attn_decode_free_fwd against vag_cgru_attn_decode_seq_fwd(free_run = 1) on the same inputs: the chosen tokens
    are identical, every output and every tensor saved for the backward pass agrees to fp32 rounding."""
    from vagnmt_hip import _lib as L
    from vagnmt_hip import ops
    lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=torch.Generator().manual_seed(B))], reverse=True)
    lens[0] = Ts
    m, src, tgt, im = make(60, Vt, 64, 256, 512, 48, B, Ts, Tt, lens, seed=B + Tt)
    with torch.no_grad():
        m.decoder.out.bias.normal_(0.0, 0.5, generator=torch.Generator().manual_seed(1))      # (the reference zeroes it)
    mg = m.cuda().eval()
    dec = mg.decoder
    g = torch.Generator().manual_seed(7)
    enc = (torch.randn(B, Ts, 1024, generator=g) * 0.5).cuda()
    mask = torch.zeros(B, Ts)
    for b, l in enumerate(lens):
        mask[b, :l] = 1
    mask = mask.cuda()
    enc = enc * mask.unsqueeze(-1)
    h0 = (torch.randn(B, 512, generator=g) * 0.5).cuda()
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda") if p_out > 0 else None
    ldl = (Vt + 3) // 4 * 4
    outs = {}
    with torch.no_grad():
        pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
        for persistent in (1, 0):
            _opt("free_persistent", persistent)
            tok = torch.zeros(Tt + 1, B, dtype=torch.int64, device="cuda")
            tok[0] = 2
            h2, c, e, tmid, logits = ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), free_run=True,
                                                         head=dec.head_params(), p_out=p_out, rng=rng, V=Vt, ldl=ldl)
            torch.cuda.synchronize()
            outs[persistent] = dict(tok=tok.cpu(), h2=h2.cpu(), c=c.cpu(), e=e.cpu(), tmid=tmid.cpu(),
                                    logits=logits.view(Tt, B, ldl)[:, :, :Vt].cpu())
    assert L.lib().vag_persistent_timeouts() == 0
    a, b = outs[1], outs[0]
    assert torch.equal(a["tok"], b["tok"]), (a["tok"] != b["tok"]).nonzero()[:5]
    assert int(a["tok"][1:].min()) >= 0 and int(a["tok"][1:].max()) < Vt
    for n in ("h2", "c", "e", "tmid", "logits"):
        err = (a[n] - b[n]).abs().max().item()
        assert err <= 2e-5 * max(1.0, b[n].abs().max().item()), (n, err)
    # the arg-max really is the arg-max of the logits it stored (first index on ties)
    assert torch.equal(a["logits"].argmax(-1), a["tok"][1:])


@pytest.mark.parametrize("train_mode", [False, True])
def test_free_running_train_step_equals_the_launch_chain(train_mode):
    """The fused training step, free running, at configs[1]'s widths: loss, tokens and every parameter gradient with the
    one-launch decoder against the launch chain (with and without dropout: fresh, equally seeded drivers)."""
    import ctypes as C
    from machine_translation_vision.losses import PairwiseRankingLoss
    from vagnmt_hip import _lib as L
    from vagnmt_hip.trainer import TrainStep
    B, Ts, Tt, Vt = 64, 40, 12, 2000
    lens = [Ts] * B
    res = []
    for persistent in (1, 0):
        _opt("free_persistent", persistent)
        m, src, tgt, im = make(300, Vt, 64, 256, 512, 48, B, Ts, Tt, lens, seed=11)
        m = m.cuda()
        vw = torch.ones(Vt, device="cuda")
        vw[0] = 0
        ts = TrainStep(m, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(0.1), use_graph=False, pad_src=1)
        m.train(train_mode)
        lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
        ts.backend.run(src.cuda(), lt, tgt.cuda(), im.cuda(), False, 7)
        torch.cuda.synchronize()
        f = ts.backend.f
        c = f.cfg(B, Ts, Tt, False, train_mode)
        off = L.lib().vag_step_ws_offset(C.byref(c), 8)
        tok = f.ws[off:off + 2 * (Tt + 1) * B].view(torch.int64)[: (Tt + 1) * B].view(Tt + 1, B).cpu().clone()
        res.append(([float(x) for x in ts.backend.outputs()], tok,
                    {n: p._vag_grad.detach().cpu().clone() for n, p in m.named_parameters()}))
    assert L.lib().vag_persistent_timeouts() == 0
    (la, ta, ga), (lb, tb, gb) = res
    assert torch.equal(ta, tb)
    assert np.allclose(la, lb, rtol=1e-5, atol=1e-6), (la, lb)
    for n in ga:
        err = (ga[n] - gb[n]).abs().max().item()
        assert err <= 3e-5 * max(gb[n].abs().max().item(), 1e-3), (n, err)


@pytest.mark.parametrize("B,Ts,Vt,L", [(37, 23, 1003, 11), (64, 40, 9391, 6), (3, 5, 40, 9)])
def test_greedy_decoding_in_one_launch_matches_chain_and_oracle(B, Ts, Vt, L):
    from oracle import vag_oracle as O
    lens = sorted([int(x) for x in torch.randint(1, Ts + 1, (B,), generator=torch.Generator().manual_seed(B))], reverse=True)
    lens[0] = Ts
    m, src, _, im = make(80, Vt, 64, 256, 512, 48, B, Ts, 3, lens, seed=B)
    P = {n: p.detach().clone() for n, p in m.named_parameters()}
    want = O.greedy_decode(P, src, lens, im, max_length=L)
    mg = m.cuda().eval()
    got = {}
    for persistent in (True, False):
        mg.decode_persistent = persistent
        got[persistent] = [[int(t) for t in h] for h in mg.beamsearch_decode(src.cuda(), lens, im.cuda(), 1, L)]
    assert got[True] == got[False]
    assert got[True] == want


def test_cfg2_free_running_full_size_matches_oracle_through_the_one_launch_form():
    """BASELINE.json configs[1] without teacher forcing (tests/test_gpu_edge_and_full.py runs the same case; here it is
    asserted that the one-launch form is what ran)."""
    from vagnmt_hip import _lib as L
    assert L.lib().vag_cgru_free_supported(64, 40, 40, 256, 512, 9391) == 1
    lens = [40] * 64
    m, src, tgt, im = make(8507, 9391, 2048, 256, 512, 512, 64, 40, 40, lens, seed=3)
    run_both(m, src, lens, tgt, im, teacher=False, check_grads=False)
    assert L.lib().vag_persistent_timeouts() == 0
