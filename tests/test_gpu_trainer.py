"""GPU: the step driver (flat parameters, fused clip+Adam, HIP-graph replay) and train-mode dropout parity."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_gpu_golden import build, criteria, close

pytestmark = pytest.mark.gpu


def _inputs(meta, z):
    src, tgt = torch.from_numpy(z["src"]).cuda(), torch.from_numpy(z["tgt"]).cuda()
    im = torch.from_numpy(z["im"]).cuda() if meta["kind"] == "mm" else None
    return src, meta["lengths"], tgt, im


@pytest.mark.parametrize("name", ["mm_dot_tied_s0_f32", "mm_mlp_untied_s1_f32", "text_tied_s0_f32"])
def test_one_step_matches_reference_adam(name):
    from vagnmt_hip.trainer import TrainStep
    meta, P, z = load_golden(name)
    m = build(meta, P)
    cm, cv = criteria(meta)
    ts = TrainStep(m, cm, cv if meta["kind"] == "mm" else None, lr=4e-4, weight_decay=1e-5, clip=1.0, use_graph=False)
    src, lens, tgt, im = _inputs(meta, z)
    loss, loss_mt, _ = ts.step(src, lens, tgt, im, teacher=True)
    close(loss, z["teacher/loss"], what="loss")
    close(ts.grad_norm[0], z["grad_norm"], 2e-4, "grad_norm")
    for n, p in m.named_parameters():
        close(p, z["P1/" + n], 2e-5, "post-Adam " + n)
    assert int(ts.step_count.item()) == 1


def test_graph_replay_equals_eager():
    from vagnmt_hip.trainer import TrainStep
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    cm, cv = criteria(meta)
    src, lens, tgt, im = _inputs(meta, z)
    res = []
    for use_graph in (False, True):
        m = build(meta, P)
        ts = TrainStep(m, cm, cv, use_graph=use_graph)
        losses = []
        for i in range(5):
            out = ts.step(src, lens, tgt, im, teacher=(i % 2 == 0))
            losses.append(float(out[0].item()))
        res.append((losses, {n: p.detach().clone() for n, p in m.named_parameters()}))
    assert np.allclose(res[0][0], res[1][0], rtol=2e-4), (res[0][0], res[1][0])
    assert res[0][0][4] < res[0][0][0]          # the loss goes down on a repeated batch
    for n in res[0][1]:
        close(res[1][1][n], res[0][1][n].cpu().numpy(), 2e-4, "graph vs eager " + n)


def test_train_mode_dropout_matches_oracle_with_same_masks():
    """Dropout masks are a pure function of (seed, step, stream, index): materialise them through the C ABI and feed
    the very same masks to the CPU oracle."""
    from machine_translation_vision.models import NMT_AttentionImagine_Seq2Seq_Beam_V11
    from vagnmt_hip import ops
    from oracle import vag_oracle as O
    meta, P, z = load_golden("mm_dot_tied_mid_f32")
    Vs, Vt, I, E, H, S, B, Ts, Tt = meta["dims"]
    pe_, pc_, po_ = 0.3, 0.5, 0.5
    m = NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, dropout_ctx=pc_, dropout_emb=pe_,
                                              dropout_out=po_, tied_emb=True)
    m.load_state_dict(P, strict=False)
    m = m.cuda().train()
    cm, cv = criteria(meta)
    src, lens, tgt, im = _inputs(meta, z)
    loss, loss_mt, loss_vse = m(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)
    loss.backward()
    rng = m._vag_rng
    masks = {
        "emb": ops.dropout_mask(rng, 1, Ts * B * E, pe_).view(Ts, B, E).cpu(),
        "ctx": ops.dropout_mask(rng, 2, B * Ts * 2 * H, pc_).view(B, Ts, 2 * H).transpose(0, 1).contiguous().cpu(),
        "out": ops.dropout_mask(rng, 3, Tt * B * E, po_).view(Tt, B, E).cpu(),
    }
    for k, p_ in (("emb", pe_), ("ctx", pc_), ("out", po_)):
        keep = float((masks[k] > 0).float().mean())
        assert abs(keep - (1 - p_)) < 0.03, (k, keep)
        vals = np.unique(masks[k].numpy())
        assert all(min(abs(v), abs(v - 1 / (1 - p_))) < 1e-5 for v in vals), vals
    leaves = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in m.named_parameters()}
    out = O.model_forward(leaves, src.cpu(), lens, tgt.cpu(), im.cpu(), teacher=True, masks=masks)
    out["loss"].backward()
    close(loss, out["loss"].detach().numpy(), what="train-mode loss")
    close(loss_vse, out["loss_vse"].detach().numpy(), what="train-mode loss_vse")
    for n, p in m.named_parameters():
        close(p.grad, leaves[n].grad.numpy(), 2e-4, "train-mode grad " + n)
    # a second forward uses a different step counter -> different masks
    loss2, _, _ = m(src, lens, tgt, im, 1.0, criterion_mt=cm, criterion_vse=cv)
    assert abs(float(loss2) - float(loss)) > 1e-6
