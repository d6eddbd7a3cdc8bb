"""Pin the CPU oracle (oracle/vag_oracle.py) to outputs of the reference itself.

The fixtures in tests/golden/ were produced by oracle/make_golden.py, which imports
and runs the reference package; nothing here reads /root/reference."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle import vag_oracle as O


def tol(meta):
    return (1e-12, 1e-10) if meta["dtype"] == "float64" else (2e-6, 2e-5)


def close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a.detach() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= atol + rtol * np.abs(b).max(), "%s: max err %.3e" % (what, err)


def fw_kwargs(meta):
    return dict(loss_w=meta["loss_w"], init_split=meta["init_split"], attn=meta["attn"],
                vse_loss=meta["vse_loss"], margin=meta["margin"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_and_intermediates(name):
    meta, P, z = load_golden(name)
    a, g = tol(meta)
    src, tgt = torch.from_numpy(z["src"]), torch.from_numpy(z["tgt"])
    im = torch.from_numpy(z["im"]) if meta["kind"] == "mm" else None
    out = O.model_forward(P, src, meta["lengths"], tgt, im, teacher=True, keep=True, **fw_kwargs(meta))
    close(out["enc"], z["enc"], a, what="enc")
    close(out["mask"], z["mask"], 0, what="mask")
    close(out["loss"], z["teacher/loss"], a, 1e-6, "loss")
    close(out["loss_mt"], z["teacher/loss_mt"], a, 1e-6, "loss_mt")
    if im is not None:
        close(out["loss_vse"], z["teacher/loss_vse"], a, 1e-6, "loss_vse")
        close(out["im_emb"], z["im_emb"], a, what="im_emb")
        close(out["txt_emb"], z["txt_emb"], a, what="txt_emb")
        close(out["alpha_vse"], z["alpha_vse"], a, what="alpha_vse")
    close(torch.stack([s["logp"] for s in out["steps"]]), z["logp_steps"], 10 * a, what="logp")
    close(torch.stack([s["h2"] for s in out["steps"]]), z["h2_steps"], a, what="h2")
    close(torch.stack([s["alpha"] for s in out["steps"]]), z["alpha_steps"], a, what="alpha")
    # hoisted attn_e projection is the same mathematics
    out2 = O.model_forward(P, src, meta["lengths"], tgt, im, teacher=True, hoist=True, **fw_kwargs(meta))
    close(out2["loss"], z["teacher/loss"], a, 1e-6, "loss(hoist)")
    # free-running pass
    out3 = O.model_forward(P, src, meta["lengths"], tgt, im, teacher=False, **fw_kwargs(meta))
    close(out3["loss"], z["free/loss"], a, 1e-6, "free loss")
    close(out3["loss_mt"], z["free/loss_mt"], a, 1e-6, "free loss_mt")


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_backward_clip_adam(name):
    meta, P, z = load_golden(name)
    a, g = tol(meta)
    src, tgt = torch.from_numpy(z["src"]), torch.from_numpy(z["tgt"])
    im = torch.from_numpy(z["im"]) if meta["kind"] == "mm" else None
    out, grads, total, newP, _ = O.train_step(P, src, meta["lengths"], tgt, im, teacher=True, **fw_kwargs(meta))
    for n in P:
        close(grads[n], z["G/" + n], g, 1e-5, "grad " + n)
    close(total, z["grad_norm"], g, 1e-6, "grad_norm")
    if ("P1/" + next(iter(P))) in z:
        for n in P:
            close(newP[n], z["P1/" + n], 1e-12 if meta["dtype"] == "float64" else 2e-5, 0, "adam " + n)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_decode(name):
    meta, P, z = load_golden(name)
    src = torch.from_numpy(z["src"])
    im = torch.from_numpy(z["im"]) if meta["kind"] == "mm" else None
    kw = dict(init_split=meta["init_split"], attn=meta["attn"])
    for k, want in meta["decode"].items():
        k = int(k)
        if k == 1:
            got = O.greedy_decode(P, src, meta["lengths"], im, max_length=meta["max_len"], **kw)
        else:
            got = O.beam_search(P, src, meta["lengths"], im, beam_size=k, max_length=meta["max_len"], **kw)
        assert got == want, (name, k)


def test_ranking_losses_and_l2norm():
    z = dict(np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "losses.npz")))
    for key in [k for k in z if k.endswith("/im") and k.startswith("B")]:
        pre = key[:-3]
        m = float(pre.split("_m")[1])
        im, s = torch.from_numpy(z[pre + "/im"]), torch.from_numpy(z[pre + "/s"])
        close(O.pairwise_ranking_loss(im, s, m), z[pre + "/pairwise"], 1e-6, what=pre)
        close(O.image_retrieval_ranking_loss(im, s, m), z[pre + "/imageretrieval"], 1e-6, what=pre)
    close(O.l2norm(torch.from_numpy(z["l2norm/x"])), z["l2norm/y"], 1e-7, what="l2norm")


def test_retrieval_metrics():
    z = dict(np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "losses.npz")))
    for N in (7, 100):
        im, cap = torch.from_numpy(z["retr%d/im" % N]), torch.from_numpy(z["retr%d/cap" % N])
        assert list(O.t2i(im, cap)) == list(z["retr%d/t2i" % N])
        assert list(O.i2t(im, cap)) == list(z["retr%d/i2t" % N])


def _toy_corpus(z):
    pairs = [[[int(v) for v in z["x"][i, :z["x_len"][i]]], [int(v) for v in z["y"][i, :z["y_len"][i]]]]
             for i in range(z["x"].shape[0])]
    return pairs, z["feats"]


def test_batch_assembly_equals_the_reference_generator():
    """tests/golden/batches.npz: batches of preprocessing.data_generator_tl_mtv itself (:308-384) under numpy seed 11."""
    import os
    from conftest import GOLDEN
    from machine_translation_vision.samplers import BucketBatchSampler
    z = dict(np.load(os.path.join(GOLDEN, "batches.npz")))
    pairs, feats = _toy_corpus(z)
    for bs in (16, 5):
        np.random.seed(11)
        got = [O.assemble_batch(pairs, feats, b) for b in BucketBatchSampler([len(p[1]) for p in pairs], bs)]
        assert len(got) == int(z["bs%d/n" % bs][0])
        for k, (bx, by, bim, xl, yl) in enumerate(got):
            assert np.array_equal(bx.numpy(), z["bs%d/%d/x" % (bs, k)]) and np.array_equal(by.numpy(), z["bs%d/%d/y" % (bs, k)])
            assert np.array_equal(bim.numpy(), z["bs%d/%d/im" % (bs, k)])
            assert list(xl) == list(z["bs%d/%d/xl" % (bs, k)]) and list(yl) == list(z["bs%d/%d/yl" % (bs, k)])
