#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Usage (from the repo root; /root/reference must exist, it does not on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference package (``/root/reference/machine_translation_vision``) is imported
as-is, never copied.  It targets torch 0.4.1; three in-process compatibility
shims let it run on torch 2.x CPU:
  1. ``masked_fill_`` with uint8 masks  -> cast the mask to bool
     (layers/VSE_Imagine_Enc.py:43-44, layers/NMT_Decoder.py:42-43)
  2. unconditional ``.cuda()`` in the ranking losses -> identity on a CPU box
     (losses/PairwiseRankingLoss.py:16,18)
  3. integer ``/`` meant floor-division in torch 0.4 -> trunc-div for integral
     operands (models/...V11.py:242,245,303)
  4. (batch fixture only) ``nltk.tokenize`` stubbed in sys.modules: preprocessing.py:9
     imports it for sent_tokenize, which data_generator_tl_mtv never calls
The fixtures are plain arrays: parameters, inputs, and what the reference
computed from them (losses, intermediates, gradients, post-Adam parameters,
decoded token lists).
"""
import json
import os
import random
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = os.environ.get("VAG_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def install_shims():
    _mf = torch.Tensor.masked_fill_

    def masked_fill_(self, mask, value):
        if mask.dtype == torch.uint8:
            mask = mask.bool()
        return _mf(self, mask, value)

    torch.Tensor.masked_fill_ = masked_fill_
    torch.Tensor.cuda = lambda self, *a, **k: self
    _td = torch.Tensor.__truediv__

    def truediv(self, other):
        o_int = isinstance(other, int) or (torch.is_tensor(other) and not other.is_floating_point())
        if not self.is_floating_point() and o_int:
            return torch.div(self, other, rounding_mode="trunc")
        return _td(self, other)

    torch.Tensor.__truediv__ = truediv


def np32(t):
    return t.detach().cpu().numpy().copy()


def make_inputs(g, B, Ts, Tt, Vs, Vt, I, ragged=True):
    if ragged:
        lens = sorted([int(x) for x in torch.randint(2, Ts + 1, (B,), generator=g)], reverse=True)
        lens[0] = Ts
    else:
        lens = [Ts] * B
    src = torch.zeros(B, Ts, dtype=torch.long)
    for b, L in enumerate(lens):
        src[b, :L] = torch.randint(4, Vs, (L,), generator=g)
    tgt = torch.zeros(B, Tt, dtype=torch.long)
    for b in range(B):
        L = int(torch.randint(2, Tt + 1, (1,), generator=g)) if ragged else Tt
        if b == 0:
            L = Tt
        tgt[b, :L - 1] = torch.randint(4, Vt, (L - 1,), generator=g)
        tgt[b, L - 1] = 3
    im = torch.randn(B, I, generator=g).abs()
    return src, lens, tgt, im


def param_groups(model, wd):
    # nmt_multimodal_beam_DE.py:303-313
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    return [{"params": [p for n, p in named if "bias" not in n], "weight_decay": wd},
            {"params": [p for n, p in named if "bias" in n]}]


def run_case(name, kind, seed, dims, attn="dot", tied=True, vse_loss="pairwise", dtype=torch.float32,
             ragged=True, store_adam=True, beams=(2, 12), max_len=10):
    import machine_translation_vision.models as M
    import machine_translation_vision.losses as Lo
    Vs, Vt, I, E, H, S, B, Ts, Tt = dims
    torch.manual_seed(seed)
    if kind == "mm":
        model = M.NMT_AttentionImagine_Seq2Seq_Beam_V11(Vs, Vt, I, E, E, H, S, 0.99, attn_model=attn,
                                                        tied_emb=tied, init_split=0.5)
    else:
        model = M.NMT_Seq2Seq_Beam_V2(Vs, Vt, E, E, H, tied_emb=tied)
    model = model.to(dtype)
    model.eval()
    g = torch.Generator().manual_seed(1000 + seed)
    src, lens, tgt, im = make_inputs(g, B, Ts, Tt, Vs, Vt, I, ragged)
    im = im.to(dtype)
    vocab_mask = torch.ones(Vt, dtype=dtype)
    vocab_mask[0] = 0
    crit_mt = torch.nn.NLLLoss(weight=vocab_mask, reduction="none")
    margin = 0.1
    crit_vse = (Lo.PairwiseRankingLoss if vse_loss == "pairwise" else Lo.ImageRetrievalRankingLoss)(margin=margin)

    out = {}
    meta = dict(name=name, kind=kind, seed=seed, dims=list(dims), attn=attn, tied=tied, vse_loss=vse_loss,
                dtype=str(dtype).replace("torch.", ""), lengths=lens, margin=margin, loss_w=0.99,
                init_split=0.5, max_len=max_len, beams=list(beams))
    for n, p in model.named_parameters():
        out["P/" + n] = np32(p)
    out["src"] = src.numpy()
    out["tgt"] = tgt.numpy()
    out["im"] = np32(im)

    # ---- intermediates via hooks (teacher-forced pass) ----
    cap = {"dec": [], "attn": []}
    hk = [model.decoder.register_forward_hook(lambda m, i, o: cap["dec"].append((np32(o[0]), np32(o[1])))),
          model.decoder.attn.register_forward_hook(lambda m, i, o: cap["attn"].append(np32(o))),
          model.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("enc", (np32(o[0]), np32(o[1]))))]

    def fwd(tfr):
        if kind == "mm":
            return model(src, lens, tgt, im, tfr, criterion_mt=crit_mt, criterion_vse=crit_vse)
        loss = model(src, lens, tgt, tfr, criterion=crit_mt)
        return loss, loss, torch.zeros(())

    model.zero_grad()
    loss, loss_mt, loss_vse = fwd(1.0)      # random.random() < 1.0 -> teacher forcing
    for h in hk:
        h.remove()
    out["enc"] = cap["enc"][0]
    out["mask"] = cap["enc"][1]
    out["logp_steps"] = np.stack([d[0] for d in cap["dec"]])          # (Tt,B,V)
    out["h2_steps"] = np.stack([d[1][0] for d in cap["dec"]])          # (Tt,B,H)
    out["alpha_steps"] = np.stack([a[:, 0, :] for a in cap["attn"]])   # (Tt,B,Ts)
    out["teacher/loss"] = np32(loss)
    out["teacher/loss_mt"] = np32(loss_mt)
    out["teacher/loss_vse"] = np32(torch.as_tensor(loss_vse))
    if kind == "mm":
        with torch.no_grad():
            ie, te = model.embed_sent_im_test(src, lens, im)
            aw = model.get_imagine_attention_test(src, lens, im)
        out["im_emb"] = np32(ie)
        out["txt_emb"] = np32(te)
        out["alpha_vse"] = np32(aw)[:, 0, :]

    # ---- backward + clip + Adam (train.py:44-49) ----
    loss.backward()
    for n, p in model.named_parameters():
        out["G/" + n] = np32(p.grad) if p.grad is not None else np.zeros_like(np32(p))
    opt = torch.optim.Adam(param_groups(model, 1e-5), lr=4e-4)
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    out["grad_norm"] = np32(torch.as_tensor(total))
    if store_adam:
        opt.step()
        for n, p in model.named_parameters():
            out["P1/" + n] = np32(p)
        # restore step-0 parameters for the remaining passes
        with torch.no_grad():
            for n, p in model.named_parameters():
                p.copy_(torch.from_numpy(out["P/" + n]))

    # ---- free-running pass (teacher_force_ratio=0 -> random() < 0 is never true) ----
    with torch.no_grad():
        l2, lmt2, lv2 = fwd(0.0)
    out["free/loss"] = np32(l2)
    out["free/loss_mt"] = np32(lmt2)
    out["free/loss_vse"] = np32(torch.as_tensor(lv2))

    # ---- decode ----
    dec = {}
    with torch.no_grad():
        for k in (1,) + tuple(beams):
            if kind == "mm":
                hyp = model.beamsearch_decode(src, lens, im, k, max_len)
            else:
                hyp = model.beamsearch_decode(src, lens, k, max_len)
            dec[str(k)] = [[int(t) for t in h] for h in hyp]
    meta["decode"] = dec
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s loss=%.6f mt=%.6f vse=%.6f |g|=%.4f  %d KB" % (
        name, float(loss), float(loss_mt), float(torch.as_tensor(loss_vse)), float(total),
        os.path.getsize(path) // 1024))


def run_losses():
    import machine_translation_vision.losses as Lo
    from machine_translation_vision.utils.utils import l2norm
    out = {}
    g = torch.Generator().manual_seed(7)
    for B in (1, 2, 5, 33):
        im = l2norm(torch.randn(B, 12, generator=g))
        s = l2norm(torch.randn(B, 12, generator=g))
        for m in (0.1, 1.0):
            out["B%d_m%g/im" % (B, m)] = np32(im)
            out["B%d_m%g/s" % (B, m)] = np32(s)
            out["B%d_m%g/pairwise" % (B, m)] = np32(Lo.PairwiseRankingLoss(margin=m)(im, s))
            out["B%d_m%g/imageretrieval" % (B, m)] = np32(Lo.ImageRetrievalRankingLoss(margin=m)(im, s))
    x = torch.randn(4, 9, generator=g)
    x[2] = 0
    out["l2norm/x"] = np32(x)
    out["l2norm/y"] = np32(l2norm(x))
    # retrieval metrics (utils/im_retrieval_eval.py) on correlated random embeddings
    import machine_translation_vision.utils.im_retrieval_eval as RE
    for N in (7, 100):
        im = l2norm(torch.randn(N, 16, generator=g))
        cap = l2norm(im + 0.6 * torch.randn(N, 16, generator=g))
        out["retr%d/im" % N] = np32(im)
        out["retr%d/cap" % N] = np32(cap)
        out["retr%d/t2i" % N] = np.array(RE.t2i(im, cap), dtype=np.float64)
        out["retr%d/i2t" % N] = np.array(RE.i2t(im, cap), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("losses.npz written")


def run_sampler():
    """Batches produced by the reference's BucketBatchSampler (samplers/bucket.py) under a fixed numpy seed."""
    from machine_translation_vision.samplers.bucket import BucketBatchSampler
    rs = np.random.RandomState(3)
    lengths = rs.randint(1, 13, size=203)
    out = {"lengths": lengths}
    for bs in (16, 64):
        smp = BucketBatchSampler(list(lengths), bs)
        np.random.seed(5)
        batches = [np.asarray(b) for b in smp]
        out["bs%d/flat" % bs] = np.concatenate(batches)
        out["bs%d/sizes" % bs] = np.array([len(b) for b in batches])
        out["bs%d/n_batches" % bs] = np.array([len(smp)])
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
    print("sampler.npz written")


def run_batches():
    """Batches produced by the reference's own generator (preprocessing.data_generator_tl_mtv, :308-384) for a seeded toy
    corpus.  preprocessing.py imports nltk at module level (:9) only for sent_tokenize, which the generator never calls:
    a stub module stands in for it (4th shim; nltk is not installed here)."""
    import types
    stub = types.ModuleType("nltk")
    tok = types.ModuleType("nltk.tokenize")
    tok.sent_tokenize = lambda s: [s]
    stub.tokenize = tok
    sys.modules.setdefault("nltk", stub)
    sys.modules.setdefault("nltk.tokenize", tok)
    import preprocessing as PP
    rs = np.random.RandomState(0)
    N = 150
    pairs = [[[int(t) for t in rs.randint(4, 50, size=rs.randint(1, 9))] + [3],
              [int(t) for t in rs.randint(4, 60, size=rs.randint(1, 7))] + [3]] for _ in range(N)]
    feats = rs.rand(N, 24).astype(np.float32)
    out = {"feats": feats, "x_len": np.array([len(p[0]) for p in pairs]), "y_len": np.array([len(p[1]) for p in pairs])}
    lx, ly = out["x_len"].max(), out["y_len"].max()
    X = np.zeros((N, lx), dtype=np.int64)
    Y = np.zeros((N, ly), dtype=np.int64)
    for i, (a, b) in enumerate(pairs):
        X[i, :len(a)] = a
        Y[i, :len(b)] = b
    out["x"], out["y"] = X, Y
    for bs in (16, 5):
        np.random.seed(11)
        nb = 0
        for k, (bx, by, bim, xl, yl) in enumerate(PP.data_generator_tl_mtv(pairs, feats, bs)):
            out["bs%d/%d/x" % (bs, k)] = bx.detach().cpu().numpy()
            out["bs%d/%d/y" % (bs, k)] = by.detach().cpu().numpy()
            out["bs%d/%d/im" % (bs, k)] = bim.detach().cpu().numpy()
            out["bs%d/%d/xl" % (bs, k)] = np.array(xl)
            out["bs%d/%d/yl" % (bs, k)] = np.array(yl)
            nb += 1
        out["bs%d/n" % bs] = np.array([nb])
    np.savez_compressed(os.path.join(OUT, "batches.npz"), **out)
    print("batches.npz written")


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    install_shims()
    random.seed(0)
    tiny = (50, 60, 96, 16, 24, 20, 5, 7, 6)      # Vs,Vt,I,E,H,S,B,Ts,Tt
    mid = (120, 130, 256, 32, 64, 48, 16, 12, 12)
    os.makedirs(OUT, exist_ok=True)
    run_case("mm_dot_tied_s0_f32", "mm", 0, tiny, attn="dot", tied=True)
    run_case("mm_dot_tied_s0_f64", "mm", 0, tiny, attn="dot", tied=True, dtype=torch.float64)
    run_case("mm_mlp_untied_s1_f32", "mm", 1, tiny, attn="mlp", tied=False, vse_loss="imageretrieval")
    run_case("mm_mlp_untied_s1_f64", "mm", 1, tiny, attn="mlp", tied=False, vse_loss="imageretrieval",
             dtype=torch.float64)
    run_case("text_tied_s0_f32", "text", 0, tiny, tied=True)
    run_case("text_untied_s1_f64", "text", 1, tiny, tied=False, dtype=torch.float64)
    run_case("mm_dot_tied_mid_f32", "mm", 2, mid, attn="dot", tied=True, store_adam=False, beams=(12,), max_len=20)
    run_case("mm_dot_full_len_f32", "mm", 3, (40, 44, 64, 16, 32, 16, 4, 6, 6), attn="dot", tied=True,
             ragged=False, beams=(3,), max_len=8)
    run_losses()
    run_sampler()
    run_batches()


if __name__ == "__main__":
    main()
