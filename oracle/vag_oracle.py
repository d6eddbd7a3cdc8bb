"""CPU oracle for the VAG-NMT hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch-CPU code, the arithmetic of the
reference's per-step training path and its beam-search decode.
It is the *checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``vag-nmt_amd/`` imports it, and the product path has no CPU fallback.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference package
from ``/root/reference`` (in the build container only), runs it on seeded
inputs and stores inputs/outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against those
vectors (<=1e-6 fp32, <=1e-12 fp64).

All ``file:line`` citations are relative to the reference checkout.  The
arithmetic below follows the reference's op ORDER (per-step ``attn_e`` GEMM,
per-step output head, both ranking-loss terms) so that timing it gives the
"reference CPU path" number; ``hoist=True`` switches to the once-per-batch
``attn_e`` projection, which is the same mathematics.

Parameters are passed as a dict ``P`` keyed by the reference's
``named_parameters()`` names (``encoder.gru.weight_ih_l0`` ...), so reference
``state_dict``s load directly.
"""
import math

import torch
import torch.nn.functional as F

SOS_token = 2  # models/NMT_AttentionImagine_Seq2Seq_Beam_V11.py:16
EOS_token = 3  # models/NMT_AttentionImagine_Seq2Seq_Beam_V11.py:17


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def l2norm(x, eps=1e-12):
    """Row-wise x / max(||x||_2, eps).  utils/utils.py:6-10."""
    n = x.pow(2).sum(1).sqrt().clamp(min=eps)
    return x / n.unsqueeze(1)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One GRU step, torch gate order (r, z, n).

    Third-party arithmetic: torch ``nn.GRU`` (pinned torch==0.4.1,
    requirements.txt:150), called at layers/Encoder.py:34,58 and
    layers/NMT_Decoder.py:83,86,121,129.  Published cell equations:
      r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
      n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
      h' = (1 - z) * n + z * h
    """
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def pairwise_ranking_loss(im, s, margin):
    """losses/PairwiseRankingLoss.py:9-24 (sum of both hinge directions,
    diagonal zeroed)."""
    scores = im @ s.t()
    d = scores.diag()
    cost_s = (margin - d).unsqueeze(0).expand_as(scores) + scores      # :16  m - d[j] + S[i,j]
    cost_im = (margin - d).unsqueeze(1).expand_as(scores) + scores     # :18  m - d[i] + S[i,j]
    cost_s = cost_s.clamp(min=0)
    cost_im = cost_im.clamp(min=0)
    off = 1.0 - torch.eye(scores.shape[0], dtype=scores.dtype)
    return (cost_s * off).sum() + (cost_im * off).sum()


def image_retrieval_ranking_loss(im, s, margin):
    """losses/ImageRetrievalRankingLoss.py:9-21 (cost_s term only)."""
    scores = im @ s.t()
    d = scores.diag()
    cost_s = ((margin - d).unsqueeze(0).expand_as(scores) + scores).clamp(min=0)
    off = 1.0 - torch.eye(scores.shape[0], dtype=scores.dtype)
    return (cost_s * off).sum()


# --------------------------------------------------------------------------
# encoder  (layers/Encoder.py:36-66)
# --------------------------------------------------------------------------
def encoder_forward(P, src, lengths, emb_mask=None, ctx_mask_drop=None, prefix="encoder."):
    """bi-GRU over variable-length rows.

    src: int64 (B,Ts) padded with 0; lengths: list[int] (descending, Encoder.py:55).
    emb_mask / ctx_mask_drop: optional dropout multipliers already scaled by
    1/(1-p), shapes (Ts,B,E) / (Ts,B,2H)  (Encoder.py:51-52, :63-64).
    Returns (enc (Ts,B,2H), mask float (Ts,B)) like the reference.
    """
    B, Ts = src.shape
    emb = P[prefix + "embedding.weight"]
    H = P[prefix + "gru.weight_hh_l0"].shape[1]
    mask = (src != 0).t().to(emb.dtype)                      # Encoder.py:47
    x = F.embedding(src, emb, padding_idx=0).transpose(0, 1)  # (Ts,B,E)  Encoder.py:22,50 (pad row: no grad)
    if emb_mask is not None:
        x = x * emb_mask
    Tmax = max(lengths)
    lens = torch.as_tensor(lengths)
    outs_f, outs_b = [], [None] * Tmax
    # forward direction, t = 0..len-1 per row; rows past their length emit 0 (pad_packed, Encoder.py:60)
    h = x.new_zeros(B, H)
    for t in range(Tmax):
        hn = gru_cell(x[t], h, P[prefix + "gru.weight_ih_l0"], P[prefix + "gru.weight_hh_l0"],
                      P[prefix + "gru.bias_ih_l0"], P[prefix + "gru.bias_hh_l0"])
        act = (t < lens).to(x.dtype).unsqueeze(1)
        h = act * hn + (1 - act) * h
        outs_f.append(act * hn)
    # reverse direction: each row runs t = len-1..0 starting from h=0
    h = x.new_zeros(B, H)
    for t in range(Tmax - 1, -1, -1):
        hn = gru_cell(x[t], h, P[prefix + "gru.weight_ih_l0_reverse"], P[prefix + "gru.weight_hh_l0_reverse"],
                      P[prefix + "gru.bias_ih_l0_reverse"], P[prefix + "gru.bias_hh_l0_reverse"])
        act = (t < lens).to(x.dtype).unsqueeze(1)
        h = act * hn + (1 - act) * h
        outs_b[t] = act * hn
    enc = torch.cat([torch.stack(outs_f, 0), torch.stack(outs_b, 0)], dim=2)   # (Tmax,B,2H)
    if Tmax < Ts:   # pad_packed_sequence returns max(lengths) rows; mask keeps Ts rows
        mask = mask[:Tmax]
    if ctx_mask_drop is not None:
        enc = enc * ctx_mask_drop
    return enc, mask


# --------------------------------------------------------------------------
# decoder  (layers/NMT_Decoder.py)
# --------------------------------------------------------------------------
def bahdanau_attn(P, h1, enc, mask, pe=None, prefix="decoder."):
    """alpha[b,s] = softmax_s( v . tanh(W_h h1[b] + W_e enc[s,b]) ), pads -> -inf.
    layers/NMT_Decoder.py:27-51.  enc (Ts,B,C); returns (B,Ts).
    ``pe`` = precomputed W_e enc (hoisted form); None = recompute (reference order, :47)."""
    W_h = P[prefix + "attn.attn_h.weight"]
    W_e = P[prefix + "attn.attn_e.weight"]
    v = P[prefix + "attn.v"]
    encb = enc.transpose(0, 1)                                # (B,Ts,C)
    if pe is None:
        pe = encb @ W_e.t()
    q = h1 @ W_h.t()                                          # (B,C)
    e = torch.tanh(pe + q.unsqueeze(1)) @ v                   # (B,Ts)
    if mask is not None:
        e = e.masked_fill(mask.t() == 0, float("-inf"))       # :41-43
    return torch.softmax(e, dim=1)                            # :44


def decoder_step(P, tok, h, enc, mask, out_drop=None, pe=None, prefix="decoder.", tied=None):
    """One cGRU step.  layers/NMT_Decoder.py:109-145.
    tok int64 (B,), h (B,H), enc (Ts,B,C) -> logp (B,V), h2 (B,H), aux dict."""
    emb_w = P[prefix + "embedding.weight"]
    e_t = F.embedding(tok.view(-1), emb_w, padding_idx=0)                          # :78,:118 (pad row: no grad)
    h1 = gru_cell(e_t, h, P[prefix + "gru_1.weight_ih_l0"], P[prefix + "gru_1.weight_hh_l0"],
                  P[prefix + "gru_1.bias_ih_l0"], P[prefix + "gru_1.bias_hh_l0"])  # :121
    alpha = bahdanau_attn(P, h1, enc, mask, pe=pe, prefix=prefix)                  # :124
    c = torch.bmm(alpha.unsqueeze(1), enc.transpose(0, 1)).squeeze(1)              # :126
    cp = c @ P[prefix + "context2hid.weight"].t()                                  # :127
    h2 = gru_cell(cp, h1, P[prefix + "gru_2.weight_ih_l0"], P[prefix + "gru_2.weight_hh_l0"],
                  P[prefix + "gru_2.bias_ih_l0"], P[prefix + "gru_2.bias_hh_l0"])  # :129
    t_ = torch.tanh(h2 @ P[prefix + "W1.weight"].t() + P[prefix + "W1.bias"]
                    + e_t @ P[prefix + "W3.weight"].t() + P[prefix + "W3.bias"]
                    + c @ P[prefix + "W2.weight"].t() + P[prefix + "W2.bias"])      # :137
    if out_drop is not None:
        t_ = t_ * out_drop                                                         # :140-141
    w_out = P.get(prefix + "out.weight", emb_w)               # tied: out.weight is embedding.weight (:105-106)
    logits = t_ @ w_out.t() + P[prefix + "out.bias"]
    logp = torch.log_softmax(logits, dim=-1)                                       # :143
    return logp, h2, dict(h1=h1, alpha=alpha, c=c, t=t_, logits=logits)


# --------------------------------------------------------------------------
# visual grounding  (layers/VSE_Imagine_Enc.py)
# --------------------------------------------------------------------------
def imagine_attn(P, im_emb, enc, mask, method="dot", prefix="vse_imagine.imagine_attn."):
    """layers/VSE_Imagine_Enc.py:29-79.  Returns alpha (B,Ts)."""
    encb = enc.transpose(0, 1)                                        # (B,Ts,C)
    ctx_ = encb @ P[prefix + "ctx2ctx.weight"].t()                    # :57 / :75
    im_ = im_emb @ P[prefix + "emb2ctx.weight"].t()                   # :58 / :76  (B,C)
    if method == "dot":
        e = torch.bmm(im_.unsqueeze(1), ctx_.transpose(1, 2)).squeeze(1)   # :64
    else:
        e = (torch.tanh(ctx_ + im_.unsqueeze(1)) @ P[prefix + "mlp.weight"].t()).squeeze(2)   # :78
    if mask is not None:
        e = e.masked_fill(mask.t() == 0, float("-inf"))               # :42-44
    return torch.softmax(e, dim=-1)                                   # :46


def vse_forward(P, im, enc, mask, method="dot", activation=True, prefix="vse_imagine."):
    """layers/VSE_Imagine_Enc.py:110-152.  Returns im_emb (B,S), txt_emb (B,S), alpha (B,Ts), ctx (B,C)."""
    im_emb = im @ P[prefix + "im_embedding.weight"].t() + P[prefix + "im_embedding.bias"]    # :123
    if activation:
        im_emb = torch.tanh(im_emb)                                                           # :125-126
    im_emb = l2norm(im_emb)                                                                   # :132
    alpha = imagine_attn(P, im_emb, enc, mask, method, prefix + "imagine_attn.")              # :135
    ctx = torch.bmm(alpha.unsqueeze(1), enc.transpose(0, 1)).squeeze(1)                        # :137
    txt = ctx @ P[prefix + "text_embedding.weight"].t() + P[prefix + "text_embedding.bias"]   # :138
    if activation:
        txt = torch.tanh(txt)
    txt = l2norm(txt)                                                                         # :145
    return im_emb, txt, alpha, ctx


def decoder_init(P, enc, mask, ctx=None, init_split=0.5):
    """h0 = tanh(W_ini (split*ctx + (1-split)*meanpool(enc)) + b).
    models/...V11.py:118 (multimodal) / models/NMT_Seq2Seq_Beam_V2.py:85 (ctx=None)."""
    mean = enc.sum(0) / mask.sum(0).unsqueeze(1)
    x = mean if ctx is None else init_split * ctx + (1 - init_split) * mean
    return torch.tanh(x @ P["decoderini.weight"].t() + P["decoderini.bias"])


# --------------------------------------------------------------------------
# full model forward  (models/...V11.py:82-168, models/NMT_Seq2Seq_Beam_V2.py:58-113)
# --------------------------------------------------------------------------
def model_forward(P, src, lengths, tgt, im=None, *, teacher=True, vocab_weight=None,
                  loss_w=0.99, init_split=0.5, attn="dot", activation=True,
                  vse_loss="pairwise", margin=0.1, masks=None, hoist=False, keep=False, ckpt=False):
    """Training/validation forward.  ``im is None`` = text-only model (a12).

    masks: optional dict of dropout multipliers {'emb' (Ts,B,E), 'ctx' (Ts,B,2H),
    'out' (Tt,B,E)} (already scaled by 1/(1-p)); None = eval mode.
    ckpt: recompute each decoder step in backward instead of keeping its activations (same values; bounds
    the memory of the configs[4]-size parity test: a step's (Ts,B,2H) attention tensors are 168 MB there).
    Returns dict(loss, loss_mt, loss_vse, + intermediates if keep).
    """
    masks = masks or {}
    B, Tt = tgt.shape
    tgt_mask = (tgt != 0).to(P["decoderini.weight"].dtype)
    enc, mask = encoder_forward(P, src, lengths, masks.get("emb"), masks.get("ctx"))
    out = {}
    if im is not None:
        im_emb, txt_emb, alpha_v, ctx = vse_forward(P, im, enc, mask, attn, activation)
        if vse_loss == "pairwise":
            loss_vse = pairwise_ranking_loss(im_emb, txt_emb, margin)
        elif vse_loss == "imageretrieval":
            loss_vse = image_retrieval_ranking_loss(im_emb, txt_emb, margin)
        else:
            loss_vse = 0.0
        h = decoder_init(P, enc, mask, ctx, init_split)
        if keep:
            out.update(im_emb=im_emb, txt_emb=txt_emb, alpha_vse=alpha_v, ctx_vse=ctx)
    else:
        loss_vse = None
        h = decoder_init(P, enc, mask)
    if keep:
        out.update(enc=enc, mask=mask, h0=h)
    pe = None
    if hoist:
        pe = enc.transpose(0, 1) @ P["decoder.attn.attn_e.weight"].t()
    tok = torch.full((B,), SOS_token, dtype=torch.long)
    if vocab_weight is None:
        vocab_weight = torch.ones(P["decoder.out.bias"].shape[0], dtype=h.dtype)
        vocab_weight[0] = 0            # nmt_multimodal_beam_DE.py:286-291
    L = h.new_zeros(B)
    steps = []
    for di in range(Tt):
        od = masks["out"][di] if "out" in masks else None
        if ckpt and not keep:
            from torch.utils.checkpoint import checkpoint
            logp, h = checkpoint(lambda t_, h_, od_=od: decoder_step(P, t_, h_, enc, mask, od_, pe=pe)[:2], tok, h,
                                 use_reentrant=False)
            aux = {}
        else:
            logp, h, aux = decoder_step(P, tok, h, enc, mask, od, pe=pe)
        tg = tgt[:, di]
        L = L + (-vocab_weight[tg] * logp.gather(1, tg.unsqueeze(1)).squeeze(1))   # NLLLoss(weight, reduce=False)
        if teacher:
            tok = tg                                                               # V11.py:146
        else:
            tok = logp.detach().argmax(dim=1)                                      # V11.py:157
        if keep:
            aux.update(logp=logp, h2=h, tok_next=tok)
            steps.append(aux)
    loss_mt = (L / tgt_mask.sum(-1)).mean()                                        # V11.py:164
    if im is not None:
        loss = loss_w * loss_mt + (1 - loss_w) * loss_vse                          # V11.py:166
    else:
        loss = loss_mt
    out.update(loss=loss, loss_mt=loss_mt, loss_vse=loss_vse)
    if keep:
        out["steps"] = steps
    return out


# --------------------------------------------------------------------------
# decode  (models/...V11.py:179-337)
# --------------------------------------------------------------------------
def _decode_prologue(P, src, lengths, im, init_split, attn, activation):
    enc, mask = encoder_forward(P, src, lengths)
    if im is not None:
        _, _, _, ctx = vse_forward(P, im, enc, mask, attn, activation)
        h = decoder_init(P, enc, mask, ctx, init_split)
    else:
        h = decoder_init(P, enc, mask)
    return enc, mask, h


def _cut_eos(rows):
    out = []
    for r in rows:
        cur = []
        for t in r:
            if int(t) == EOS_token:
                break
            cur.append(int(t))
        out.append(cur)
    return out


def greedy_decode(P, src, lengths, im=None, max_length=80, init_split=0.5, attn="dot", activation=True):
    """beam_size==1 branch, V11.py:207-226: argmax for exactly max_length steps, cut at EOS."""
    with torch.no_grad():
        enc, mask, h = _decode_prologue(P, src, lengths, im, init_split, attn, activation)
        B = src.shape[0]
        tok = torch.full((B,), SOS_token, dtype=torch.long)
        toks = []
        for _ in range(max_length):
            logp, h, _ = decoder_step(P, tok, h, enc, mask)
            tok = logp.argmax(dim=1)
            toks.append(tok)
        return _cut_eos(torch.stack(toks, 1).tolist())


def beam_search(P, src, lengths, im=None, beam_size=12, max_length=80, init_split=0.5,
                attn="dot", activation=True, return_scores=False):
    """Batched beam search, V11.py:233-337 (avoid_double=True, avoid_unk=False).

    Returns list[B] of token lists (EOS cut); with return_scores also the
    length-normalised score of the chosen hypothesis per sentence."""
    with torch.no_grad():
        enc, mask, h = _decode_prologue(P, src, lengths, im, init_split, attn, activation)
        B = src.shape[0]
        k = beam_size
        V = P["decoder.out.bias"].shape[0]
        nk = torch.arange(B * k)
        pdxs_mask = (nk // k) * k                                         # :242
        tile = nk // k                                                    # :245
        beam = torch.zeros(max_length, B, k, dtype=torch.long)            # :248
        enc_di = enc[:, tile, :]                                          # :253
        mask_di = mask[:, tile]
        inf = -1e5                                                        # :257
        tok = torch.full((B,), SOS_token, dtype=torch.long)
        nll = None
        for di in range(max_length):
            if di == 0:
                logp, h, _ = decoder_step(P, tok, h, enc, mask)            # :260
                nll, topk = logp.topk(k, dim=1)                           # :261
                beam[0] = topk
            else:
                cur = beam[di - 1].reshape(-1)
                fini = (cur == EOS_token).nonzero()[:, 0]                 # :266
                if fini.numel() == B * k:
                    break                                                 # :268-269
                h = h[tile]                                               # :273
                logp, h, _ = decoder_step(P, cur, h, enc_di, mask_di)      # :275
                logp = logp.clone()
                logp.view(-1)[cur + nk * V] = inf                         # :279-280
                if fini.numel() > 0:
                    logp[fini] = inf                                      # :293
                    logp.view(-1)[fini * V + EOS_token] = 0               # :294
                tot = (nll.unsqueeze(2) + logp.view(B, k, V)).view(B, -1)  # :297
                nll, idxs = tot.topk(k, dim=1)                            # :300
                pdxs = idxs // V                                          # :303
                beam[di] = idxs % V                                       # :306
                beam[:di] = beam[:di].gather(2, pdxs.unsqueeze(0).expand(di, B, k))   # :309
                tile = pdxs.view(-1) + pdxs_mask                          # :313
        beam[max_length - 1] = EOS_token                                  # :315
        lens = (beam.permute(2, 1, 0) > 3).sum(-1).t().to(nll.dtype).clamp(min=1)     # :318
        nll = nll / lens                                                  # :321
        best_score, top = nll.max(dim=1)                                  # :322
        hyps = beam[:, torch.arange(B), top].t()                          # :324
        res = _cut_eos(hyps.tolist())
        if return_scores:
            return res, best_score
        return res


def embed_sent_im(P, src, lengths, im, attn="dot", activation=True):
    """V11.py:370-397 -> VSE_Imagine_Enc.get_emb_vec (:154-172)."""
    with torch.no_grad():
        enc, mask = encoder_forward(P, src, lengths)
        im_emb, txt_emb, _, _ = vse_forward(P, im, enc, mask, attn, activation)
        return im_emb, txt_emb


# --------------------------------------------------------------------------
# optimiser step  (train.py:36-51 + nmt_multimodal_beam_DE.py:303-332)
# --------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train.py:46): global L2 norm over all grads,
    scale by max_norm/(norm+1e-6) when that is < 1.  Returns (total_norm, scaled grads)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    coef = max_norm / (total + 1e-6)
    coef = torch.clamp(coef, max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def adam_step(P, grads, state, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
              lr_scale=None):
    """torch.optim.Adam with L2 weight decay on parameters whose NAME lacks 'bias'
    (nmt_multimodal_beam_DE.py:303-332).  Published update (torch 0.4.1 Adam):
      g += wd*p ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
      p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t))   [== torch form below]
    state: dict name -> (m, v), plus state['step'].  Returns new params dict.
    lr_scale: optional dict name->multiplier (vse_separate groups, :316-329)."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    new = {}
    for name, p in P.items():
        g = grads[name]
        if "bias" not in name:
            g = g + weight_decay * p
        m, v = state.get(name, (torch.zeros_like(p), torch.zeros_like(p)))
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        state[name] = (m, v)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = v.sqrt() / math.sqrt(bc2) + eps
        step = lr * (lr_scale.get(name, 1.0) if lr_scale else 1.0) / bc1
        new[name] = p - step * (m / denom)
    return new


def train_step(P, src, lengths, tgt, im=None, *, clip=1.0, lr=4e-4, weight_decay=1e-5,
               state=None, **fw):
    """One full optimiser step (fwd + bwd + clip + Adam), train.py:36-51.
    P: dict of leaf tensors.  Returns (out dict, grads, total_norm, new params, state)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = model_forward(leaves, src, lengths, tgt, im, **fw)
    out["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    # padding_idx=0 rows of both embeddings receive no gradient (nn.Embedding(padding_idx=0),
    # Encoder.py:22, NMT_Decoder.py:78).  With a tied output layer the row still gets the
    # output-projection gradient, exactly as in torch.
    total, cg = clip_grad_norm(grads, clip)
    state = {} if state is None else state
    with torch.no_grad():
        newP = adam_step({k: v.detach() for k, v in leaves.items()}, cg, state, lr=lr, weight_decay=weight_decay)
    return out, grads, total, newP, state


# --------------------------------------------------------------------------
# retrieval evaluation  (utils/im_retrieval_eval.py:4-57)
# --------------------------------------------------------------------------
def retrieval_ranks(queries, keys):
    """rank[i] = position of key i in the descending sort of queries[i] . keys^T  (:15-24)."""
    d = queries @ keys.t()
    inds = torch.sort(d, dim=1, descending=True, stable=True)[1]
    return (inds == torch.arange(d.shape[0]).unsqueeze(1)).nonzero()[:, 1]


def retrieval_metrics(ranks):
    """(R@1, R@5, R@10, median rank)  (:25-30)."""
    import numpy as np
    r = ranks.numpy().astype("float64")
    return (100.0 * (r < 1).sum() / len(r), 100.0 * (r < 5).sum() / len(r), 100.0 * (r < 10).sum() / len(r),
            float(np.floor(np.median(r)) + 1))


def t2i(images, captions):
    return retrieval_metrics(retrieval_ranks(captions, images))


def i2t(images, captions):
    return retrieval_metrics(retrieval_ranks(images, captions))


# --------------------------------------------------------------------------
# batch assembly  (preprocessing.py:308-384 data_generator_tl_mtv).  Pinned: tests/golden/batches.npz holds the batches the
# reference's own generator produced for a seeded toy corpus (oracle/make_golden.py:run_batches, nltk stubbed)
# --------------------------------------------------------------------------
def assemble_batch(data_pairs, data_im, bidx):
    import numpy as np
    xs = [data_pairs[i][0] for i in bidx]
    ys = [data_pairs[i][1] for i in bidx]
    x_length = max(len(x) for x in xs)                                   # :346
    y_length = max(len(y) for y in ys)
    x_lens = [len(x) for x in xs]
    order = [i for i in reversed(list(np.argsort(x_lens)))]              # :354-356
    pad = lambda seq, n: list(seq) + [0] * (n - len(seq))                 # pad_seq
    bx = torch.tensor([pad(xs[i], x_length) for i in order], dtype=torch.long)
    by = torch.tensor([pad(ys[i], y_length) for i in order], dtype=torch.long)
    bim = torch.from_numpy(np.asarray(data_im)[np.asarray(bidx)][order]).float() if data_im is not None else None   # :370-372
    return bx, by, bim, list(reversed(sorted(x_lens))), [len(ys[i]) for i in order]              # :384
