"""torch.autograd glue around the C ABI of libvagnmt.so.

Each Function is ONE forward and ONE backward call into the HIP library (include/vag_nmt.h); torch only
owns the memory and strings the coarse operators together.  Nothing here computes on the CPU or with
torch operators: without the library (or with CPU tensors) every op raises.

Gradient routing: a parameter may carry ``_vag_grad`` (a view into the trainer's flat gradient buffer).
Backward then accumulates straight into that view and returns None for the parameter, so no per-step
gradient tensors are allocated and autograd does no extra accumulation passes.
"""
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import DecW, GruW, HeadW, call, gru_w, ptr, stream

I64 = torch.int64


# Guard pair {void flag, give-up count} of the step driver the operators currently run for (trainer._AutogradBackend): a PROCESS-wide
# Python value, because autograd runs backward() on its own device worker thread -- a thread-local set by the driver's thread
# (vag_set_operator_guard alone) never reaches the backward recurrences, whose give-ups then landed in the process-wide pair that
# the driver's optimiser does not read.  The recurrence operators below carry it into each ABI call themselves.
_OPERATOR_GUARD = None


def set_operator_guard(address):
    """Device address of the guard pair (or None) for the persistent recurrence launches of the operators, on every thread."""
    global _OPERATOR_GUARD
    _OPERATOR_GUARD = address


def _recurrence_call(name, *args):
    """An ABI call that may launch a persistent recurrence kernel: runs with the current driver's guard pair on THIS thread."""
    g = _OPERATOR_GUARD
    if g is None:
        return call(name, *args)
    call("vag_set_operator_guard", g)
    try:
        return call(name, *args)
    finally:
        call("vag_set_operator_guard", None)


def _f32(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _zeros(*shape, like):
    return torch.zeros(*shape, dtype=torch.float32, device=like.device)


def _grad_views(params):
    """Forward time: remember each parameter's flat-gradient view (or None)."""
    return [getattr(p, "_vag_grad", None) for p in params]


def _grad_targets(views, params):
    """Backward time: (buffer to accumulate into, whether autograd should receive it) per parameter."""
    return [(v, False) if v is not None else (torch.zeros_like(p), True) for v, p in zip(views, params)]


def _ret(targets):
    return tuple(t if give else None for t, give in targets)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------
class BiGRUEncode(Function):
    """layers/Encoder.py:36-66.  Returns enc (B,Ts,2H) batch-major and mask (B,Ts)."""

    @staticmethod
    def forward(ctx, src, lengths, emb, wf_ih, wf_hh, bf_ih, bf_hh, wb_ih, wb_hh, bb_ih, bb_hh, p_emb, p_ctx, rng):
        B, Ts = src.shape
        E, H = emb.shape[1], wf_hh.shape[1]
        enc = _f32(B, Ts, 2 * H, like=emb)
        mask = _f32(B, Ts, like=emb)
        ws = _f32(L.lib().vag_bigru_ws_floats(B, Ts, E, H), like=emb)
        src = _c(src)
        _recurrence_call("vag_bigru_seq_fwd", ptr(src, I64), ptr(lengths, torch.int32), ptr(emb),
             gru_w(wf_ih, wf_hh, bf_ih, bf_hh), gru_w(wb_ih, wb_hh, bb_ih, bb_hh), p_emb, p_ctx,
             ptr(rng, torch.int64) if rng is not None else None, B, Ts, E, H, ptr(enc), ptr(mask), ptr(ws), stream())
        params = (emb, wf_ih, wf_hh, bf_ih, bf_hh, wb_ih, wb_hh, bb_ih, bb_hh)
        ctx.save_for_backward(src, lengths, ws, *params)
        ctx.gviews = _grad_views(params)
        ctx.cfg = (B, Ts, E, H, p_emb, p_ctx, rng)
        ctx.mark_non_differentiable(mask)
        return enc, mask

    @staticmethod
    def backward(ctx, d_enc, _d_mask):
        src, lengths, ws, emb, wf_ih, wf_hh, bf_ih, bf_hh, wb_ih, wb_hh, bb_ih, bb_hh = ctx.saved_tensors
        B, Ts, E, H, p_emb, p_ctx, rng = ctx.cfg
        d_enc = _c(d_enc)
        t = _grad_targets(ctx.gviews, ctx.saved_tensors[3:])
        g = [x[0] for x in t]
        _recurrence_call("vag_bigru_seq_bwd", ptr(src, I64), ptr(lengths, torch.int32), gru_w(wf_ih, wf_hh, bf_ih, bf_hh),
             gru_w(wb_ih, wb_hh, bb_ih, bb_hh), p_emb, p_ctx, ptr(rng, torch.int64) if rng is not None else None,
             B, Ts, E, H, ptr(d_enc), ptr(ws), ptr(g[0]), gru_w(g[1], g[2], g[3], g[4]), gru_w(g[5], g[6], g[7], g[8]),
             stream())
        return (None, None) + _ret(t) + (None, None, None)


# ----------------------------------------------------------------------------------------------------
class LinearAct(Function):
    """nn.Linear (+ optional tanh) on the fp32 MFMA GEMM kernels."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x2 = _c(x).view(-1, x.shape[-1])
        M, K = x2.shape
        N = W.shape[0]
        y = _f32(M, N, like=x)
        call("vag_linear_fwd", M, N, K, ptr(x2), ptr(W), ptr(b) if b is not None else None, int(act), ptr(y), stream())
        ctx.save_for_backward(x2, W, y)
        ctx.gviews = _grad_views((W,) + ((b,) if b is not None else ()))
        ctx.bshape = b.shape if b is not None else None
        ctx.act = int(act)
        ctx.has_b = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        dy = _c(dy).view(M, N).clone()
        dx = _f32(M, K, like=x2) if ctx.needs_input_grad[0] else None
        t = [(ctx.gviews[0], False) if ctx.gviews[0] is not None else (torch.zeros_like(W), True)]
        if ctx.has_b:
            t.append((ctx.gviews[1], False) if ctx.gviews[1] is not None else (_zeros(*ctx.bshape, like=W), True))
        call("vag_linear_bwd", M, N, K, ptr(x2), ptr(W), ptr(y), ptr(dy), ctx.act, ptr(dx) if dx is not None else None,
             0, ptr(t[0][0]), ptr(t[1][0]) if ctx.has_b else None, stream())
        r = _ret(t)
        return (dx.view(ctx.xshape) if dx is not None else None, r[0], r[1] if ctx.has_b else None, None)


class Embedding(Function):
    """nn.Embedding(padding_idx=0) lookup."""

    @staticmethod
    def forward(ctx, idx, W):
        flat = _c(idx).view(-1)
        out = _f32(flat.numel(), W.shape[1], like=W)
        call("vag_embed_fwd", ptr(flat, I64), flat.numel(), ptr(W), W.shape[1], ptr(out), stream())
        ctx.save_for_backward(flat, W)
        ctx.gviews = _grad_views((W,))
        ctx.E = W.shape[1]
        return out.view(*idx.shape, W.shape[1])

    @staticmethod
    def backward(ctx, d_out):
        flat, W = ctx.saved_tensors
        d_out = _c(d_out).view(-1, ctx.E)
        t = _grad_targets(ctx.gviews, (W,))
        call("vag_embed_bwd", ptr(flat, I64), flat.numel(), ptr(d_out), ctx.E, ptr(t[0][0]), stream())
        return (None,) + _ret(t)


# ----------------------------------------------------------------------------------------------------
class KeysProj(Function):
    """pe = enc W_e^T  (attn_e, layers/NMT_Decoder.py:47), once per batch."""

    @staticmethod
    def forward(ctx, enc, W_e):
        B, Ts, Cc = enc.shape
        pe = _f32(B, Ts, Cc, like=enc)
        call("vag_attn_keys_proj", ptr(enc), ptr(W_e), B * Ts, Cc, ptr(pe), stream())
        ctx.save_for_backward(enc, W_e)
        ctx.gviews = _grad_views((W_e,))
        return pe

    @staticmethod
    def backward(ctx, d_pe):
        enc, W_e = ctx.saved_tensors
        B, Ts, Cc = enc.shape
        d_pe = _c(d_pe)
        d_enc = _f32(B, Ts, Cc, like=enc)
        t = _grad_targets(ctx.gviews, (W_e,))
        call("vag_attn_keys_proj_bwd", ptr(enc), ptr(W_e), ptr(d_pe), B * Ts, Cc, ptr(d_enc), 0, ptr(t[0][0]), stream())
        return (d_enc,) + _ret(t)


# ----------------------------------------------------------------------------------------------------
def _dec_w(emb, p):
    return DecW(ptr(emb), gru_w(p[0], p[1], p[2], p[3]), ptr(p[4]), ptr(p[5]), ptr(p[6]), gru_w(p[7], p[8], p[9], p[10]))


def _head_w(p):
    return HeadW(*[ptr(x) for x in p])


class _CGRUDecodeSeq(Function):
    """Whole-sequence cGRU + Bahdanau attention (layers/NMT_Decoder.py:109-131 inside the loop of
    models/...V11.py:138-160).  dec params order: gru1 (w_ih,w_hh,b_ih,b_hh), attn_h, attn_v, c2h, gru2 (4).
    head params (free-running only): w1,b1,w2,b2,w3,b3,out_w,out_b.
    Returns h2_all (Tt,B,H), c_all (Tt,B,C), e_all (Tt,B,E) [, tmid, logits when free_run]."""

    @staticmethod
    def forward(ctx, enc, pe, mask, h0, tok, emb, free_run, p_out, rng, V, ldl, ndec, *rest):
        dec, head = rest[:ndec], rest[ndec:]
        B, Ts, Cc = enc.shape
        H = Cc // 2
        E = emb.shape[1]
        Tt = tok.shape[0] - 1
        # h0 and the Tt hidden states share one buffer, so that backward sees the previous-state sequence
        # [h0, h2_0 .. h2_{Tt-2}] as ONE (Tt*B, H) operand of the W_hh1 gradient product
        hseq = _f32(Tt + 1, B, H, like=enc)
        hseq[0].copy_(h0)
        h0, h2 = hseq[0], hseq[1:]
        c = _f32(Tt, B, Cc, like=enc)
        e = _f32(Tt, B, E, like=enc)
        ws = _f32(L.lib().vag_cgru_ws_floats(B, Ts, Tt, E, H), like=enc)
        tmid = logits = None
        hw = None
        if free_run:
            tmid = _f32(Tt, B, E, like=enc)
            logits = _f32(Tt * B, ldl, like=enc)
            hw = C.byref(_head_w(head))
        if free_run and L.lib().vag_cgru_free_supported(B, Ts, Tt, E, H, V):
            # one launch for all steps (persist.hip, free-running form); same outputs and saved tensors
            tables = _f32(L.lib().vag_cgru_free_tables_floats(B, Ts, Tt, E, H, V), like=enc)
            _recurrence_call("vag_cgru_attn_decode_free_fwd", ptr(enc), ptr(pe), ptr(mask), ptr(h0), ptr(tok, I64), _dec_w(emb, dec),
                 B, Ts, Tt, E, H, V, ptr(h2), ptr(c), ptr(e), ptr(ws), hw, float(p_out),
                 ptr(rng, torch.int64) if rng is not None else None, ptr(tmid), ptr(logits), ldl, ptr(tables), stream())
        else:
            _recurrence_call("vag_cgru_attn_decode_seq_fwd", ptr(enc), ptr(pe), ptr(mask), ptr(h0), ptr(tok, I64), _dec_w(emb, dec),
                 B, Ts, Tt, E, H, V, ptr(h2), ptr(c), ptr(e), ptr(ws), int(bool(free_run)), hw, float(p_out),
                 ptr(rng, torch.int64) if rng is not None else None, ptr(tmid), ptr(logits), ldl, stream())
        ctx.save_for_backward(enc, pe, mask, h0, tok, ws, h2, c, e, emb, *dec)
        ctx.gviews = _grad_views((emb,) + tuple(dec))
        ctx.cfg = (B, Ts, Tt, E, H, V)
        ctx.nrest = len(rest)
        ctx.ndec = ndec
        if free_run:
            ctx.mark_non_differentiable(tmid, logits)
            return h2, c, e, tmid, logits
        return h2, c, e

    @staticmethod
    def backward(ctx, d_h2, d_c, d_e, *_unused):
        enc, pe, mask, h0, tok, ws, h2, c, e, emb = ctx.saved_tensors[:10]
        dec = ctx.saved_tensors[10:]
        B, Ts, Tt, E, H, V = ctx.cfg
        Cc = 2 * H
        d_h2 = _c(d_h2).clone() if d_h2 is not None else _zeros(Tt, B, H, like=enc)
        d_c = _c(d_c).clone() if d_c is not None else _zeros(Tt, B, Cc, like=enc)
        d_e = _c(d_e) if d_e is not None else None
        d_enc = _f32(B, Ts, Cc, like=enc)
        d_pe = _f32(B, Ts, Cc, like=enc)
        d_h0 = _f32(B, H, like=enc)
        scratch = _f32(L.lib().vag_cgru_bwd_scratch_floats(B, Ts, Tt, E, H), like=enc)
        t = _grad_targets(ctx.gviews, (emb,) + tuple(dec))
        g = [x[0] for x in t]
        gdec = DecW(ptr(g[0]), gru_w(g[1], g[2], g[3], g[4]), ptr(g[5]), ptr(g[6]), ptr(g[7]),
                    gru_w(g[8], g[9], g[10], g[11]))
        _recurrence_call("vag_cgru_attn_decode_seq_bwd", ptr(enc), ptr(pe), ptr(mask), ptr(h0), ptr(tok, I64), _dec_w(emb, dec),
             B, Ts, Tt, E, H, V, ptr(h2), ptr(c), ptr(e), ptr(d_h2), ptr(d_c), ptr(d_e), ptr(ws), ptr(d_enc), 0,
             ptr(d_pe), ptr(d_h0), gdec, ptr(scratch), stream())
        r = _ret(t)
        return (d_enc, d_pe, None, d_h0, None, r[0], None, None, None, None, None, None) + tuple(r[1:]) + \
            (None,) * (ctx.nrest - ctx.ndec)


def cgru_decode_seq(enc, pe, mask, h0, tok, emb, dec, free_run=False, head=None, p_out=0.0, rng=None, V=0, ldl=0):
    """autograd.Function inputs must be flat, so the parameter tuples are splatted."""
    return _CGRUDecodeSeq.apply(enc, pe, mask, h0, tok, emb, bool(free_run), p_out, rng, V, ldl, len(dec), *dec,
                                *(head or ()))


# ----------------------------------------------------------------------------------------------------
class HeadCE(Function):
    """Output head + weighted NLL + per-sentence normalisation -> loss_mt (scalar).
    layers/NMT_Decoder.py:137-143, models/...V11.py:140,164.  head = w1,b1,w2,b2,w3,b3,out_w,out_b."""

    @staticmethod
    def forward(ctx, h2, c, e, tgt, vw, p_out, rng, tmid, logits, ldl, *head):
        Tt, B, H = h2.shape
        E = e.shape[2]
        V = head[7].shape[0]
        R = Tt * B
        ready = logits is not None
        if not ready:
            tmid = _f32(Tt, B, E, like=h2)
            logits = _f32(R, ldl, like=h2)
        lse = _f32(R, like=h2)
        nll = _f32(R, like=h2)
        inv_cnt = _f32(B, like=h2)
        loss = _f32(1, like=h2)
        tgt = _c(tgt)
        call("vag_head_ce_seq_fwd", ptr(h2), ptr(c), ptr(e), _head_w(head), ptr(tgt, I64), ptr(vw), B, Tt, E, H, V, p_out,
             ptr(rng, torch.int64) if rng is not None else None, int(ready), ptr(tmid), ptr(logits), ldl, ptr(lse),
             ptr(nll), ptr(inv_cnt), ptr(loss), stream())
        ctx.save_for_backward(h2, c, e, tgt, vw, tmid, logits, lse, inv_cnt, *head)
        ctx.gviews = _grad_views(head)
        ctx.cfg = (B, Tt, E, H, V, p_out, rng, ldl)
        ctx.nll = nll
        return loss.view(())

    @staticmethod
    def backward(ctx, d_loss):
        h2, c, e, tgt, vw, tmid, logits, lse, inv_cnt = ctx.saved_tensors[:9]
        head = ctx.saved_tensors[9:]
        B, Tt, E, H, V, p_out, rng, ldl = ctx.cfg
        d_loss = _c(d_loss).view(1)
        d_h2 = _f32(Tt, B, H, like=h2)
        d_c = _f32(Tt, B, 2 * H, like=h2)
        d_e = _f32(Tt, B, E, like=h2)
        scratch = _f32(Tt * B * E, like=h2)
        t = _grad_targets(ctx.gviews, head)
        g = [x[0] for x in t]
        # tied embeddings: out_w and the decoder embedding are the same Parameter, so g[6] is its gradient buffer
        rp = ptr(rng, torch.int64) if rng is not None else None
        call("vag_head_ce_seq_bwd", ptr(h2), ptr(c), ptr(e), _head_w(head), ptr(tgt, I64), ptr(vw), B, Tt, E, H, V, p_out,
             rp, ptr(tmid), ptr(logits), ldl, ptr(lse), ptr(inv_cnt), ptr(d_loss), ptr(d_h2), ptr(d_c), ptr(d_e),
             HeadW(*[ptr(x) for x in g]), ptr(scratch), stream())
        return (d_h2, d_c, d_e, None, None, None, None, None, None, None) + _ret(t)


class HeadLogp(Function):
    """logp = log_softmax(out(dropout(tanh(W1 h2 + W2 c + W3 e + b))))  (layers/NMT_Decoder.py:137-143) for R rows,
    differentiable w.r.t. everything -- used by the per-step layer API and for criteria other than nn.NLLLoss."""

    @staticmethod
    def forward(ctx, h2, c, e, p_out, rng, *head):
        h2, c, e = _c(h2), _c(c), _c(e)
        R, H = h2.shape
        E = e.shape[1]
        V = head[7].shape[0]
        ldl = (V + 3) // 4 * 4
        tmid = _f32(R, E, like=h2)
        logp = _f32(R, ldl, like=h2)
        call("vag_head_logp_seq_fwd", ptr(h2), ptr(c), ptr(e), _head_w(head), R, E, H, V, float(p_out),
             ptr(rng, torch.int64) if rng is not None else None, ptr(tmid), ptr(logp), ldl, stream())
        ctx.save_for_backward(h2, c, e, tmid, logp, *head)
        ctx.gviews = _grad_views(head)
        ctx.cfg = (R, E, H, V, float(p_out), rng, ldl)
        return logp[:, :V]

    @staticmethod
    def backward(ctx, d_logp):
        h2, c, e, tmid, logp = ctx.saved_tensors[:5]
        head = ctx.saved_tensors[5:]
        R, E, H, V, p_out, rng, ldl = ctx.cfg
        d = _zeros(R, ldl, like=h2)
        d[:, :V].copy_(d_logp)
        d_h2 = _f32(R, H, like=h2)
        d_c = _f32(R, 2 * H, like=h2)
        d_e = _f32(R, E, like=h2)
        scratch = _f32(R * E, like=h2)
        t = _grad_targets(ctx.gviews, head)
        call("vag_head_logp_seq_bwd", ptr(h2), ptr(c), ptr(e), _head_w(head), R, E, H, V, p_out,
             ptr(rng, torch.int64) if rng is not None else None, ptr(tmid), ptr(logp), ptr(d), ldl, ptr(d_h2), ptr(d_c),
             ptr(d_e), HeadW(*[ptr(x[0]) for x in t]), ptr(scratch), stream())
        return (d_h2, d_c, d_e, None, None) + _ret(t)


# ----------------------------------------------------------------------------------------------------
class ImgProjL2(Function):
    """l2norm(act(x W^T + b))  (layers/VSE_Imagine_Enc.py:123-132 / :138-145, utils/utils.py:6-10)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x = _c(x)
        B, K = x.shape
        S = W.shape[0]
        y = _f32(B, S, like=x)
        nrm = _f32(B, like=x)
        out = _f32(B, S, like=x)
        call("vag_img_proj_l2_fwd", ptr(x), ptr(W), ptr(b), B, K, S, int(act), ptr(y), ptr(nrm), ptr(out), stream())
        ctx.save_for_backward(x, W, y, nrm, out, b)
        ctx.gviews = _grad_views((W, b))
        ctx.act = int(act)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, W, y, nrm, out, b = ctx.saved_tensors
        B, K = x.shape
        S = W.shape[0]
        d_out = _c(d_out).clone()
        dx = _f32(B, K, like=x) if ctx.needs_input_grad[0] else None
        t = _grad_targets(ctx.gviews, (W, b))
        call("vag_img_proj_l2_bwd", ptr(x), ptr(W), ptr(y), ptr(nrm), ptr(out), ptr(d_out), B, K, S, ctx.act,
             ptr(dx) if dx is not None else None, ptr(t[0][0]), ptr(t[1][0]), stream())
        return (dx,) + _ret(t) + (None,)


class ImagineAttnCtx(Function):
    """Image-conditioned attention weights and attended context (VSE_Imagine_Enc.py:29-79, :135-137)."""

    @staticmethod
    def forward(ctx, im_emb, enc, mask, W_cc, W_ec, mlp_w, method):
        B, Ts, Cc = enc.shape
        S = im_emb.shape[1]
        im_emb = _c(im_emb)
        alpha = _f32(B, Ts, like=enc)
        cvec = _f32(B, Cc, like=enc)
        ws = _f32(L.lib().vag_imagine_ws_floats(B, Ts, Cc, S, method), like=enc)
        mw = mlp_w.view(-1) if mlp_w is not None else None
        call("vag_imagine_attn_ctx_fwd", ptr(im_emb), ptr(enc), ptr(mask), ptr(W_cc), ptr(W_ec), ptr(mw), method, B, Ts,
             Cc, S, ptr(alpha), ptr(cvec), ptr(ws), stream())
        params = (W_cc, W_ec) + ((mlp_w,) if mlp_w is not None else ())
        ctx.save_for_backward(im_emb, enc, mask, alpha, ws, *params)
        ctx.gviews = _grad_views(params)
        ctx.method = method
        ctx.mark_non_differentiable(alpha)
        return alpha, cvec

    @staticmethod
    def backward(ctx, _d_alpha, d_ctx):
        im_emb, enc, mask, alpha, ws = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        B, Ts, Cc = enc.shape
        S = im_emb.shape[1]
        d_ctx = _c(d_ctx)
        d_enc = _f32(B, Ts, Cc, like=enc)
        d_im = _f32(B, S, like=enc)
        t = _grad_targets(ctx.gviews, params)
        mw = params[2].view(-1) if ctx.method == 1 else None
        gm = t[2][0].view(-1) if ctx.method == 1 else None
        call("vag_imagine_attn_ctx_bwd", ptr(im_emb), ptr(enc), ptr(mask), ptr(params[0]), ptr(params[1]), ptr(mw),
             ctx.method, B, Ts, Cc, S, ptr(alpha), ptr(d_ctx), ptr(ws), ptr(d_enc), 0, ptr(d_im), ptr(t[0][0]),
             ptr(t[1][0]), ptr(gm), stream())
        r = _ret(t)
        return (d_im, d_enc, None, r[0], r[1], r[2] if ctx.method == 1 else None, None)


class RankLoss(Function):
    """losses/PairwiseRankingLoss.py:9-24 (kind 0) / ImageRetrievalRankingLoss.py:9-21 (kind 1)."""

    @staticmethod
    def forward(ctx, im, s, margin, kind):
        im, s = _c(im), _c(s)
        B, S = im.shape
        scores = _f32(B, B, like=im)
        G = _f32(B, B, like=im)
        loss = _f32(1, like=im)
        call("vag_rank_loss_fwd", ptr(im), ptr(s), B, S, float(margin), int(kind), ptr(scores), ptr(G), ptr(loss), stream())
        ctx.save_for_backward(im, s, G)
        return loss.view(())

    @staticmethod
    def backward(ctx, d_loss):
        im, s, G = ctx.saved_tensors
        B, S = im.shape
        d_im = _f32(B, S, like=im)
        d_s = _f32(B, S, like=im)
        call("vag_rank_loss_bwd", ptr(im), ptr(s), ptr(G), ptr(_c(d_loss).view(1)), B, S, ptr(d_im), ptr(d_s), stream())
        return d_im, d_s, None, None


class DecInit(Function):
    """h0 = tanh(W (split*ctx + (1-split)*meanpool(enc)) + b)   (models/...V11.py:118; V2.py:85 with ctx=None)."""

    @staticmethod
    def forward(ctx, enc, mask, cvec, W, b, split):
        B, Ts, Cc = enc.shape
        H = W.shape[0]
        xmix = _f32(B, Cc, like=enc)
        h0 = _f32(B, H, like=enc)
        call("vag_dec_init_fwd", ptr(enc), ptr(mask), ptr(_c(cvec)) if cvec is not None else None, float(split), ptr(W),
             ptr(b), B, Ts, Cc, H, ptr(xmix), ptr(h0), stream())
        ctx.save_for_backward(mask, xmix, h0, W, b)
        ctx.gviews = _grad_views((W, b))
        ctx.cfg = (B, Ts, Cc, H, float(split), cvec is not None)
        return h0

    @staticmethod
    def backward(ctx, d_h0):
        mask, xmix, h0, W, b = ctx.saved_tensors
        B, Ts, Cc, H, split, has_ctx = ctx.cfg
        d_h0 = _c(d_h0).clone()
        d_enc = _f32(B, Ts, Cc, like=h0)
        d_ctx = _f32(B, Cc, like=h0) if has_ctx else None
        scratch = _f32(B * Cc, like=h0)
        t = _grad_targets(ctx.gviews, (W, b))
        call("vag_dec_init_bwd", ptr(mask), ptr(xmix), ptr(h0), split, ptr(W), ptr(d_h0), B, Ts, Cc, H, ptr(d_enc), 0,
             ptr(d_ctx), ptr(t[0][0]), ptr(t[1][0]), ptr(scratch), stream())
        return (d_enc, None, d_ctx) + _ret(t) + (None,)


# ----------------------------------------------------------------------------------------------------
# inference-only helpers (no autograd)
# ----------------------------------------------------------------------------------------------------
def decode_prepare(emb, dec):
    """Derived decoder weights for the inference step (once per decode call)."""
    H = dec[1].shape[1]
    prep = _f32(L.lib().vag_cgru_prep_floats(H), like=emb)
    call("vag_cgru_prepare", _dec_w(emb, dec), H, ptr(prep), stream())
    return prep


def decode_hoisted_ok(N, emb, dec, head):
    """The hoisted decoding step (decode_keys / decode_step_h) takes up to 256 hypotheses and 16-byte aligned parameters."""
    E, H = emb.shape[1], dec[1].shape[1]
    ts = [emb, dec[0], head[0], head[4]]
    return N <= 256 and E % 4 == 0 and H % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in ts)


def decode_keys(enc, prep, head, out=None):
    """Once per decode call: [(W_ih2 W_c2h) enc | enc W2^T], what a hoisted decoding step needs of the source side."""
    B, Ts, Cc = enc.shape
    H = Cc // 2
    E = head[2].shape[0]
    keys = out if out is not None else _f32(L.lib().vag_cgru_decode_keys_floats(B, Ts, E, H), like=enc)
    call("vag_cgru_decode_keys", ptr(_c(enc)), ptr(prep), ptr(head[2]), B, Ts, E, H, ptr(keys), stream())
    return keys


DECODE_TABLES_MAX_BYTES = 256 << 20


def decode_tables(emb, dec, head, out=None):
    """Once per decode call: [emb W_ih1^T + b_ih1 (V,3H) | emb W3^T (V,E)] -- what a step needs of a token, for every vocabulary
    entry (the free-running recurrence kernel's tables).  None when they would exceed DECODE_TABLES_MAX_BYTES.  out: a buffer of
    an earlier call to fill instead of a new one."""
    V, E = emb.shape
    H = dec[1].shape[1]
    n = L.lib().vag_cgru_decode_tables_floats(V, E, H)
    if 4 * n > DECODE_TABLES_MAX_BYTES:
        return None
    tables = out if out is not None else _f32(n, like=emb)
    call("vag_cgru_decode_tables", _dec_w(emb, dec), ptr(head[4]), V, E, H, ptr(tables), stream())
    return tables


def decode_step_h(pe, mask, keys, rows_per_src, tok, h_in, emb, dec, prep, tables=None):
    """One cGRU step for N hypotheses in the hoisted form -> (h_out (N,H), cw (N,E) = W2 c, e (N,E) or None with tables,
    alpha (N,Ts))."""
    B, Ts, Cc = pe.shape
    H = Cc // 2
    E = emb.shape[1]
    N = tok.numel()
    h_out = _f32(N, H, like=pe)
    cw = _f32(N, E, like=pe)
    e = _f32(N, E, like=pe) if tables is None else None
    alpha = _f32(N, Ts, like=pe)
    scratch = _f32(L.lib().vag_cgru_step_scratch_floats(N, Ts, E, H), like=pe)
    call("vag_cgru_attn_decode_step_h", ptr(pe), ptr(mask), ptr(keys), ptr(tables) if tables is not None else None, emb.shape[0],
         rows_per_src, ptr(_c(tok).view(-1), I64), ptr(_c(h_in)), _dec_w(emb, dec), ptr(prep), N, Ts, E, H, ptr(h_out), ptr(cw),
         ptr(e) if e is not None else None, ptr(alpha), ptr(scratch), stream())
    return h_out, cw, e, alpha


def decode_step(enc, pe, mask, rows_per_src, tok, h_in, emb, dec, prep):
    """One cGRU step for N hypotheses -> (h_out (N,H), c (N,C), e (N,E), alpha (N,Ts))."""
    B, Ts, Cc = enc.shape
    H = Cc // 2
    E = emb.shape[1]
    N = tok.numel()
    h_out = _f32(N, H, like=enc)
    c = _f32(N, Cc, like=enc)
    e = _f32(N, E, like=enc)
    alpha = _f32(N, Ts, like=enc)
    scratch = _f32(L.lib().vag_cgru_step_scratch_floats(N, Ts, E, H), like=enc)
    call("vag_cgru_attn_decode_step", ptr(enc), ptr(pe), ptr(mask), rows_per_src, ptr(_c(tok).view(-1), I64), ptr(_c(h_in)),
         _dec_w(emb, dec), ptr(prep), N, Ts, E, H, ptr(h_out), ptr(c), ptr(e), ptr(alpha), ptr(scratch), stream())
    return h_out, c, e, alpha


def greedy_decode_supported(B, Ts, steps, E, H, V):
    return bool(L.lib().vag_cgru_free_supported(B, Ts, steps, E, H, V))


def greedy_decode(enc, pe, mask, h0, emb, dec, head, steps, sos):
    """Arg-max decoding for exactly `steps` steps (V11.py:207-226) in ONE launch of the free-running recurrence kernel.
    Returns the chosen tokens (steps, B) int64.  Only for shapes greedy_decode_supported accepts."""
    B, Ts, Cc = enc.shape
    H = Cc // 2
    E = emb.shape[1]
    V = head[7].shape[0]
    tok = torch.empty(steps + 1, B, dtype=I64, device=enc.device)
    tok[0].fill_(sos)
    hseq = _f32(steps, B, H, like=enc)
    tmid = _f32(steps, B, E, like=enc)
    ws = _f32(L.lib().vag_cgru_ws_floats(B, Ts, steps, E, H), like=enc)
    tables = _f32(L.lib().vag_cgru_free_tables_floats(B, Ts, steps, E, H, V), like=enc)
    call("vag_cgru_attn_decode_free_fwd", ptr(_c(enc)), ptr(_c(pe)), ptr(_c(mask)), ptr(_c(h0)), ptr(tok, I64), _dec_w(emb, dec),
         B, Ts, steps, E, H, V, ptr(hseq), None, None, ptr(ws), C.byref(_head_w(head)), 0.0, None, ptr(tmid), None, 0,
         ptr(tables), stream())
    return tok[1:]


def head_logits_parts_count(head, N, E, V):
    """Pieces per row that head_logits_step leaves of every row's log-sum-exp (0: shape not taken, use head_logp_step)."""
    return int(L.lib().vag_head_logits_parts_count(_head_w(head), N, E, V))


def head_logits_step(h2, c, e, head, nparts, hoisted=False, tables=None, tok=None):
    """Raw logits (N, ldl) of one decoding step and the (N, nparts, 2) pieces of their rows' log-sum-exp (beam search on raw
    logits: vag_beam_step_logits_dev)."""
    N, H = h2.shape
    E = head[4].shape[0]
    V = head[7].shape[0]
    ldl = (V + 3) // 4 * 4
    logits = _f32(N, ldl, like=h2)
    parts = _f32(nparts, N, 2, like=h2)
    scratch = _f32(2 * N * E, like=h2)
    if hoisted:
        call("vag_head_logits_step_h", ptr(h2), ptr(c), ptr(e) if e is not None else None,
             ptr(tables) if tables is not None else None, ptr(_c(tok).view(-1), I64) if tables is not None else None,
             _head_w(head), N, E, H, V, ptr(logits), ldl, ptr(parts), ptr(scratch), stream())
    else:
        call("vag_head_logits_step", ptr(h2), ptr(c), ptr(e), _head_w(head), N, E, H, V, ptr(logits), ldl, ptr(parts),
             ptr(scratch), stream())
    return logits, parts


def head_logp_step(h2, c, e, head, want_argmax=False, argmax_out=None, hoisted=False, tables=None, tok=None):
    """argmax_out: optional (N,) int64 HIP tensor the arg-max tokens are written into (greedy decode hands in a row of its token
    chunk, which saves a copy launch per step)."""
    N, H = h2.shape
    E = head[4].shape[0]
    V = head[7].shape[0]
    ldl = (V + 3) // 4 * 4
    logp = _f32(N, ldl, like=h2)
    am = None
    if want_argmax:
        am = argmax_out if argmax_out is not None else torch.empty(N, dtype=I64, device=h2.device)
        assert am.dtype == I64 and am.is_contiguous() and am.numel() == N
    scratch = _f32(2 * N * E, like=h2)
    if hoisted:
        call("vag_head_logp_step_h", ptr(h2), ptr(c), ptr(e) if e is not None else None,
             ptr(tables) if tables is not None else None, ptr(_c(tok).view(-1), I64) if tables is not None else None,
             _head_w(head), N, E, H, V, ptr(logp), ldl, ptr(am, I64) if am is not None else None, ptr(scratch), stream())
    else:
        call("vag_head_logp_step", ptr(h2), ptr(c), ptr(e), _head_w(head), N, E, H, V, ptr(logp), ldl,
             ptr(am, I64) if am is not None else None, ptr(scratch), stream())
    return logp, am


def dropout_mask(rng, which, n, p):
    out = torch.empty(n, dtype=torch.float32, device=rng.device)
    call("vag_dropout_mask", ptr(rng, torch.int64), which, n, float(p), ptr(out), stream())
    return out
