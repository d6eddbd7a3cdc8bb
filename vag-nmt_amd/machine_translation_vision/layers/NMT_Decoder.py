"""Conditional-GRU decoder with Bahdanau attention, drop-in for layers/NMT_Decoder.py of the reference."""
import math

import torch
import torch.nn as nn

from vagnmt_hip import ops
from vagnmt_hip._lib import call, ptr, stream
from vagnmt_hip.state import dropout_rng


class BahdanauAttn(nn.Module):
    """alpha = softmax_s( v . tanh(W_h h + W_e enc_s) )  (layers/NMT_Decoder.py:10-51).
    Parameter names as in the reference: ``attn_h.weight`` (C,H), ``attn_e.weight`` (C,C), ``v`` (C)."""

    def __init__(self, context_size, hidden_size):
        super(BahdanauAttn, self).__init__()
        self.hidden_size = hidden_size
        self.context_size = context_size
        self.attn_h = nn.Linear(self.hidden_size, self.context_size, bias=False)
        self.attn_e = nn.Linear(self.context_size, self.context_size, bias=False)
        self.v = nn.Parameter(torch.rand(self.context_size))
        stdv = 1. / math.sqrt(self.v.size(0))
        self.v.data.normal_(mean=0, std=stdv)

    def forward(self, hidden, encoder_outputs, ctx_mask=None):
        """hidden (1,B,H); encoder_outputs (S,B,C); ctx_mask (S,B) -> attention weights (B,1,S).  Inference only
        (the training path runs the attention inside the fused sequence operator)."""
        enc = encoder_outputs.transpose(0, 1).contiguous()
        B, S, C = enc.shape
        mask = (torch.ones(B, S, dtype=torch.float32, device=enc.device) if ctx_mask is None
                else ctx_mask.t().contiguous().float())
        with torch.no_grad():
            pe = ops.KeysProj.apply(enc, self.attn_e.weight)
            q = ops.LinearAct.apply(hidden[0], self.attn_h.weight, None, 0)
            scores = torch.empty(B, S, dtype=torch.float32, device=enc.device)
            alpha = torch.empty_like(scores)
            ctx = torch.empty(B, C, dtype=torch.float32, device=enc.device)
            call("vag_bahdanau_attn_fwd", ptr(pe), ptr(q.contiguous()), ptr(self.v), ptr(mask), ptr(enc), B, 1, S, C,
                 ptr(scores), ptr(alpha), ptr(ctx), stream())
        return alpha.unsqueeze(1)


class NMT_Decoder(nn.Module):
    """One cGRU step: emb -> gru_1 -> attention -> context2hid -> gru_2 -> tanh(W1 h + W2 c + W3 e) -> dropout ->
    (tied) vocabulary projection -> log_softmax.  Same constructor/forward as layers/NMT_Decoder.py:55-145 and the
    same parameter names (``embedding``, ``gru_1``, ``attn.{v,attn_h,attn_e}``, ``context2hid``, ``gru_2``,
    ``W1``, ``W2``, ``W3``, ``out``)."""

    def __init__(self, output_size, embedding_size, hidden_size, context_size, n_layers=1, dropout_emb=0.0,
                 dropout_rnn=0.0, dropout_out=0.0, bias_zero=True, tied_emb=False):
        super(NMT_Decoder, self).__init__()
        if n_layers != 1:
            raise NotImplementedError("only n_layers=1 works in the reference as well (NMT_Decoder.py:38)")
        if dropout_emb > 0.0:
            raise NotImplementedError("decoder embedding dropout is never enabled by the reference models (V11.py:67)")
        self.embedding_size = embedding_size
        self.hidden_size = hidden_size
        self.context_size = context_size
        self.n_layers = n_layers
        self.dropout_emb = dropout_emb
        self.dropout_out = dropout_out
        self.bias_zero = bias_zero
        self.tied_emb = tied_emb
        self.embedding = nn.Embedding(output_size, embedding_size, padding_idx=0)
        self.gru_1 = nn.GRU(embedding_size, hidden_size, num_layers=n_layers)
        self.attn = BahdanauAttn(context_size, hidden_size)
        self.context2hid = nn.Linear(context_size, hidden_size, bias=False)
        self.gru_2 = nn.GRU(hidden_size, hidden_size, num_layers=n_layers)
        self.W1 = nn.Linear(hidden_size, embedding_size)
        self.W2 = nn.Linear(context_size, embedding_size)
        self.W3 = nn.Linear(embedding_size, embedding_size)
        self.out = nn.Linear(embedding_size, output_size)
        if self.bias_zero:
            for lin in (self.W1, self.W2, self.W3, self.out):
                torch.nn.init.constant_(lin.bias.data, 0.0)
        if self.tied_emb:
            self.out.weight = self.embedding.weight

    # parameter tuples in the order the C ABI structs expect (include/vag_nmt.h: vag_dec_w / vag_head_w)
    def dec_params(self):
        g1, g2 = self.gru_1, self.gru_2
        return (g1.weight_ih_l0, g1.weight_hh_l0, g1.bias_ih_l0, g1.bias_hh_l0, self.attn.attn_h.weight, self.attn.v,
                self.context2hid.weight, g2.weight_ih_l0, g2.weight_hh_l0, g2.bias_ih_l0, g2.bias_hh_l0)

    def head_params(self):
        return (self.W1.weight, self.W1.bias, self.W2.weight, self.W2.bias, self.W3.weight, self.W3.bias,
                self.out.weight, self.out.bias)

    def forward(self, word_input, last_hidden, encoder_outputs, ctx_mask=None):
        """word_input (B,) or (B,1) int64; last_hidden (1,B,H); encoder_outputs (T,B,C); ctx_mask (T,B).
        Returns (log-probabilities (B,V), hidden (1,B,H)).  Differentiable (a sequence of length one through the
        fused operators); attn_e(encoder_outputs) is recomputed per call exactly as the reference does (:47)."""
        B = word_input.size(0)
        enc = encoder_outputs.transpose(0, 1).contiguous()
        mask = (torch.ones(B, enc.shape[1], dtype=torch.float32, device=enc.device) if ctx_mask is None
                else ctx_mask.t().contiguous().float())
        pe = ops.KeysProj.apply(enc, self.attn.attn_e.weight)
        tok = torch.stack([word_input.reshape(-1), torch.zeros_like(word_input.reshape(-1))], 0)
        h2, c, e = ops.cgru_decode_seq(enc, pe, mask, last_hidden[0].contiguous(), tok, self.embedding.weight,
                                       self.dec_params(), V=self.out.bias.shape[0])
        rng = None
        p_out = 0.0
        if self.training and self.dropout_out > 0.0:
            rng = dropout_rng(self, enc.device)
            call("vag_rng_advance", ptr(rng, torch.int64), stream())
            p_out = float(self.dropout_out)
        logp = ops.HeadLogp.apply(h2[0], c[0], e[0], p_out, rng, *self.head_params())
        return logp, h2
