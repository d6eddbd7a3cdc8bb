"""Text-only baseline, drop-in for models/NMT_Seq2Seq_Beam_V2.py of the reference."""
import torch
import torch.nn as nn

from vagnmt_hip import ops

from ..layers import LIUMCVC_Encoder, NMT_Decoder
from ._seq2seq import Seq2SeqBase


class NMT_Seq2Seq_Beam_V2(Seq2SeqBase):
    """Same constructor / forward / beamsearch_decode as models/NMT_Seq2Seq_Beam_V2.py:19-171.
    forward() returns the single translation loss (V2.py:111-113)."""

    def __init__(self, src_size, tgt_size, src_embedding_size, tgt_embedding_size, hidden_size, beam_size=1, n_layers=1,
                 dropout_ctx=0.0, dropout_emb=0.0, dropout_out=0.0, dropout_rnn=0.0, tied_emb=False):
        super(NMT_Seq2Seq_Beam_V2, self).__init__()
        self.src_size = src_size
        self.tgt_size = tgt_size
        self.src_embedding_size = src_embedding_size
        self.tgt_embedding_size = tgt_embedding_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.beam_size = beam_size
        self.tied_emb = tied_emb
        self.encoder = LIUMCVC_Encoder(src_size, src_embedding_size, hidden_size, n_layers, dropout_rnn=dropout_rnn,
                                       dropout_ctx=dropout_ctx, dropout_emb=dropout_emb)
        self.decoder = NMT_Decoder(tgt_size, tgt_embedding_size, hidden_size, 2 * hidden_size, n_layers,
                                   dropout_out=dropout_out, tied_emb=tied_emb)
        self.decoderini = nn.Linear(2 * hidden_size, hidden_size)
        self.reset_parameters()

    def _prologue(self, src_var, src_lengths, rng):
        enc, mask = self._encode(src_var, src_lengths, rng)
        h0 = ops.DecInit.apply(enc, mask, None, self.decoderini.weight, self.decoderini.bias, 0.0)   # V2.py:85
        return enc, mask, h0

    def forward(self, src_var, src_lengths, tgt_var, teacher_force_ratio=1.0, max_length=80, criterion=None):
        self.tgt_l = tgt_var.size()[1]
        rng = self._train_rng(src_var.device)
        enc, mask, h0 = self._prologue(src_var, src_lengths, rng)
        return self._translation_loss(enc, mask, h0, tgt_var, teacher_force_ratio, criterion, rng)

    def beamsearch_decode(self, src_var, src_lengths, beam_size=1, max_length=80, tgt_var=None):
        tgt_l = max_length
        if tgt_var is not None:
            tgt_l = tgt_var.size()[1]
        self.tgt_l = tgt_l
        self.beam_size = beam_size
        with torch.no_grad():
            enc, mask, h0 = self._prologue(src_var, src_lengths, None)
            if beam_size == 1:
                self.final_sample = self._greedy(enc, mask, h0, tgt_l)
            else:
                self.final_sample = self._beam(enc, mask, h0, beam_size, tgt_l)
        return self.final_sample
