"""VAG-NMT multimodal model, drop-in for models/NMT_AttentionImagine_Seq2Seq_Beam_V11.py of the reference."""
import torch
import torch.nn as nn

from vagnmt_hip import ops

from ..layers import LIUMCVC_Encoder, NMT_Decoder, VSE_Imagine_Enc
from ._seq2seq import Seq2SeqBase, SOS_token, EOS_token


class NMT_AttentionImagine_Seq2Seq_Beam_V11(Seq2SeqBase):
    """Same positional constructor arguments, attributes and method signatures as the reference class
    (models/...V11.py:21-80).  forward() returns (loss, loss_mt, loss_vse) with
    loss = loss_w * loss_mt + (1 - loss_w) * loss_vse (V11.py:166)."""

    def __init__(self, src_size, tgt_size, im_feats_size, src_embedding_size, tgt_embedding_size, hidden_size,
                 shared_embedding_size, loss_w, beam_size=1, attn_model='dot', n_layers=1, dropout_ctx=0.0,
                 dropout_emb=0.0, dropout_out=0.0, dropout_rnn_enc=0.0, dropout_rnn_dec=0.0, dropout_im_emb=0.0,
                 dropout_txt_emb=0.0, activation_vse=True, tied_emb=False, init_split=0.5):
        super(NMT_AttentionImagine_Seq2Seq_Beam_V11, self).__init__()
        self.src_size = src_size
        self.tgt_size = tgt_size
        self.im_feats_size = im_feats_size
        self.src_embedding_size = src_embedding_size
        self.tgt_embedding_size = tgt_embedding_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.shared_embedding_size = shared_embedding_size
        self.beam_size = beam_size
        self.loss_w = loss_w
        self.tied_emb = tied_emb
        self.dropout_im_emb = dropout_im_emb
        self.dropout_txt_emb = dropout_txt_emb
        self.activation_vse = activation_vse
        self.attn_model = attn_model
        self.init_split = init_split
        self.encoder = LIUMCVC_Encoder(src_size, src_embedding_size, hidden_size, n_layers, dropout_rnn=dropout_rnn_enc,
                                       dropout_ctx=dropout_ctx, dropout_emb=dropout_emb)
        self.decoder = NMT_Decoder(tgt_size, tgt_embedding_size, hidden_size, 2 * hidden_size, n_layers,
                                   dropout_rnn=dropout_rnn_dec, dropout_out=dropout_out, dropout_emb=0.0,
                                   tied_emb=tied_emb)
        self.vse_imagine = VSE_Imagine_Enc(self.attn_model, self.im_feats_size, 2 * hidden_size,
                                           self.shared_embedding_size, self.dropout_im_emb, self.dropout_txt_emb,
                                           self.activation_vse)
        self.decoderini = nn.Linear(2 * hidden_size, hidden_size)
        self.reset_parameters()

    def _prologue(self, src_var, src_lengths, im_var, criterion_vse, rng):
        enc, mask = self._encode(src_var, src_lengths, rng)
        loss_vse, ctx = self.vse_imagine.forward_bm(im_var, enc, mask, criterion_vse)
        h0 = ops.DecInit.apply(enc, mask, ctx, self.decoderini.weight, self.decoderini.bias, self.init_split)
        return enc, mask, loss_vse, h0

    def forward(self, src_var, src_lengths, tgt_var, im_var, teacher_force_ratio=1.0, max_length=80, criterion_mt=None,
                criterion_vse=None):
        """src_var (B,W_s) int64 (pad 0, rows sorted by length, descending); src_lengths list[B]; tgt_var (B,W_t) int64;
        im_var (B,I) fp32.  Returns (loss, loss_mt, loss_vse)."""
        self.tgt_l = tgt_var.size()[1]
        rng = self._train_rng(src_var.device)
        enc, mask, loss_vse, h0 = self._prologue(src_var, src_lengths, im_var, criterion_vse, rng)
        loss_mt = self._translation_loss(enc, mask, h0, tgt_var, teacher_force_ratio, criterion_mt, rng)
        loss = self.loss_w * loss_mt + (1 - self.loss_w) * loss_vse
        return loss, loss_mt, loss_vse

    def beamsearch_decode(self, src_var, src_lengths, im_var, beam_size=1, max_length=80, tgt_var=None):
        tgt_l = max_length
        if tgt_var is not None:
            tgt_l = tgt_var.size()[1]
        self.tgt_l = tgt_l
        self.beam_size = beam_size
        with torch.no_grad():
            enc, mask, _, h0 = self._prologue(src_var, src_lengths, im_var, None, None)
            if beam_size == 1:
                self.final_sample = self._greedy(enc, mask, h0, tgt_l)
            else:
                self.final_sample = self._beam(enc, mask, h0, beam_size, tgt_l)
        return self.final_sample

    # ---- image retrieval (V11.py:341-397) ----
    def embed_sent_im_eval(self, src_var, src_lengths, tgt_var, im_feats):
        self.tgt_l = tgt_var.size()[1]
        return self._embed(src_var, src_lengths, im_feats)

    def embed_sent_im_test(self, src_var, src_lengths, im_feats, max_length=80):
        self.tgt_l = max_length
        return self._embed(src_var, src_lengths, im_feats)

    def _embed(self, src_var, src_lengths, im_feats):
        with torch.no_grad():
            enc, mask = self._encode(src_var, src_lengths, None)
            im_emb, txt_emb, _, _ = self.vse_imagine.embed_bm(im_feats, enc, mask)
        return im_emb.data, txt_emb.data

    def get_imagine_attention_eval(self, src_var, src_lengths, tgt_var, im_feats):
        self.tgt_l = tgt_var.size()[1]
        return self._imagine_weights(src_var, src_lengths, im_feats)

    def get_imagine_attention_test(self, src_var, src_lengths, im_feats, max_length=80):
        self.tgt_l = max_length
        return self._imagine_weights(src_var, src_lengths, im_feats)

    def _imagine_weights(self, src_var, src_lengths, im_feats):
        with torch.no_grad():
            enc, mask = self._encode(src_var, src_lengths, None)
            _, _, alpha, _ = self.vse_imagine.embed_bm(im_feats, enc, mask)
        return alpha.unsqueeze(1).data
