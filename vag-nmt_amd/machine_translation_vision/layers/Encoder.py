"""bi-GRU encoder, drop-in for layers/Encoder.py of the reference."""
import torch
import torch.nn as nn

from vagnmt_hip import ops
from vagnmt_hip._lib import call, ptr, stream
from vagnmt_hip.state import dropout_rng, lengths_tensor


class LIUMCVC_Encoder(nn.Module):
    """Embedding(padding_idx=0) -> dropout -> bidirectional GRU over variable-length rows -> dropout.

    Constructor and forward signature of the reference class (layers/Encoder.py:12-66); parameter names are
    identical (``embedding.weight``, ``gru.weight_ih_l0`` ... ``gru.bias_hh_l0_reverse``), so reference
    state_dicts load.  The nn.GRU / nn.Embedding members only hold parameters: the arithmetic is the
    ``vag_bigru_seq_*`` HIP path (input projection as one MFMA GEMM, one fused recurrent kernel per time
    step covering both directions, length masking and output dropout fused)."""

    def __init__(self, input_size, embedding_size, hidden_size, n_layers=1, dropout_rnn=0, dropout_emb=0, dropout_ctx=0):
        super(LIUMCVC_Encoder, self).__init__()
        if n_layers != 1:
            raise NotImplementedError("only n_layers=1 works in the reference as well (BahdanauAttn.forward, NMT_Decoder.py:38)")
        self.n_layers = n_layers
        self.hidden_size = hidden_size
        self.n_direction = 2
        self.dropout_rnn = dropout_rnn      # no effect on a single-layer GRU (torch applies it between layers)
        self.dropout_emb = dropout_emb
        self.dropout_ctx = dropout_ctx
        self.embedding = nn.Embedding(input_size, embedding_size, padding_idx=0)
        self.gru = nn.GRU(embedding_size, hidden_size, num_layers=n_layers, bidirectional=True)

    def encode_bm(self, input_var, input_lengths):
        """Batch-major result used inside the models: enc (B,Ts,2H), mask (B,Ts)."""
        g = self.gru
        train = self.training
        rng = None
        if train and (self.dropout_emb > 0 or self.dropout_ctx > 0):
            # stand-alone layer call: this module owns the generator, so it draws fresh masks per call like the reference
            # (inside the models the step counter is advanced once per forward by Seq2SeqBase._train_rng)
            rng = dropout_rng(self, input_var.device)
            call("vag_rng_advance", ptr(rng, torch.int64), stream())
        return ops.BiGRUEncode.apply(
            input_var, lengths_tensor(input_lengths, input_var.device), self.embedding.weight,
            g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
            g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse,
            float(self.dropout_emb) if train else 0.0, float(self.dropout_ctx) if train else 0.0, rng)

    def forward(self, input_var, input_lengths):
        """input_var (B,W) int64 padded with 0; input_lengths list/tensor (descending).
        Returns (output (W,B,2H), ctx_mask float (W,B)) like the reference."""
        enc, mask = self.encode_bm(input_var, input_lengths)
        return enc.transpose(0, 1), mask.t()
