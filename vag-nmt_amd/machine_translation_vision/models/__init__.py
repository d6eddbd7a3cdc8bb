from .NMT_Seq2Seq_Beam_V2 import NMT_Seq2Seq_Beam_V2
from .NMT_AttentionImagine_Seq2Seq_Beam_V11 import NMT_AttentionImagine_Seq2Seq_Beam_V11

from .. import _checkout

_checkout.extend_path(__path__, "models")


def __getattr__(name):
    """Model variants off the hot path (nmt_monomodal_beam_DE.py:15 imports two of them): the checkout's class, or a placeholder."""
    return _checkout.resolve("models", name)
