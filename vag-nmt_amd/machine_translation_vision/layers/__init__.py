from .Encoder import LIUMCVC_Encoder
from .NMT_Decoder import NMT_Decoder, BahdanauAttn
from .VSE_Imagine_Enc import VSE_Imagine_Enc, ImagineAttn

from .. import _checkout

_checkout.extend_path(__path__, "layers")


def __getattr__(name):
    """Layers of the model variants off the hot path: the checkout's class, or a placeholder."""
    return _checkout.resolve("layers", name)
