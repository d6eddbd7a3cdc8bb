from .Encoder import LIUMCVC_Encoder
from .NMT_Decoder import NMT_Decoder, BahdanauAttn
from .VSE_Imagine_Enc import VSE_Imagine_Enc, ImagineAttn
