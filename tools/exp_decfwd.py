"""Experiment: decoder sequence forward (vag_cgru_attn_decode_seq_fwd, cfg2) per step under the library's switches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch, bench
from vagnmt_hip import ops
c = bench.CFG2
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lt = torch.tensor(lens, dtype=torch.int32, device=dev)
with torch.no_grad():
    enc, mask = m._encode(src, lt, None)
    _, ctx = m.vse_imagine.forward_bm(im, enc, mask, None)
    h0 = ops.DecInit.apply(enc, mask, ctx, m.decoderini.weight, m.decoderini.bias, 0.5)
    pe = ops.KeysProj.apply(enc, m.decoder.attn.attn_e.weight)
    sos = torch.full((1, c["B"]), 2, dtype=torch.int64, device=dev)
    tok = torch.cat([sos, tgt.t()], 0).contiguous()
    dec = m.decoder
    t = bench._time_graph(lambda: ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), V=c["V"]))
    print("%s: decoder_seq_fwd %.1f us = %.2f us/step" % (os.environ.get("TAG", "default"), t * 1e6, t * 1e6 / c["Tt"]))
