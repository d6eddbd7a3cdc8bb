"""Workload for PMC passes: N launches of the decoder sequence forward (cfg2) + 200 isolated GRU-cell launches
(forward) + 200 isolated GRU backward-step launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vag-nmt_amd"))
import torch
import bench
from vagnmt_hip import ops
from vagnmt_hip._lib import call, ptr, stream
c = bench.CFG2
dev = torch.device("cuda:0")
m = bench.build_model(c, dev).eval()
src, lens, tgt, im = bench.make_batch(c, 0, dev)
lens_t = torch.tensor(lens, dtype=torch.int32, device=dev)
with torch.no_grad():
    enc, mask = m._encode(src, lens_t, None)
    for _ in range(3):
        m._encode(src, lens_t, None)           # the persistent encoder kernel, for its traffic per launch
    _, ctx = m.vse_imagine.forward_bm(im, enc, mask, None)
    h0 = ops.DecInit.apply(enc, mask, ctx, m.decoderini.weight, m.decoderini.bias, 0.5)
    pe = ops.KeysProj.apply(enc, m.decoder.attn.attn_e.weight)
    sos = torch.full((1, c["B"]), 2, dtype=torch.int64, device=dev)
    tok = torch.cat([sos, tgt.t()], 0).contiguous()
    dec = m.decoder
    marker = torch.zeros(2, dtype=torch.int64, device=dev)      # rng_advance_kernel launches bracket the sequence calls
    ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), V=c["V"])     # warm-up
    call("vag_rng_advance", ptr(marker, torch.int64), stream())
    for _ in range(3):
        ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), V=c["V"])
    call("vag_rng_advance", ptr(marker, torch.int64), stream())
    B, H = c["B"], c["H"]
    gi = torch.randn(B, 3 * H, device=dev); hp = torch.randn(B, H, device=dev)
    ho = torch.empty(B, H, device=dev); sv = torch.empty(4, B, H, device=dev)
    for _ in range(200):
        call("vag_gru_cell_fwd", ptr(gi), ptr(hp), ptr(dec.gru_1.weight_hh_l0), ptr(dec.gru_1.bias_hh_l0), B, H, ptr(ho), ptr(sv), stream())
    # 200 isolated launches of the backward step kernel (encoder / decoder gru_1 shape)
    dgh_next = torch.randn(B, 3 * H, device=dev); wt = dec.gru_1.weight_hh_l0.t().contiguous()
    carry = torch.randn(B, H, device=dev); d_out = torch.randn(B, H, device=dev); sv.uniform_(0.1, 0.9)
    dgi = torch.empty(B, 3 * H, device=dev); dgh = torch.empty(B, 3 * H, device=dev); cout = torch.empty(B, H, device=dev)
    for _ in range(200):
        call("vag_gru_cell_bwd", ptr(dgh_next), ptr(wt), ptr(carry), ptr(d_out), ptr(sv), ptr(hp), B, H, ptr(dgi), ptr(dgh),
             ptr(cout), stream())
torch.cuda.synchronize()
# three full optimiser steps (eager launches of vag_train_step): the backward persistent recurrences, for their traffic per launch
from vagnmt_hip.trainer import TrainStep
from machine_translation_vision.losses import PairwiseRankingLoss
mt = bench.build_model(c, dev)
vw = torch.ones(c["V"], device=dev); vw[0] = 0
ts = TrainStep(mt, torch.nn.NLLLoss(weight=vw, reduction="none"), PairwiseRankingLoss(margin=0.1), lr=4e-4, weight_decay=1e-5,
               clip=1.0, teacher_force_ratio=1.0, use_graph=False)
for _ in range(3):
    ts.step(src, lens_t, tgt, im)
torch.cuda.synchronize()
ts.check()
print("done")
