"""Binding of the fused training step (vag_train_step, include/vag_nmt.h): the model's parameters and their flat-gradient
views as the C structs, the shape/config struct, the workspace and the static input buffers.

One ``FusedStep`` serves every batch shape: the workspace and the input buffers are sized for the largest shape seen and
each (B, Ts, Tt) batch is laid out densely at their front, so HIP graphs captured for different shapes share all memory
(a captured step owns nothing but its kernel nodes)."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import DecW, GruW, HeadW, ModelW, StepCfg, call, gru_w, ptr, stream
from .state import dropout_rng


def fusable(model, criterion_mt, criterion_vse):
    """The fused step implements the reference's own criteria (nmt_multimodal_beam_DE.py:291-299); anything else goes
    through the per-operator autograd path."""
    from machine_translation_vision.losses import ImageRetrievalRankingLoss, PairwiseRankingLoss
    ok_mt = (type(criterion_mt) is nn.NLLLoss and criterion_mt.reduction == 'none' and criterion_mt.weight is not None
             and criterion_mt.ignore_index < 0)
    ok_vse = criterion_vse is None or type(criterion_vse) in (PairwiseRankingLoss, ImageRetrievalRankingLoss)
    return ok_mt and ok_vse and hasattr(model, "encoder") and hasattr(model, "decoder")


def _views(params, grad):
    return [p._vag_grad if grad else p for p in params]


def model_struct(model, grad=False):
    """vag_model_w (grad=False) or vag_model_g (grad=True: the parameters' views into the flat gradient buffer)."""
    enc, dec = model.encoder, model.decoder
    g = enc.gru
    t = lambda p: ptr(p._vag_grad if grad else p)       # noqa: E731
    gw = lambda a, b, c, d: GruW(t(a), t(b), t(c), t(d))   # noqa: E731
    mm = hasattr(model, "vse_imagine")
    s = ModelW()
    s.enc_emb = t(enc.embedding.weight)
    s.enc_fw = gw(g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
    s.enc_bw = gw(g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse)
    if mm:
        v = model.vse_imagine
        s.im_w, s.im_b = t(v.im_embedding.weight), t(v.im_embedding.bias)
        s.txt_w, s.txt_b = t(v.text_embedding.weight), t(v.text_embedding.bias)
        s.ctx2ctx, s.emb2ctx = t(v.imagine_attn.ctx2ctx.weight), t(v.imagine_attn.emb2ctx.weight)
        s.mlp_w = t(v.imagine_attn.mlp.weight) if v.imagine_attn.method == "mlp" else None
    s.ini_w, s.ini_b = t(model.decoderini.weight), t(model.decoderini.bias)
    s.attn_e = t(dec.attn.attn_e.weight)
    dp = dec.dec_params()
    s.dec = DecW(t(dec.embedding.weight), gw(dp[0], dp[1], dp[2], dp[3]), t(dp[4]), t(dp[5]), t(dp[6]),
                 gw(dp[7], dp[8], dp[9], dp[10]))
    s.head = HeadW(*[t(p) for p in dec.head_params()])
    return s


class FusedStep:
    LOSS_RING = 256
    def __init__(self, model, criterion_mt, criterion_vse, storage="f32"):
        """storage: "f32", or "f16" = BASELINE configs[4]'s 2-byte storage: the recurrences' weights and the attention keys
        are kept as fp16 in HBM on teacher-forced steps (free-running steps of the same driver use fp32 storage)."""
        if storage not in ("f32", "f16"):
            raise ValueError("storage must be 'f32' or 'f16'")
        self.storage16 = storage == "f16"
        self.model = model
        self.mm = hasattr(model, "vse_imagine")
        self.vw = criterion_mt.weight
        self.dev = self.vw.device
        dec, enc = model.decoder, model.encoder
        self.H = dec.hidden_size
        self.V = dec.out.bias.shape[0]
        self.ldl = (self.V + 3) // 4 * 4
        self.Es = enc.embedding.weight.shape[1]
        self.Et = dec.embedding.weight.shape[1]
        if self.storage16 and self.H % 8 != 0:
            raise ValueError("fp16 storage needs hidden_size % 8 == 0")
        self.S = model.vse_imagine.shared_embedding_size if self.mm else 0
        self.I = model.vse_imagine.im_size if self.mm else 0
        self.rank_kind, self.margin = -1, 0.0
        if self.mm and criterion_vse is not None:
            from machine_translation_vision.losses import PairwiseRankingLoss
            self.rank_kind = 0 if type(criterion_vse) is PairwiseRankingLoss else 1
            self.margin = float(criterion_vse.margin)
        self.w = model_struct(model, grad=False)
        self.g = model_struct(model, grad=True)
        self.derived = torch.empty(L.lib().vag_derived_floats(self.H), dtype=torch.float32, device=self.dev)
        # [loss, loss_mt, loss_vse, execution count] + a ring of the last LOSS_RING steps' results (vag_step_cfg.loss_ring): what a
        # step hands out stays valid for that many further steps without a copy launch per step
        self.losses = torch.zeros(4 + 4 * self.LOSS_RING, dtype=torch.float32, device=self.dev)
        self.executed = 0             # forward phases executed so far (eager calls and graph replays; the device counts the same)
        self.ws = None
        self.cap = (0, 0, 0)          # (B*Ts, B*Tt, B) capacity of the static input buffers
        self.src = self.tgt = self.im = self.lens = None
        self.generation = 0           # bumped whenever a static buffer is re-allocated (captured graphs are then stale)
        # device address of the owning driver's guard pair {void flag, give-up count} (vag_step_cfg.guard; TrainStep keeps it in its
        # optimiser scratch) or None: the process-wide pair, which no optimiser reads
        self.guard = None
        self.refresh_derived()

    # ---- configuration ----
    def cfg(self, B, Ts, Tt, teacher, train=True):
        m = self.model
        c = StepCfg()
        c.B, c.Ts, c.Tt, c.Es, c.Et, c.H, c.S, c.I, c.V, c.ldl = B, Ts, Tt, self.Es, self.Et, self.H, self.S, self.I, self.V, self.ldl
        c.multimodal = 1 if self.mm else 0
        c.attn_method = 1 if (self.mm and m.vse_imagine.imagine_attn.method == "mlp") else 0
        c.activation_vse = 1 if (self.mm and m.vse_imagine.activation_vse) else 0
        c.rank_kind = self.rank_kind
        c.free_run = 0 if teacher else 1
        c.storage = 1 if (self.storage16 and teacher) else 0
        c.margin = self.margin
        c.loss_w = float(m.loss_w) if self.mm else 1.0
        c.init_split = float(m.init_split) if self.mm else 0.0
        c.p_emb = float(m.encoder.dropout_emb) if train else 0.0
        c.p_ctx = float(m.encoder.dropout_ctx) if train else 0.0
        c.p_out = float(m.decoder.dropout_out) if train else 0.0
        c.loss_ring = self.LOSS_RING
        c.guard = self.guard
        return c

    def reserve(self, B, Ts, Tt):
        """Make the static buffers large enough for a (B,Ts,Tt) batch; returns True if anything was re-allocated."""
        c = self.cfg(B, Ts, Tt, True)
        need = L.lib().vag_step_ws_floats(C.byref(c))
        if need < 0:
            raise L.VagError("vag_step_ws_floats: bad configuration")
        grown = False
        if self.ws is None or self.ws.numel() < need:
            self.ws = torch.empty(int(need * 1.25), dtype=torch.float32, device=self.dev)
            grown = True
        cap = (max(self.cap[0], B * Ts), max(self.cap[1], B * Tt), max(self.cap[2], B))
        if cap != self.cap:
            self.cap = cap
            self.src = torch.zeros(cap[0], dtype=torch.int64, device=self.dev)
            self.tgt = torch.zeros(cap[1], dtype=torch.int64, device=self.dev)
            self.lens = torch.zeros(cap[2], dtype=torch.int32, device=self.dev)
            self.im = torch.zeros(cap[2] * max(self.I, 1), dtype=torch.float32, device=self.dev)
            grown = True
        if grown:
            self.generation += 1
        return grown

    def load_batch(self, src, lengths, tgt, im):
        """The batch into the static buffers, densely at their front: one launch (vag_copy4)."""
        B, Ts = src.shape
        Tt = tgt.shape[1]
        srcs = [src, tgt, lengths] + ([im] if self.mm else [])
        dsts = [self.src, self.tgt, self.lens] + ([self.im] if self.mm else [])
        for t in srcs:
            if not (t.is_cuda and t.is_contiguous()):
                raise L.VagError("batch tensors must be contiguous HIP tensors")
        if src.dtype != torch.int64 or tgt.dtype != torch.int64 or lengths.dtype != torch.int32 or \
                (self.mm and im.dtype != torch.float32):
            raise L.VagError("batch dtypes: src/tgt int64, lengths int32, im float32")
        n = len(srcs)
        sp = (C.c_void_p * 4)(*[t.data_ptr() for t in srcs])
        dp = (C.c_void_p * 4)(*[t.data_ptr() for t in dsts])
        nb = (C.c_int64 * 4)(*[t.numel() * t.element_size() for t in srcs])
        assert srcs[0].numel() == B * Ts and srcs[1].numel() == B * Tt and srcs[2].numel() == B
        call("vag_copy4", sp, dp, nb, n, stream())

    # ---- launches ----
    def refresh_derived(self):
        """Per optimiser step: the stacked / folded / transposed weights the recurrences read."""
        g = self.model.encoder.gru
        call("vag_derive_weights", self.w.dec, ptr(g.weight_hh_l0), ptr(g.weight_hh_l0_reverse), self.H,
             1 if self.storage16 else 0, ptr(self.derived), stream())

    def run(self, B, Ts, Tt, teacher, phases):
        m = self.model
        train = m.training
        c = self.cfg(B, Ts, Tt, teacher, train)
        rng = None
        if train and max(c.p_emb, c.p_ctx, c.p_out) > 0:
            rng = dropout_rng(m, self.dev)
        if (int(phases) & 1) and not (self.dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self.executed += 1
        call("vag_train_step", C.byref(c), C.byref(self.w), C.byref(self.g), ptr(self.src, torch.int64),
             ptr(self.lens, torch.int32), ptr(self.tgt, torch.int64), ptr(self.im) if self.mm else None, ptr(self.vw),
             ptr(rng, torch.int64) if rng is not None else None, ptr(self.derived), ptr(self.ws), ptr(self.losses),
             int(phases), stream())
