"""Shared machinery of the two model classes on the hot path (text-only V2 and multimodal V11)."""
import random

import torch
import torch.nn as nn

from vagnmt_hip import ops
from vagnmt_hip._lib import call, ptr, stream
from vagnmt_hip import _lib
from vagnmt_hip.state import dropout_rng

SOS_token = 2
EOS_token = 3
UNK_token = 1


class Seq2SeqBase(nn.Module):
    """encoder -> [visual grounding] -> decoder init -> cGRU decoder over the target -> loss / decode."""

    def __getstate__(self):
        """Whole-module pickles (``torch.save(model)``, nmt_multimodal_beam_DE.py:491-520, right after an evaluation pass) carry the
        module, not what decoding cached on it: static buffers, per-weights tables and captured HIP graphs of the decode shapes
        (``_decode_cache`` / ``_decode_wcache``: a CUDAGraph does not pickle) are rebuilt on the next decode call."""
        state = dict(self.__dict__)
        for k in [k for k in state if k.startswith("_decode_")]:
            del state[k]
        return state

    def reset_parameters(self):
        # models/...V11.py:77-80: every >=2-D non-bias parameter (embeddings included) gets kaiming_normal_
        for name, param in self.named_parameters():
            if param.requires_grad and 'bias' not in name and param.data.dim() > 1:
                nn.init.kaiming_normal_(param.data)

    # ------------------------------------------------------------------------------------------ training
    def _train_rng(self, device):
        """Advance the dropout counter once per training forward; backward re-derives the same masks."""
        if not self.training:
            return None
        if max(self.encoder.dropout_emb, self.encoder.dropout_ctx, self.decoder.dropout_out) <= 0:
            return None
        rng = dropout_rng(self, device)
        call("vag_rng_advance", ptr(rng, torch.int64), stream())
        return rng

    def _encode(self, src_var, src_lengths, rng):
        enc_mod = self.encoder
        g = enc_mod.gru
        train = self.training and rng is not None
        from vagnmt_hip.state import lengths_tensor
        return ops.BiGRUEncode.apply(
            src_var, lengths_tensor(src_lengths, src_var.device), enc_mod.embedding.weight,
            g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
            g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse,
            float(enc_mod.dropout_emb) if train else 0.0, float(enc_mod.dropout_ctx) if train else 0.0, rng)

    def _translation_loss(self, enc, mask, h0, tgt_var, teacher_force_ratio, criterion, rng):
        """The decoder loop of models/...V11.py:136-164 as two fused operators.
        teacher forcing (python ``random`` coin, V11.py:136): whole-sequence cGRU + batched head/CE;
        free running: the sequence operator also runs the head per step and feeds back the argmax."""
        dec = self.decoder
        B, Tt = tgt_var.shape
        V = dec.out.bias.shape[0]
        ldl = (V + 3) // 4 * 4
        pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
        is_teacher = random.random() < teacher_force_ratio
        sos = torch.full((1, B), SOS_token, dtype=torch.int64, device=tgt_var.device)
        tok = torch.cat([sos, tgt_var.t()], 0).contiguous()          # (Tt+1,B): inputs of steps 0..Tt-1 (+1 unused row)
        p_out = float(dec.dropout_out) if (self.training and rng is not None) else 0.0
        head = dec.head_params()
        fused = (type(criterion) is nn.NLLLoss and criterion.reduction == 'none' and criterion.weight is not None
                 and criterion.ignore_index < 0)
        if is_teacher:
            h2, c, e = ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(), V=V)
            tmid = logits = None
        else:
            h2, c, e, tmid, logits = ops.cgru_decode_seq(enc, pe, mask, h0, tok, dec.embedding.weight, dec.dec_params(),
                                                         free_run=True, head=head, p_out=p_out, rng=rng, V=V, ldl=ldl)
        if fused:
            return ops.HeadCE.apply(h2, c, e, tgt_var, criterion.weight, p_out, rng, tmid, logits, ldl, *head)
        # any other criterion: materialise the log-probabilities and call it per step, as the reference does
        H = h2.shape[2]
        logp = ops.HeadLogp.apply(h2.view(Tt * B, H), c.view(Tt * B, 2 * H), e.view(Tt * B, -1), p_out, rng, *head)
        logp = logp.view(Tt, B, V)
        loss = 0
        for di in range(Tt):
            loss = loss + criterion(logp[di], tgt_var[:, di])
        tgt_mask = (tgt_var != 0).float()
        return (loss / tgt_mask.sum(-1)).mean()

    # ------------------------------------------------------------------------------------------ decoding
    # Both decoders replay ONE captured HIP graph of DECODE_CHUNK steps per (batch, beam, padded source length): the
    # step index of the beam search lives in device memory (vag_beam_step_dev), the source side is padded to a
    # multiple of 8 positions with mask 0 (exactly zero attention weight, so results do not change), and the host
    # only looks at the device every chunk.  ``model.decode_graph = False`` runs the same kernels launch by launch.
    DECODE_CHUNK = 8
    decode_graph = True
    decode_persistent = True      # greedy decoding in one launch where the shape allows it (ops.greedy_decode)
    decode_raw_logits = True      # beam search: expansion on raw logits + log-sum-exp pieces (no normalising pass) where available
    decode_hoisted = True         # decoding steps on keys projected once per call (4 launches, no context): ops.decode_step_h

    def _decode_pool(self):
        """One graph memory pool for every decode shape this model captures (their per-step outputs are allocated inside the captures):
        bounded by the largest shapes met, not by the number of shapes.  The handle lives IN the decode cache: a pool
        dies with the last graph that used it, and the cache is where those graphs live -- dropped or cleared together."""
        cache = self.__dict__.setdefault("_decode_cache", {})
        pool = cache.get("__pool__")
        if pool is None:
            pool = cache["__pool__"] = torch.cuda.graph_pool_handle()
        return pool

    def _decode_weights(self, dp, hp, emb, hoisted):
        """What decoding derives from the WEIGHTS alone -- the stacked / folded decoder matrices (ops.decode_prepare) and, for
        the hoisted step, the per-token tables (ops.decode_tables) -- shared by every decode shape and kept until the weights
        change: an optimiser step of the step driver bumps ``_vag_weights_version``, anything else that writes a parameter in
        place bumps the tensor's own version counter, a re-assigned parameter has another address.  Refreshed in place, so the
        captured graphs that read these buffers stay valid."""
        ts = list(dp) + list(hp) + [emb]
        ptrs = tuple(t.data_ptr() for t in ts) + (hoisted,)
        ver = (getattr(self, "_vag_weights_version", 0),) + tuple(int(t._version) for t in ts)
        wc = self.__dict__.setdefault("_decode_wcache", {})
        e = wc.get(ptrs)
        if e is None:
            if len(wc) >= 4:
                wc.clear()
                self.__dict__.pop("_decode_cache", None)          # (their graphs point at the buffers just dropped)
            H = dp[1].shape[1]
            e = {"prep": torch.empty(_lib.lib().vag_cgru_prep_floats(H), device=emb.device), "tables": None, "ver": None}
            wc[ptrs] = e
        if e["ver"] != ver:
            e["prep"].copy_(ops.decode_prepare(emb, dp))
            if hoisted:
                if e["ver"] is None:
                    e["tables"] = ops.decode_tables(emb, dp, hp)
                elif e["tables"] is not None:
                    ops.decode_tables(emb, dp, hp, out=e["tables"])
            e["ver"] = ver
        return e

    def _decode_state(self, kind, enc, mask, k, max_length):
        """Static buffers (+ captured graph, filled in by the caller) for one decode shape; refreshed per call."""
        dec = self.decoder
        B, Ts, C = enc.shape
        H = C // 2
        dev = enc.device
        dp, hp, emb = dec.dec_params(), dec.head_params(), dec.embedding.weight
        Tp = (Ts + 7) // 8 * 8
        hoisted = self.decode_hoisted and ops.decode_hoisted_ok(B * k, emb, dp, hp)
        wd = self._decode_weights(dp, hp, emb, hoisted)
        key = (kind, B, k, Tp, max_length, self.decode_raw_logits, hoisted) + \
            tuple(t.data_ptr() for t in list(dp) + list(hp) + [emb, dec.attn.attn_e.weight])
        cache = self.__dict__.setdefault("_decode_cache", {})
        st = cache.get(key)
        if st is None:
            if len(cache) >= 64:
                cache.clear()
            st = {"enc": torch.zeros(B, Tp, C, device=dev), "pe": torch.zeros(B, Tp, C, device=dev),
                  "mask": torch.zeros(B, Tp, device=dev), "h": torch.empty(B * k, H, device=dev),
                  "tok": torch.empty(B * k, dtype=torch.int64, device=dev),
                  "prep": wd["prep"], "tables": wd["tables"], "graph": None, "hoisted": hoisted}
            if hoisted:
                st["keys"] = torch.empty(_lib.lib().vag_cgru_decode_keys_floats(B, Tp, emb.shape[1], H), device=dev)
            cache[key] = st
        if Ts < Tp:
            st["enc"][:, Ts:].zero_(); st["pe"][:, Ts:].zero_(); st["mask"][:, Ts:].zero_()
        st["enc"][:, :Ts].copy_(enc)
        st["pe"][:, :Ts].copy_(ops.KeysProj.apply(enc, dec.attn.attn_e.weight))
        st["mask"][:, :Ts].copy_(mask)
        if hoisted:
            ops.decode_keys(st["enc"], st["prep"], hp, out=st["keys"])
        return st, dp, hp, emb

    def _greedy(self, enc, mask, h, tgt_l):
        """beam_size == 1 branch (V11.py:207-226): argmax for exactly tgt_l steps, cut at EOS on the host."""
        dec = self.decoder
        B = enc.shape[0]
        dev = enc.device
        toks = torch.empty(tgt_l, B, dtype=torch.int64, device=dev)
        self.last_decode_steps = tgt_l
        if self.decode_persistent and enc.is_cuda:
            dp, hp, emb = dec.dec_params(), dec.head_params(), dec.embedding.weight
            if ops.greedy_decode_supported(B, enc.shape[1], tgt_l, emb.shape[1], h.shape[1], hp[7].shape[0]):
                # every step in ONE launch: the recurrence kernel forms the logits and the arg-max itself (persist.hip)
                pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
                toks = ops.greedy_decode(enc, pe, mask, h, emb, dp, hp, tgt_l, SOS_token)
                return self._cut(toks.t().cpu().numpy())
        if not (self.decode_graph and enc.is_cuda):
            pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
            tok = torch.full((B,), SOS_token, dtype=torch.int64, device=dev)
            dp, hp, emb = dec.dec_params(), dec.head_params(), dec.embedding.weight
            prep = ops.decode_prepare(emb, dp)
            for di in range(tgt_l):
                h, c, e, _ = ops.decode_step(enc, pe, mask, 1, tok, h, emb, dp, prep)
                _, tok = ops.head_logp_step(h, c, e, hp, want_argmax=True)
                toks[di] = tok
            return self._cut(toks.t().cpu().numpy())
        CH = self.DECODE_CHUNK
        st, dp, hp, emb = self._decode_state("greedy", enc, mask, 1, tgt_l)
        st["h"].copy_(h)
        st["tok"].fill_(SOS_token)
        if st["graph"] is None:
            st["chunk"] = torch.empty(CH, B, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with _lib.capture(g, pool=self._decode_pool()):
                hc, tc = st["h"], st["tok"]
                for i in range(CH):
                    tin = tc
                    if st["hoisted"]:
                        hc, c, e, _ = ops.decode_step_h(st["pe"], st["mask"], st["keys"], 1, tc, hc, emb, dp, st["prep"],
                                                        tables=st.get("tables"))
                    else:
                        hc, c, e, _ = ops.decode_step(st["enc"], st["pe"], st["mask"], 1, tc, hc, emb, dp, st["prep"])
                    _, tc = ops.head_logp_step(hc, c, e, hp, want_argmax=True, argmax_out=st["chunk"][i], hoisted=st["hoisted"],
                                               tables=st.get("tables") if st["hoisted"] else None, tok=tin)
                st["h"].copy_(hc)
                st["tok"].copy_(tc)
            st["graph"] = g
        for d0 in range(0, tgt_l, CH):
            st["graph"].replay()
            n = min(CH, tgt_l - d0)
            toks[d0:d0 + n].copy_(st["chunk"][:n])
        return self._cut(toks.t().cpu().numpy())

    def _beam(self, enc, mask, h, beam_size, max_length):
        """Batched beam search (V11.py:233-337): avoid_double=True, avoid_unk=False."""
        dec = self.decoder
        B, k = enc.shape[0], beam_size
        H = h.shape[1]
        V = dec.out.bias.shape[0]
        dev = enc.device
        graphed = self.decode_graph and enc.is_cuda
        st = None
        if graphed:
            st, dp, hp, emb = self._decode_state("beam", enc, mask, k, max_length)
            enc_s, pe, mask_s, prep = st["enc"], st["pe"], st["mask"], st["prep"]
            hoisted, keys, tables = st["hoisted"], st.get("keys"), st.get("tables")
        if st is not None and "beam" in st:
            # the search state of this decode shape lives in ONE buffer the captured graph points into: history (words |
            # back-pointers), running scores, the alive counter and the device-side step index -- one fill per call instead of a
            # fresh tensor and a copy each (22 small copy / fill launches per call before)
            st["flat"].zero_()
            beam, nll, n_alive, scratch = st["beam"], st["nll"], st["n_alive"], st["scratch"]
        else:
            nb = 2 * max_length * B * k
            flat = torch.zeros(nb + (B * k + 8 + 1) // 2, dtype=torch.int64, device=dev)
            beam = flat[:nb].view(2 * max_length, B, k)                                  # words | back-pointers
            tail = flat[nb:].view(torch.int32)
            nll = tail[:B * k].view(torch.float32).view(B, k)
            n_alive = tail[B * k:B * k + 1]
            di_state = tail[B * k + 2:B * k + 4]
            scratch = torch.empty(_lib.lib().vag_beam_scratch_bytes(B, k, V, max_length), dtype=torch.uint8, device=dev)
            if st is not None:
                st["flat"], st["beam"], st["nll"], st["n_alive"], st["scratch"], st["di"] = flat, beam, nll, n_alive, scratch, di_state
                st["one"] = torch.ones(1, dtype=torch.int32, device=dev)
        tok = torch.full((B,), SOS_token, dtype=torch.int64, device=dev)
        if not graphed:
            pe = ops.KeysProj.apply(enc, dec.attn.attn_e.weight)
            dp, hp, emb = dec.dec_params(), dec.head_params(), dec.embedding.weight
            prep = ops.decode_prepare(emb, dp)
            enc_s, mask_s = enc, mask
            hoisted = self.decode_hoisted and ops.decode_hoisted_ok(B * k, emb, dp, hp)
            keys = ops.decode_keys(enc, prep, hp) if hoisted else None
            tables = ops.decode_tables(emb, dp, hp) if hoisted else None
        h_next = st["h"] if graphed else torch.empty(B * k, H, dtype=torch.float32, device=dev)
        steps = 0
        for di in range(max_length):
            rps = 1 if di == 0 else k
            if hoisted:
                h, c, e, _ = ops.decode_step_h(pe, mask_s, keys, rps, tok, h, emb, dp, prep, tables=tables)
            else:
                h, c, e, _ = ops.decode_step(enc_s, pe, mask_s, rps, tok, h, emb, dp, prep)
            logp, _ = ops.head_logp_step(h, c, e, hp, hoisted=hoisted, tables=tables if hoisted else None, tok=tok)
            call("vag_beam_step", ptr(logp), logp.shape[1], ptr(nll), ptr(beam, torch.int64), di, max_length, ptr(h),
                 ptr(h_next), B, k, V, H, ptr(n_alive, torch.int32), scratch.data_ptr(), stream())
            steps = di + 1
            if graphed:
                break                                  # step 0 only (one hypothesis per sentence); the rest is replayed
            h, h_next = h_next, torch.empty(B * k, H, dtype=torch.float32, device=dev)
            tok = beam[di].view(-1)
            # the reference stops once every hypothesis has emitted EOS (V11.py:266-269); running on is harmless
            # (finished hypotheses only re-emit EOS at cost 0), so the device counter is polled only now and then.
            if di % 8 == 7 and int(n_alive.item()) == 0:
                break
        if graphed and max_length > 1:
            CH = self.DECODE_CHUNK
            st["tok"].copy_(beam[0].view(-1))
            st["di"][0:1].copy_(st["one"])              # the replayed steps start at step 1 (device to device: no host wait)
            if st["graph"] is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # raw logits + the pieces of their rows' log-sum-exp where the vocabulary product provides them: the beam
                # expansion normalises on the fly, no pass over the (B k, V) logits in between
                nparts = ops.head_logits_parts_count(hp, B * k, emb.shape[1], V) if self.decode_raw_logits else 0
                with _lib.capture(g, pool=self._decode_pool()):
                    for _ in range(CH):
                        if hoisted:
                            h2, c, e, _ = ops.decode_step_h(pe, mask_s, keys, k, st["tok"], st["h"], emb, dp, prep, tables=tables)
                        else:
                            h2, c, e, _ = ops.decode_step(enc_s, pe, mask_s, k, st["tok"], st["h"], emb, dp, prep)
                        if nparts > 0:
                            logits, parts = ops.head_logits_step(h2, c, e, hp, nparts, hoisted=hoisted,
                                                                 tables=tables if hoisted else None, tok=st["tok"])
                            call("vag_beam_step_logits_dev", ptr(logits), logits.shape[1], ptr(parts), nparts, ptr(nll),
                                 ptr(beam, torch.int64), ptr(st["di"], torch.int32), max_length, ptr(h2), ptr(st["h"]),
                                 ptr(st["tok"], torch.int64), B, k, V, H, ptr(n_alive, torch.int32), scratch.data_ptr(), stream())
                            continue
                        logp, _ = ops.head_logp_step(h2, c, e, hp, hoisted=hoisted, tables=tables if hoisted else None, tok=st["tok"])
                        call("vag_beam_step_dev", ptr(logp), logp.shape[1], ptr(nll), ptr(beam, torch.int64),
                             ptr(st["di"], torch.int32), max_length, ptr(h2), ptr(st["h"]), ptr(st["tok"], torch.int64),
                             B, k, V, H, ptr(n_alive, torch.int32), scratch.data_ptr(), stream())
                st["graph"] = g
            while steps < max_length:
                st["graph"].replay()
                steps = min(steps + CH, max_length)
                if int(n_alive.item()) == 0:           # V11.py:266-269, polled once per chunk
                    break
        out = torch.empty(B, max_length, dtype=torch.int64, device=dev)
        best = torch.empty(B, dtype=torch.float32, device=dev)
        call("vag_beam_finish", ptr(nll), ptr(beam, torch.int64), max_length, steps, B, k, ptr(out, torch.int64), ptr(best),
             stream())
        self.last_beam_scores = best
        self.last_decode_steps = steps            # decoder steps actually run (bench.py prices one step)
        return self._cut(out.cpu().numpy())

    def _validate_args(self, src_var, tgt_var, max_length):
        """(batch_size, tgt_l) as the reference computes them (models/...V11.py:170-177, NMT_Seq2Seq_Beam_V2.py:115-122)."""
        return src_var.size()[0], (max_length if tgt_var is None else tgt_var.size()[1])

    def beamsearch(self, encoder_outputs, context_mask, decoder_input, decoder_hidden, beam_size, max_length, avoid_double=True,
                   avoid_unk=False):
        """The reference's public entry to the batched beam search (models/...V11.py:233-337, NMT_Seq2Seq_Beam_V2.py:173-277), with
        ITS argument layout: encoder_outputs (Ts, B, 2H) time-major, context_mask (Ts, B), decoder_input (B, 1) = SOS,
        decoder_hidden (1, B, H).  Returns the list of token lists cut at EOS.  Only the reference's defaults are implemented
        (avoid_double=True: EOS hypotheses only continue with EOS; avoid_unk=False); no entry script passes anything else."""
        if not avoid_double or avoid_unk:
            raise NotImplementedError("beamsearch: only avoid_double=True, avoid_unk=False (the reference's defaults) run on the HIP path")
        if decoder_input is not None and not bool((decoder_input == SOS_token).all()):
            raise ValueError("beamsearch starts every hypothesis from SOS (models/...V11.py:186-188)")
        enc = encoder_outputs.transpose(0, 1).contiguous()
        mask = context_mask.transpose(0, 1).contiguous().to(enc.dtype)
        h = decoder_hidden.reshape(-1, decoder_hidden.shape[-1]).contiguous()
        with torch.no_grad():
            return self._beam(enc, mask, h, int(beam_size), int(max_length))

    @staticmethod
    def _cut(hyps):
        final = []
        for row in hyps:
            cur = []
            for t in row:
                if t == EOS_token:
                    break
                cur.append(t)
            final.append(cur)
        return final
