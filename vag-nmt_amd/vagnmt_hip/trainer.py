"""Step driver: the MI355X-native counterpart of the reference's train.py (train_imagine_beam / train_nmt).

    model.train(); zero_grad(); loss = model(...); loss.backward(); clip_grad_norm_(params, clip); optimizer.step()

with these changes in mechanism, none in arithmetic:
  * every parameter lives in ONE flat fp32 buffer laid out by the reference's Adam param groups
    (nmt_multimodal_beam_DE.py:303-332: names without 'bias' get L2 weight decay, names with 'bias' do not; optional
    half-learning-rate groups for 'vse_imagine'), with a matching flat gradient buffer the HIP backward kernels
    accumulate into directly.  Inside the buffer the segments are ordered by the moment their gradients become final
    in backward: everything except the encoder first, the encoder's parameters last, so a data-parallel run
    all-reduces the first bucket while the encoder's backward recurrence is still running;
  * forward + backward of a mini-batch is ONE call into the library (vag_train_step) on one static workspace, captured
    once per batch shape into a HIP graph and replayed;
  * global-norm clipping and Adam are one fused pass over the flat buffer (vag_clip_adam_flat, three launches) that also
    leaves the gradient buffer zeroed for the next step, followed by the refresh of the derived weights;
and, for data parallelism (one process per GPU over torch.distributed, backend "nccl" = RCCL), bucketed sum all-reduces
of the flat gradient between the backward phases and the optimiser.  Clipping acts on the average of the ranks' gradients:
with equal batch sizes per rank (what data.shard_batches hands out; bucket remainders of different size on different ranks
would make it a mean of per-rank means) that is a single-GPU step on the global batch's mean translation-loss gradient,
with the ranking loss taken per shard."""
import collections
import ctypes as C
import random

import torch

from ._lib import call, ptr, stream, capture as _capture


def param_groups(named_params, vse_separate=False):
    """The reference's optimiser grouping (nmt_multimodal_beam_DE.py:303-329) as (name, [param names], wd?, lr_mult)."""
    names = [n for n, p in named_params if p.requires_grad]
    if not vse_separate:
        return [("weight", [n for n in names if "bias" not in n], True, 1.0),
                ("bias", [n for n in names if "bias" in n], False, 1.0)]
    return [("mt_weight", [n for n in names if "bias" not in n and "vse_imagine" not in n], True, 1.0),
            ("mt_bias", [n for n in names if "bias" in n and "vse_imagine" not in n], False, 1.0),
            ("vse_weight", [n for n in names if "bias" not in n and "vse_imagine" in n], True, 0.5),
            ("vse_bias", [n for n in names if "bias" in n and "vse_imagine" in n], False, 0.5)]


def is_late(name):
    """Gradients that are only final after the encoder's backward recurrence (the last thing backward does)."""
    return name.startswith("encoder.")


def is_mid(name):
    """With three buckets (TrainStep(three_buckets=True)): gradients that become final in the SECOND half of the backward pass down
    to the encoder states -- the visual-grounding branch and the decoder's initial state (VSE_Imagine_Enc.py:110-152, V11.py:118) --
    while the head's, the decoder's and attn_e's are final after its first half."""
    return name.startswith("vse_imagine.") or name.startswith("decoderini.")


def flat_layout(named_params, vse_separate=False, groups=None, three_buckets=False):
    """Offsets of every parameter in the flat buffer.  Each optimiser group is split into an early and a late part
    (see is_late); all early segments come first.  Segments are contiguous, slots 256-byte aligned.
    Returns (segments [(name, [param names], wd?, lr_mult)], offsets dict, segment boundaries, total floats);
    the early bucket is [0, boundary of the first late segment).  Pure host logic (no GPU needed).
    ``groups``: an explicit grouping [(name, [param names], wd, lr_mult)] instead of the reference's two / four groups by
    name (the ``train`` shim passes what the caller's torch.optim.Adam holds; ``wd`` is then the group's own decay)."""
    byname = dict(named_params)
    early, mid, late = [], [], []
    for gname, names, wd, mult in (groups if groups is not None else param_groups(named_params, vse_separate)):
        e = [n for n in names if not is_late(n) and not (three_buckets and is_mid(n))]
        mi = [n for n in names if three_buckets and is_mid(n)]
        la = [n for n in names if is_late(n)]
        if e:
            early.append((gname, e, wd, mult))
        if mi:
            mid.append((gname + "/vse+init", mi, wd, mult))
        if la:
            late.append((gname + "/encoder", la, wd, mult))
    segs = early + mid + late
    offs, seg_off, o = {}, [0], 0
    for _, names, _, _ in segs:
        for n in names:
            offs[n] = o
            o += (byname[n].numel() + 63) // 64 * 64          # 256-byte slots: matrix rows start on cache-line boundaries
        seg_off.append(o)
    return segs, offs, seg_off, o


def _param_pickle_state():
    """``__getstate__`` of a re-homed parameter: nothing.  ``torch.save(model)`` (the reference checkpoints whole modules,
    nmt_multimodal_beam_DE.py:491-520) would otherwise pickle the parameter's ``_vag_grad`` view -- the whole flat gradient buffer --
    and a model loaded from such a pickle would route its gradients into that dead copy (vagnmt_hip.ops accumulates into
    ``_vag_grad`` and hands autograd nothing)."""
    return {}


class FlatParams:
    """Re-homes a module's parameters into one flat buffer (+ gradient, Adam m/v buffers)."""

    def __init__(self, model, vse_separate=False, groups=None, three_buckets=False, pad_to=1):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]      # tied weights appear once
        dev = named[0][1].device
        self.groups, self.offsets, self.seg_off, self.n = flat_layout(named, vse_separate, groups, three_buckets)
        # pad_to (the sharded optimiser: world_size x 64): the four buffers are ALLOCATED to a multiple of it so that they cut into
        # equal shards for reduce-scatter / all-gather; self.n stays the number of floats that belong to parameters
        self.n_alloc = (self.n + pad_to - 1) // pad_to * pad_to
        n_early = sum(1 for g in self.groups if not g[0].endswith("/encoder"))
        self.early_end = self.seg_off[n_early]              # [0, early_end): final before the encoder's backward
        n_first = sum(1 for g in self.groups if not g[0].endswith("/encoder") and not g[0].endswith("/vse+init"))
        self.first_end = self.seg_off[n_first]              # three buckets: [0, first_end) is final after the decoder's backward
        self._alloc = [torch.zeros(self.n_alloc, dtype=torch.float32, device=dev) for _ in range(4)]
        self.flat, self.grad, self.m, self.v = [t[:self.n] for t in self._alloc]
        self.flat_alloc, self.grad_alloc, self.m_alloc, self.v_alloc = self._alloc
        with torch.no_grad():
            for n, p in named:
                k, o = p.numel(), self.offsets[n]
                view = self.flat[o:o + k].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._vag_grad = self.grad[o:o + k].view_as(p)
                p.__getstate__ = _param_pickle_state          # (pickles carry the values only)
                p.grad = p._vag_grad
        self.named = named

    def buckets(self):
        """Gradient buckets in the order backward finishes them."""
        if self.first_end != self.early_end:
            return [(0, self.first_end), (self.first_end, self.early_end), (self.early_end, self.n)]
        return [(0, self.early_end), (self.early_end, self.n)]


class TrainStep:
    """One optimiser step per call.  ``step(src, lengths, tgt, im)`` returns (loss, loss_mt, loss_vse) as device
    tensors of this step (no host sync); call ``.item()`` on them only when a number is needed (the reference syncs
    every step, train.py:51)."""

    def __init__(self, model, criterion_mt, criterion_vse=None, lr=4e-4, weight_decay=1e-5, clip=1.0,
                 teacher_force_ratio=0.8, betas=(0.9, 0.999), eps=1e-8, vse_separate=False, use_graph=True,
                 process_group=None, world_size=1, max_graphs=256, pad_src=4, fused=None, backend=None,
                 force_phased=False, storage="f32", comm=None, groups=None, capture_after=1, three_buckets=False, zero1=False):
        self.model = model
        self.criterion_mt = criterion_mt
        self.criterion_vse = criterion_vse
        self.multimodal = hasattr(model, "vse_imagine")
        self.lr, self.wd, self.clip = lr, weight_decay, clip
        self.tfr = teacher_force_ratio
        self.betas, self.eps = betas, eps
        self.use_graph = use_graph
        self.pg, self.world = process_group, world_size
        self.comm_enabled = True              # False: keep the phases but skip the all-reduces (measurement only)
        # comm: a vagnmt_hip.comm.Comm (RCCL through the C ABI's vag_comm_*) instead of torch.distributed's all_reduce on
        # process_group; the sequence of phases and buckets is the same
        self.comm = comm
        self.force_phased = force_phased      # tests: the data-parallel sequence (two phases, two buckets) at world_size 1
        self.max_graphs = max_graphs
        # eager visits of a (shape, phases) key before it is captured: replay and eager launches take the same time while the GPU
        # is the bottleneck (measured: 3.284 vs 3.292 ms at configs[1], 1.91 vs 2.02 ms per step on a bucketed stream), a capture
        # costs ~1 ms, and a bucketed epoch holds ~200 keys of which many are met once or twice: shapes met once never pay for one (capture_after = 1: the second visit captures)
        self.capture_after = max(1, int(capture_after))
        self.pad_src = max(1, int(pad_src))
        # three_buckets: a third cut of the flat gradient after the decoder's backward (31.6 of bucket 0's 45.6 MB at configs[1] start
        # their all-reduce ~0.15 ms earlier, behind the VSE / initial-state backward and the encoder's); a switch for the first
        # multi-GPU session to A/B (what it buys depends on link bandwidth): the default stays two buckets
        # zero1 (SURVEY 8e / section 5 option for train.py:46-49): reduce-scatter of the flat gradient, each rank clips and updates ONE
        # contiguous shard (vag_clip_adam_shard), the updated shards come back by all-gather.  Optimiser traffic per rank drops from
        # 7 x 4 B per parameter to 1/world of it, the exchange moves the same bytes as the all-reduce it replaces but its second half
        # (the all-gather) can no longer hide behind the backward pass: an option for models whose optimiser pass outweighs that; at
        # configs[1] (16 M parameters: 72 us of Adam) the replicated default is expected to win.  Off by default.
        self.zero1 = bool(zero1)
        if self.zero1 and comm is not None:
            raise ValueError("zero1 uses torch.distributed's reduce-scatter / all-gather (process_group), not a vag Comm")
        self.fp = FlatParams(model, vse_separate, groups, three_buckets, pad_to=max(1, world_size) * 64 if self.zero1 else 1)
        dev = self.fp.flat.device
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.fp.flat, src=0, group=process_group)       # identical replicas
        ns = len(self.fp.groups)
        self._seg_off = (C.c_int64 * (ns + 1))(*self.fp.seg_off)
        # g[2]: True / False = the driver's weight_decay or none (the reference's grouping by name), a number = that group's own
        self._seg_wd = (C.c_float * ns)(*[(weight_decay if g[2] else 0.0) if isinstance(g[2], bool) else float(g[2])
                                          for g in self.fp.groups])
        self._seg_lr = (C.c_float * ns)(*[g[3] for g in self.fp.groups])       # relative rates; the rate itself is on the device
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(512, dtype=torch.float32, device=dev)        # VAG_ADAM_SCRATCH_BYTES, zero once
        self._graphs = collections.OrderedDict()      # LRU: (B,Ts,Tt,teacher) -> dict(graphs, generation)
        self._seen = collections.OrderedDict()        # shapes run once eagerly (captured on the second visit)
        self._opt_graphs = {}
        self.stats = {"captures": 0, "evictions": 0, "eager_steps": 0, "replays": 0}
        # the compute back end: the fused HIP step when the criteria are the reference's own, else the per-operator
        # autograd path; tests inject a CPU stand-in to exercise the data-parallel bookkeeping without a GPU
        self.backend = backend
        if backend is None and criterion_mt is not None and dev.type == "cuda":
            from .fused import FusedStep, fusable
            if (fused is None or fused) and fusable(model, criterion_mt, criterion_vse):
                self.backend = _FusedBackend(self, FusedStep(model, criterion_mt, criterion_vse, storage=storage))
            else:
                if storage != "f32":
                    raise ValueError("fp16 storage is a mode of the fused step (vag_train_step)")
                self.backend = _AutogradBackend(self)

    def set_lr(self, lr):
        """ReduceLROnPlateau equivalent hook (nmt_multimodal_beam_DE.py:335,469).  The rate lives in a device word the
        optimiser kernels read, so captured graphs serve every rate (a per-step schedule costs one 4-byte fill per step,
        no re-capture)."""
        if float(lr) != self.lr:
            self.lr = float(lr)
            self._lr_dev.fill_(self.lr)
        if self.fp.flat.is_cuda and self.backend is not None and getattr(self.backend, "with_optimizer", False):
            self.check()                  # a validation point: the loop is synchronising anyway (it just read the dev loss)

    def retune(self, seg_lr=None, seg_wd=None, clip=None, betas=None, eps=None):
        """Change what the captured graphs hold BY VALUE: the segments' relative learning rates and weight decays (one entry per
        segment of ``self.fp.groups``), the clip norm, Adam's betas / eps.  Returns True when something changed; the captured
        graphs are dropped then (a shape is captured again on its next visit).  The learning rate itself is a device word:
        ``set_lr``."""
        ns = len(self.fp.groups)
        changed = False
        for arr, new in ((self._seg_lr, seg_lr), (self._seg_wd, seg_wd)):
            if new is None:
                continue
            if len(new) != ns:
                raise ValueError("one value per segment (%d), got %d" % (ns, len(new)))
            for i, x in enumerate(new):
                if C.c_float(float(x)).value != arr[i]:
                    arr[i] = float(x)
                    changed = True
        if clip is not None and float(clip) != float(self.clip):
            self.clip, changed = float(clip), True
        if betas is not None and tuple(betas) != tuple(self.betas):
            self.betas, changed = (float(betas[0]), float(betas[1])), True
        if eps is not None and float(eps) != float(self.eps):
            self.eps, changed = float(eps), True
        if changed:
            self._graphs.clear()
            self._opt_graphs.clear()
        return changed

    # ---- optimiser + collectives ----
    def _optimizer(self):
        if hasattr(self.backend, "optimizer"):
            return self.backend.optimizer()
        fp = self.fp
        ns = len(fp.groups)
        call("vag_clip_adam_flat", ptr(fp.flat), ptr(fp.grad), ptr(fp.m), ptr(fp.v), fp.n, ns, self._seg_off, self._seg_lr,
             self._seg_wd, float(self.clip), 1.0 / self.world, self.betas[0], self.betas[1], self.eps, 1,
             ptr(self.step_count, torch.int32), ptr(self.grad_norm), self._scratch.data_ptr(), ptr(self._lr_dev), stream())
        if hasattr(self.backend, "after_optimizer"):
            self.backend.after_optimizer()

    def _shard(self):
        """[lo, hi) of this rank's shard of the flat buffers, and the (equal, padded) shard length."""
        import torch.distributed as dist
        rank = dist.get_rank(self.pg) if (self.world > 1 and dist.is_initialized()) else 0
        size = self.fp.n_alloc // max(1, self.world)
        lo = min(rank * size, self.fp.n)
        return lo, min(lo + size, self.fp.n), size, rank

    def _zero1_optimizer(self):
        """Reduce-scatter -> sharded sum of squares -> all-reduce of one double -> clip + Adam on the shard -> all-gather."""
        import torch.distributed as dist
        fp = self.fp
        lo, hi, size, rank = self._shard()
        multi = self.world > 1 and dist.is_initialized()
        if multi and self.comm_enabled:
            own = fp.grad_alloc[rank * size:(rank + 1) * size]
            try:
                dist.reduce_scatter_tensor(own, fp.grad_alloc, op=dist.ReduceOp.SUM, group=self.pg)      # RCCL: in place
            except (RuntimeError, NotImplementedError):
                dist.all_reduce(fp.grad_alloc, op=dist.ReduceOp.SUM, group=self.pg)      # (gloo has no reduce-scatter: same sums)
        if not hasattr(self, "_sumsq"):
            self._sumsq = torch.zeros(1, dtype=torch.float64, device=fp.flat.device)
        ns = len(fp.groups)
        args = (ptr(fp.flat), ptr(fp.grad), ptr(fp.m), ptr(fp.v), fp.n, ns, self._seg_off, self._seg_lr, self._seg_wd,
                float(self.clip), 1.0 / self.world, self.betas[0], self.betas[1], self.eps, 1, ptr(self.step_count, torch.int32),
                ptr(self.grad_norm), self._scratch.data_ptr(), ptr(self._lr_dev), lo, hi)
        call("vag_clip_adam_shard", *args, 0, ptr(self._sumsq, torch.float64), stream())
        if multi and self.comm_enabled:
            dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.pg)
        call("vag_clip_adam_shard", *args, 1, ptr(self._sumsq, torch.float64), stream())
        if multi and self.comm_enabled:
            own = fp.flat_alloc[rank * size:(rank + 1) * size]
            try:
                dist.all_gather_into_tensor(fp.flat_alloc, own, group=self.pg)
            except (RuntimeError, NotImplementedError):
                parts = [torch.empty_like(own) for _ in range(self.world)]
                dist.all_gather(parts, own.clone(), group=self.pg)
                fp.flat_alloc.copy_(torch.cat(parts))
        if hasattr(self.backend, "after_optimizer"):
            self.backend.after_optimizer()

    def gather_optimizer_state(self):
        """zero1: Adam's moments are current on the owning rank's shard only; make every rank's m / v whole (checkpointing)."""
        import torch.distributed as dist
        if not (self.zero1 and self.world > 1 and dist.is_initialized()):
            return
        _, _, size, rank = self._shard()
        for buf in (self.fp.m_alloc, self.fp.v_alloc):
            parts = [torch.empty(size, dtype=buf.dtype, device=buf.device) for _ in range(self.world)]
            dist.all_gather(parts, buf[rank * size:(rank + 1) * size].clone(), group=self.pg)
            buf.copy_(torch.cat(parts))

    def _run_optimizer(self):
        """Clip + Adam (+ derived-weight refresh) replayed from a small graph of its own (one graph: the rate is a device word)."""
        self.model._vag_weights_version = getattr(self.model, "_vag_weights_version", 0) + 1      # (see step())
        if self.zero1 and not hasattr(self.backend, "optimizer"):
            return self._zero1_optimizer()        # (eager: two small launches around an all-reduce of one double)
        if not (self.use_graph and self.fp.flat.is_cuda) or hasattr(self.backend, "optimizer"):
            return self._optimizer()
        key = ("opt",)
        g = self._opt_graphs.get(key)
        if g is None:
            if key not in self._seen:
                self._seen[key] = True
                return self._optimizer()
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                self._optimizer()
            self._opt_graphs[key] = g
        g.replay()

    def _allreduce_async(self, lo, hi):
        import torch.distributed as dist
        if not self.comm_enabled:
            return _NoWork()
        if self.comm is not None:
            return self.comm.all_reduce(self.fp.grad[lo:hi])
        return dist.all_reduce(self.fp.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def resync(self):
        """Make the replicas identical again (rank 0's parameters and optimiser state), e.g. after a measurement that ran
        steps without the all-reduces."""
        if self.world > 1:
            import torch.distributed as dist
            for t in (self.fp.flat, self.fp.m, self.fp.v):
                dist.broadcast(t, src=0, group=self.pg)
            if hasattr(self.backend, "after_optimizer"):
                self.backend.after_optimizer()

    SKIPPED_OFFSET = 28          # VAG_ADAM_SCRATCH_SKIPPED_OFFSET (include/vag_nmt.h), bytes into the optimiser scratch
    GUARD_OFFSET = 32            # VAG_ADAM_SCRATCH_GUARD_OFFSET: this driver's {void flag, give-up count} (vag_step_cfg.guard)

    def guard_ptr(self):
        """Device address of this driver's guard pair: its persistent recurrence launches report a give-up there, its
        optimiser kernels read it there -- another driver's (or a decoder's) give-up on the same device never voids a step here."""
        return self._scratch.data_ptr() + self.GUARD_OFFSET if self._scratch.is_cuda else None

    def skipped_steps(self):
        """Optimiser steps the device refused to apply so far (non-finite gradient norm, or a persistent recurrence kernel
        gave up a wait: the parameters, moments and step counter were left alone, ``grad_norm`` reads NaN for such a
        step).  One 4-byte read = one device sync; 0 in a healthy run."""
        if not self._scratch.is_cuda:
            return 0
        return int(self._scratch.view(torch.int32)[self.SKIPPED_OFFSET // 4].item())

    def check(self, raise_on_skip=True):
        """Synchronises and raises if steps were skipped or a persistent recurrence kernel had to give up a wait since the
        last check (its workgroups were not all resident, e.g. another process or a collective's kernels on the same GPU:
        the results of those launches are void; the optimiser skipped those steps on the device, so the weights are
        intact).  Called by ``set_lr`` (the validation point of the reference's loop) and by ``save_checkpoint``."""
        from ._lib import lib, VagError
        n = sk = 0
        if self.fp.flat.is_cuda:
            w = self._scratch.view(torch.int32)
            sk, _, n = w[self.SKIPPED_OFFSET // 4:self.GUARD_OFFSET // 4 + 2].tolist()      # one read: skipped, flag, give-ups
            if n:
                w[self.GUARD_OFFSET // 4 + 1:self.GUARD_OFFSET // 4 + 2].zero_()
            self.process_timeouts = lib().vag_persistent_timeouts()      # every driver's and every unguarded launch's; read + reset
        new_sk, self._skipped_seen = sk - getattr(self, "_skipped_seen", 0), sk
        f = getattr(self.backend, "f", None)
        if f is not None and f.losses.is_cuda:
            dev_n = int(f.losses[3:4].view(torch.int32).item())
            if dev_n != (f.executed & 0x7fffffff):           # the result ring is addressed by this count on both sides
                raise VagError("result ring out of step: the device executed %d forward phases, the host counted %d"
                               % (dev_n, f.executed))
        if n != 0 or (raise_on_skip and new_sk != 0):
            raise VagError("%d optimiser step(s) skipped on the device (void gradient); persistent recurrence kernels of this "
                           "driver: %d waits gave up%s" % (new_sk, n, "; a GPU that is not this process's alone needs "
                                                           "set_option('persistent', 0)" if n else ""))

    # ---- public ----
    def step(self, src, lengths, tgt, im=None, teacher=None):
        self.model.train()
        # the optimiser kernels write the parameters through the flat buffer: torch's version counters do not see that.  What is
        # cached per set of weights (the decoding tables of models._seq2seq) looks at this counter as well.
        self.model._vag_weights_version = getattr(self.model, "_vag_weights_version", 0) + 1
        if teacher is None:
            teacher = random.random() < self.tfr                     # models/...V11.py:136
        if not torch.is_tensor(lengths):
            dt = getattr(lengths, "device_tensor", None)      # vagnmt_hip.data.LengthList: already on the device
            lengths = dt if dt is not None else torch.tensor(list(lengths), dtype=torch.int32, device=src.device)
        lengths = lengths.to(torch.int32)
        be = self.backend
        if be is None:
            raise RuntimeError("TrainStep needs HIP tensors and criteria (or an injected backend) to run a step")
        bks = self.fp.buckets()
        if self.zero1 and not hasattr(be, "optimizer"):
            # the whole backward, then the sharded optimiser (its reduce-scatter needs every bucket final)
            be.run(src, lengths, tgt, im, teacher, 7)
            out = be.outputs()
            self._run_optimizer()
            return out
        if (self.world > 1 or (self.force_phased and (self.pg is not None or self.comm is not None))) and \
                getattr(be, "phased", False):
            if len(bks) == 3:
                # three cuts (vag_train_step phases 16 / 32: the two halves of phase 2)
                be.run(src, lengths, tgt, im, teacher, 1 | 16)
                w0 = self._allreduce_async(*bks[0])
                be.run(src, lengths, tgt, im, teacher, 32, reuse=True)
                wm = self._allreduce_async(*bks[1])
                be.run(src, lengths, tgt, im, teacher, 4, reuse=True)
                w1 = self._allreduce_async(*bks[2])
                w0.wait()
                wm.wait()
                w1.wait()
            else:
                # backward in two phases; the first bucket's all-reduce runs beside the encoder's backward
                be.run(src, lengths, tgt, im, teacher, 3)
                w0 = self._allreduce_async(*bks[0])
                be.run(src, lengths, tgt, im, teacher, 4, reuse=True)
                w1 = self._allreduce_async(*bks[1])
                w0.wait()
                w1.wait()
        elif self.world == 1 and getattr(be, "with_optimizer", False) and self.use_graph:
            # single GPU: forward, backward and the optimiser in ONE captured graph per shape
            be.run(src, lengths, tgt, im, teacher, 7, optimizer=True)
            return be.outputs()
        else:
            be.run(src, lengths, tgt, im, teacher, 7)
            if self.world > 1:
                w0 = self._allreduce_async(0, self.fp.n)
                w0.wait()
        out = be.outputs()
        self._run_optimizer()
        return out


class _NoWork:
    def wait(self):
        return True


class _FusedBackend:
    """vag_train_step replayed from HIP graphs: one LRU-bounded entry per (B, padded Ts, Tt, teacher) holding the captured
    phase graphs; all entries share the FusedStep's static workspace and input buffers."""
    phased = True
    with_optimizer = True

    def __init__(self, ts, fused):
        self.ts, self.f = ts, fused
        fused.guard = ts.guard_ptr()

    @property
    def generation(self):
        return self.f.generation

    def _pad(self, src, lengths):
        """Source length up to a multiple of pad_src: padded positions carry token 0 / mask 0 and are exact no-ops
        (zero attention weight, zero encoder state), so fewer distinct shapes need a captured graph."""
        B, Ts = src.shape
        p = self.ts.pad_src
        Tp = (Ts + p - 1) // p * p
        if Tp != Ts:
            src = torch.nn.functional.pad(src, (0, Tp - Ts))
        return src

    def run(self, src, lengths, tgt, im, teacher, phases, reuse=False, optimizer=False):
        ts, f = self.ts, self.f
        if not reuse:
            src = self._pad(src, lengths)
            B, Ts = src.shape
            Tt = tgt.shape[1]
            if f.reserve(B, Ts, Tt):
                ts._graphs.clear()                    # static buffers moved: every captured graph is stale
                ts._opt_graphs.clear()
            f.load_batch(src, lengths, tgt, im)
            self._cur = (B, Ts, Tt, bool(teacher))
        B, Ts, Tt, teacher = self._cur
        key = self._cur

        def launch():
            f.run(B, Ts, Tt, teacher, phases)
            if optimizer:
                ts._optimizer()
        if optimizer:
            key = key + ("opt",)
        if not ts.use_graph:
            ts.stats["eager_steps"] += 1
            return launch()
        ent = ts._graphs.get(key)
        if ent is not None:
            ts._graphs.move_to_end(key)
        if ent is None or phases not in ent:
            seen = ts._seen.get((key, phases), 0)
            if seen < ts.capture_after:
                # the first visits of a shape run eagerly (rare shapes never pay for a capture), a later visit captures
                ts._seen[(key, phases)] = seen + 1
                while len(ts._seen) > 8192:
                    ts._seen.popitem(last=False)
                ts.stats["eager_steps"] += 1
                return launch()
            if ent is None:
                ent = {}
                ts._graphs[key] = ent
                while len(ts._graphs) > ts.max_graphs:
                    ts._graphs.popitem(last=False)
                    ts.stats["evictions"] += 1
            g = torch.cuda.CUDAGraph()
            with _capture(g):
                launch()
            ent[phases] = g
            ts.stats["captures"] += 1
        ts.stats["replays"] += 1
        ent[phases].replay()
        if phases & 1:
            f.executed += 1               # (a capture does not execute; FusedStep.run counts its eager executions itself)

    def outputs(self):
        """(loss, loss_mt, loss_vse) of the step just enqueued: views of that step's slot in the result ring the forward phase
        writes (vag_step_cfg.loss_ring), valid for FusedStep.LOSS_RING further steps -- no copy launch per step."""
        f = self.f
        o = 4 + 4 * ((f.executed - 1) % f.LOSS_RING)
        return f.losses[o], f.losses[o + 1], f.losses[o + 2]

    def outputs_row(self):
        """The same three numbers as one contiguous 3-element view (one device-to-host copy reads them all)."""
        f = self.f
        o = 4 + 4 * ((f.executed - 1) % f.LOSS_RING)
        return f.losses[o:o + 3]

    def after_optimizer(self):
        self.f.refresh_derived()


class _AutogradBackend:
    """Per-operator path through torch.autograd (criteria other than the reference's own): eager, one all-reduce."""
    phased = False

    def __init__(self, ts):
        self.ts = ts
        self._out = None

    def run(self, src, lengths, tgt, im, teacher, phases, reuse=False):
        ts = self.ts
        tfr = 1.0 if teacher else 0.0       # the coin is drawn by the caller
        # the operators' persistent launches -- forward on this thread, backward on autograd's worker thread -- report to this
        # driver's guard pair, which its optimiser kernels read (ops._recurrence_call carries it into each call)
        from . import ops
        ops.set_operator_guard(ts.guard_ptr())
        try:
            self._run(src, lengths, tgt, im, tfr)
        finally:
            ops.set_operator_guard(None)

    def _run(self, src, lengths, tgt, im, tfr):
        ts = self.ts
        if ts.multimodal:
            loss, loss_mt, loss_vse = ts.model(src, lengths, tgt, im, tfr, criterion_mt=ts.criterion_mt,
                                               criterion_vse=ts.criterion_vse)
        else:
            loss = ts.model(src, lengths, tgt, tfr, criterion=ts.criterion_mt)
            loss_mt, loss_vse = loss, None
        loss.backward()
        zero = torch.zeros((), device=loss.device)
        self._out = (loss.detach(), loss_mt.detach(), loss_vse.detach() if torch.is_tensor(loss_vse) else zero)

    def outputs(self):
        return self._out
