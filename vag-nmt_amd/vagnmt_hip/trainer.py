"""Step driver: the MI355X-native counterpart of the reference's train.py (train_imagine_beam / train_nmt).

    model.train(); zero_grad(); loss = model(...); loss.backward(); clip_grad_norm_(params, clip); optimizer.step()

with three changes in mechanism, none in arithmetic:
  * every parameter lives in ONE flat fp32 buffer ordered like the reference's Adam param groups
    (nmt_multimodal_beam_DE.py:303-332: names without 'bias' get L2 weight decay, names with 'bias' do not;
    optional half-learning-rate groups for 'vse_imagine'), with a matching flat gradient buffer the HIP backward
    kernels accumulate into directly;
  * global-norm clipping and Adam are one fused pass over that buffer (vag_clip_adam_flat);
  * zero-grad + forward + backward are captured once per batch shape into a HIP graph and replayed, so the ~1.5k
    kernel launches of a step cost one graph launch on the host;
and, for data parallelism (one process per GPU), one RCCL sum all-reduce of the flat gradient buffer between
backward and the optimiser (clipping acts on the averaged gradient, exactly what a single-GPU step on the global
batch's mean gradient would do)."""
import ctypes as C
import random

import torch

from ._lib import call, ptr, stream


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


def flat_layout(named_params, vse_separate=False):
    """Offsets of every parameter in the flat buffer: groups are contiguous segments, slots 16-byte aligned.
    Returns (groups, offsets dict, segment boundaries, total floats).  Pure host logic (no GPU needed)."""
    byname = dict(named_params)
    groups = param_groups(named_params, vse_separate)
    offs, seg_off, o = {}, [0], 0
    for _, names, _, _ in groups:
        for n in names:
            offs[n] = o
            o += (byname[n].numel() + 3) // 4 * 4
        seg_off.append(o)
    return groups, offs, seg_off, o


class FlatParams:
    """Re-homes a module's parameters into one flat buffer (+ gradient, Adam m/v buffers)."""

    def __init__(self, model, vse_separate=False):
        named = list(model.named_parameters())          # tied weights appear once
        dev = named[0][1].device
        self.groups, self.offsets, self.seg_off, self.n = flat_layout(named, vse_separate)
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in named:
                k, o = p.numel(), self.offsets[n]
                view = self.flat[o:o + k].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._vag_grad = self.grad[o:o + k].view_as(p)
                p.grad = p._vag_grad
        self.named = named


class TrainStep:
    """One optimiser step per call.  ``step(src, lengths, tgt, im)`` returns (loss, loss_mt, loss_vse) as device
    tensors (no host sync); call ``.item()`` on them only when a number is needed (the reference syncs every step,
    train.py:51)."""

    def __init__(self, model, criterion_mt, criterion_vse=None, lr=4e-4, weight_decay=1e-5, clip=1.0,
                 teacher_force_ratio=0.8, betas=(0.9, 0.999), eps=1e-8, vse_separate=False, use_graph=True,
                 process_group=None, world_size=1, overlap=False):
        self.model = model
        self.criterion_mt = criterion_mt
        self.criterion_vse = criterion_vse
        self.multimodal = hasattr(model, "vse_imagine")
        self.lr, self.wd, self.clip = lr, weight_decay, clip
        self.tfr = teacher_force_ratio
        self.betas, self.eps = betas, eps
        self.use_graph = use_graph
        self.pg, self.world = process_group, world_size
        self.fp = FlatParams(model, vse_separate)
        dev = self.fp.flat.device
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.fp.flat, src=0, group=process_group)       # identical replicas
        ns = len(self.fp.groups)
        self._seg_off = (C.c_int64 * (ns + 1))(*self.fp.seg_off)
        self._seg_wd = (C.c_float * ns)(*[weight_decay if g[2] else 0.0 for g in self.fp.groups])
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._scratch = torch.zeros(512, dtype=torch.float32, device=dev)        # VAG_ADAM_SCRATCH_BYTES
        self._graphs = {}
        self._eager_done = set()
        # optional second stream for the weight-gradient products.  Measured (round 1): no gain -- the products' blocks
        # fill every CU and the recurrence's small kernels queue behind them -- so it is off by default.
        self._side = torch.cuda.Stream(device=dev) if overlap and dev.type == "cuda" else None

    def set_lr(self, lr):
        """ReduceLROnPlateau equivalent hook (nmt_multimodal_beam_DE.py:335,469): lr is a host scalar per call."""
        self.lr = lr

    # ---- pieces ----
    def _fwd_bwd(self, src, lengths, tgt, im, teacher):
        self.fp.grad.zero_()
        tfr = 1.0 if teacher else 0.0       # the coin is drawn by the caller so each captured graph is one fixed path
        if self.multimodal:
            loss, loss_mt, loss_vse = self.model(src, lengths, tgt, im, tfr, criterion_mt=self.criterion_mt,
                                                 criterion_vse=self.criterion_vse)
        else:
            loss = self.model(src, lengths, tgt, tfr, criterion=self.criterion_mt)
            loss_mt, loss_vse = loss, None
        from . import ops
        if self._side is not None:
            ops.SIDE.enable(self._side)
        try:
            loss.backward()
        finally:
            if self._side is not None:
                ops.SIDE.disable()        # joins: the gradients are complete on the current stream after this
        return loss, loss_mt, loss_vse

    def _optimizer(self):
        fp = self.fp
        ns = len(fp.groups)
        seg_lr = (C.c_float * ns)(*[self.lr * g[3] for g in fp.groups])
        call("vag_clip_adam_flat", ptr(fp.flat), ptr(fp.grad), ptr(fp.m), ptr(fp.v), fp.n, ns, self._seg_off, seg_lr,
             self._seg_wd, float(self.clip), 1.0 / self.world, self.betas[0], self.betas[1], self.eps,
             ptr(self.step_count, torch.int32), ptr(self.grad_norm), self._scratch.data_ptr(), stream())

    def _allreduce(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM, group=self.pg)

    # ---- public ----
    def step(self, src, lengths, tgt, im=None, teacher=None):
        self.model.train()
        if teacher is None:
            teacher = random.random() < self.tfr                     # models/...V11.py:136
        if not torch.is_tensor(lengths):
            lengths = torch.tensor(list(lengths), dtype=torch.int32, device=src.device)
        key = (tuple(src.shape), tuple(tgt.shape), bool(teacher))
        if not self.use_graph:
            out = self._fwd_bwd(src, lengths, tgt, im, teacher)
        elif key not in self._graphs:
            if key not in self._eager_done:
                # first visit of a shape: run eagerly (loads code objects, sizes the allocator), capture on the next
                self._eager_done.add(key)
                out = self._fwd_bwd(src, lengths, tgt, im, teacher)
            else:
                out = self._capture(key, src, lengths, tgt, im, teacher)
        else:
            g = self._graphs[key]
            g["src"].copy_(src)
            g["len"].copy_(lengths)
            g["tgt"].copy_(tgt)
            if im is not None:
                g["im"].copy_(im)
            g["graph"].replay()
            out = g["out"]
        self._allreduce()
        self._optimizer()
        return out

    def _capture(self, key, src, lengths, tgt, im, teacher):
        st = {"src": src.clone(), "len": lengths.clone().to(torch.int32), "tgt": tgt.clone(),
              "im": im.clone() if im is not None else None}
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            out = self._fwd_bwd(st["src"], st["len"], st["tgt"], st["im"], teacher)
        st["graph"], st["out"] = graph, out
        self._graphs[key] = st
        graph.replay()
        return out
