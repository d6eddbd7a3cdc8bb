"""Shadow of the reference's step-driver module ``train`` (train.py), for the reference's own entry scripts:

    from train import *                                          (nmt_multimodal_beam_DE.py:16, nmt_monomodal_beam_DE.py:20)
    train_imagine_beam(batch_x, batch_y, batch_im, batch_x_lengths, model, optimizer, criterion_mt, criterion_vse,
                       loss_w, teacher_force_ratio, clip=clip)   (nmt_multimodal_beam_DE.py:394)
    train_nmt(batch_x, batch_y, batch_x_lengths, model, criterion, optimizer, teacher_force_ratio)   (monomodal :320)

Under ``python -m vagnmt_hip.run SCRIPT`` this directory sits ahead of the user's checkout on ``sys.path``, so the script's
``from train import *`` lands here.  Two functions are ours -- ``train_imagine_beam`` (train.py:36-51) and ``train_nmt``
(train.py:19-32), same signatures, same return values (Python floats) -- and run the step as ONE library call
(``vag_train_step`` + fused clip/Adam, replayed from a HIP graph per batch shape: vagnmt_hip.trainer.TrainStep).  Every other
name of the module (``random_sample_display``, ``train_imagine_beam_v2``, ``MAX_LENGTH``, ``SOS_token`` ...) is served from
the checkout's own ``train.py`` -- the one in the launched script's directory (``VAG_REFERENCE_CHECKOUT``), nowhere else;
nothing of it is copied here.

What the fused step takes from the caller's objects, every call:
  * the optimiser's param groups (nmt_multimodal_beam_DE.py:303-332): which parameters, each group's ``lr`` /
    ``weight_decay`` / ``betas`` / ``eps`` -- so ``ReduceLROnPlateau`` (:335,469) keeps working: the rate is one device word;
  * ``clip`` and ``teacher_force_ratio`` (one ``random.random()`` per step, as models/...V11.py:136 draws it);
  * the criteria (nmt_multimodal_beam_DE.py:291-299).
Adam's moments live in the step driver's flat buffers; ``optimizer.state`` exposes them per parameter as views
(``exp_avg`` / ``exp_avg_sq``) so that ``optimizer.state_dict()`` keeps meaning what it says.

What the fused step cannot serve runs the reference's literal sequence on the per-operator HIP path instead (the checkout's
own function when there is one): criteria other than the reference's, an optimiser that is not a plain ``torch.optim.Adam``
(amsgrad, maximize, per-group betas), parameters outside the optimiser, CPU tensors (which raise there: no CPU fallback)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

import torch as _torch

_HERE = _os.path.dirname(_os.path.abspath(__file__))


def _load_checkout_module():
    """The checkout's train.py: ``$VAG_REFERENCE_CHECKOUT/train.py`` -- the directory of the script ``python -m vagnmt_hip.run``
    launched (or what the user put there) -- and only if its text defines the reference's step functions.  sys.path and the working
    directory are NOT searched: an unrelated ``train.py`` lying around must never be executed at import time."""
    d = _os.environ.get("VAG_REFERENCE_CHECKOUT")
    if not d:
        return None
    d = _os.path.abspath(d)
    f = _os.path.join(d, "train.py")
    if _os.path.realpath(d) == _os.path.realpath(_HERE) or not _os.path.isfile(f):
        return None
    try:
        with open(f, errors="replace") as fh:
            text = fh.read()
    except OSError:
        return None
    if "def train_imagine_beam" not in text and "def train_nmt" not in text:
        import warnings
        warnings.warn("%s does not define train_imagine_beam / train_nmt: not the reference's train.py, ignored" % f)
        return None
    spec = _ilu.spec_from_file_location("_vag_checkout_train", f)
    mod = _ilu.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_checkout = _load_checkout_module()
if _checkout is not None:
    # what `from train import *` hands out: every public name of the checkout's module (it defines no __all__) ...
    globals().update({k: v for k, v in vars(_checkout).items() if not k.startswith("_")})
CLIP = globals().get("CLIP", 1.0)            # train.py:14, read by train_nmt

_CHECK_EVERY = 256                           # steps between TrainStep.check() calls (one extra 4-byte read)


class _Driver:
    """One fused step driver per (model, optimizer) pair, kept on the optimiser object (torch's Optimizer.__getstate__
    pickles defaults / state / param_groups only, so a pickled optimiser or ``torch.save(model)`` never sees it)."""

    def __init__(self, model, optimizer, criterion_mt, criterion_vse, clip, tfr):
        from vagnmt_hip.trainer import TrainStep
        byid = {id(p): n for n, p in model.named_parameters()}
        groups = []
        for i, g in enumerate(optimizer.param_groups):
            groups.append(("g%d" % i, [byid[id(p)] for p in g["params"]], float(g.get("weight_decay", 0.0)), 1.0))
        g0 = optimizer.param_groups[0]
        self.model, self.optimizer = model, optimizer
        self.criteria = (criterion_mt, criterion_vse)
        self.ts = TrainStep(model, criterion_mt, criterion_vse, lr=float(g0["lr"]), clip=float(clip),
                            teacher_force_ratio=float(tfr), betas=tuple(g0["betas"]), eps=float(g0["eps"]), groups=groups,
                            capture_after=2)      # a bucketed epoch meets many (B, Ts, Tt) shapes once or twice: those stay eager
        # segment -> optimiser group (the flat layout splits a group into its encoder / non-encoder parts)
        self.seg_group = [int(name.split("/")[0][1:]) for name, _, _, _ in self.ts.fp.groups]
        self.calls = 0
        self.n_groups = len(optimizer.param_groups)
        self.install_state_views(adopt=True)
        if hasattr(optimizer, "register_state_dict_pre_hook"):
            optimizer.register_state_dict_pre_hook(lambda opt: self.export_steps())

    def install_state_views(self, adopt):
        """Adam's moments under torch.optim.Adam's names, as views of the step driver's flat buffers.  ``adopt``: what the optimiser
        holds for a parameter right now (it stepped by itself before the first fused step, or ``optimizer.load_state_dict()`` put
        restored tensors there) is copied into the flat buffers first, and the largest restored ``step`` becomes the device's
        counter -- the fused step then continues from the restored moments instead of silently ignoring them."""
        fp, opt = self.ts.fp, self.optimizer
        steps = []
        with _torch.no_grad():
            for n, p in fp.named:
                o, k = fp.offsets[n], p.numel()
                mv, vv = fp.m[o:o + k].view_as(p), fp.v[o:o + k].view_as(p)
                st = opt.state.get(p)
                if adopt and st and "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    if "step" in st:
                        steps.append(int(float(st["step"])))
                opt.state[p] = {"step": _torch.zeros((), dtype=_torch.float32), "exp_avg": mv, "exp_avg_sq": vv}
        if steps:
            self.ts.step_count.fill_(max(steps))
            self.export_steps()

    def views_intact(self):
        """False once something (``optimizer.load_state_dict``) has replaced the installed views by other tensors."""
        fp = self.ts.fp
        n, p = fp.named[0]
        st = self.optimizer.state.get(p)
        return bool(st) and "exp_avg" in st and st["exp_avg"].data_ptr() == fp.m[fp.offsets[n]:].data_ptr()

    def export_steps(self):
        """The device's step counter into torch's per-parameter ``step`` entries (before ``optimizer.state_dict()``, and before
        a step torch.optim.Adam takes itself on the same state)."""
        t = float(int(self.ts.step_count.item()))
        for st in self.optimizer.state.values():
            if "step" in st:
                st["step"].fill_(t)

    def import_steps(self):
        """... and back, after torch.optim.Adam stepped: counter, derived weights, decode tables."""
        steps = [int(st["step"].item()) for st in self.optimizer.state.values() if "step" in st]
        if steps:
            self.ts.step_count.fill_(max(steps))
        self.model._vag_weights_version = getattr(self.model, "_vag_weights_version", 0) + 1
        if hasattr(self.ts.backend, "after_optimizer"):
            self.ts.backend.after_optimizer()

    def sync_hyper(self, clip, tfr):
        ts, pg = self.ts, self.optimizer.param_groups
        base = float(pg[0]["lr"])
        mult = [1.0 if base == 0.0 else float(g["lr"]) / base for g in pg]
        wd = [float(g.get("weight_decay", 0.0)) for g in pg]
        ts.retune(seg_lr=[mult[i] for i in self.seg_group], seg_wd=[wd[i] for i in self.seg_group], clip=float(clip),
                  betas=pg[0]["betas"], eps=pg[0]["eps"])
        if base != ts.lr:                     # ReduceLROnPlateau.step() wrote the groups' rates: one 4-byte fill
            ts.lr = base
            ts._lr_dev.fill_(base)
        ts.tfr = float(tfr)

    def step(self, src, lengths, tgt, im, clip, tfr):
        self.sync_hyper(clip, tfr)
        self.ts.step(src, lengths, tgt, im)
        vals = self.ts.backend.outputs_row().tolist()          # ONE device-to-host read (train.py:51: three .item() calls)
        self.calls += 1
        if self.calls % _CHECK_EVERY == 0:
            self.ts.check()
        return vals


def _plain_adam(optimizer):
    """torch.optim.Adam in the form the fused optimiser kernels implement: one (betas, eps) for all groups, L2 weight decay."""
    if type(optimizer) is not _torch.optim.Adam:
        return False
    pg = optimizer.param_groups
    b0, e0 = pg[0]["betas"], pg[0]["eps"]
    for g in pg:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or \
                g.get("decoupled_weight_decay") or g["betas"] != b0 or g["eps"] != e0 or _torch.is_tensor(g["lr"]):
            return False
    return True


def _covers(model, optimizer):
    """The optimiser steps exactly the model's trainable parameters (each once)."""
    want = sorted(id(p) for p in model.parameters() if p.requires_grad)
    have = sorted(id(p) for g in optimizer.param_groups for p in g["params"])
    return want == have


def _driver(model, optimizer, criterion_mt, criterion_vse, clip, tfr):
    """(the pair's fused driver or None when the fused step does not serve this call, the pair's existing driver or None)."""
    d = getattr(optimizer, "_vag_driver", None)
    if d is not None:
        same = d.model is model and d.criteria[0] is criterion_mt and d.criteria[1] is criterion_vse and \
            _plain_adam(optimizer) and len(optimizer.param_groups) == d.n_groups
        if same and not d.views_intact():     # optimizer.load_state_dict() since the last call: take the restored moments over
            d.install_state_views(adopt=True)
            d.model._vag_weights_version = getattr(d.model, "_vag_weights_version", 0) + 1
            if hasattr(d.ts.backend, "after_optimizer"):
                d.ts.backend.after_optimizer()      # a restore writes the weights too: what derives from them follows
        return (d if same else None), d
    from vagnmt_hip.fused import fusable
    p0 = next(model.parameters())
    if not (p0.is_cuda and fusable(model, criterion_mt, criterion_vse) and _plain_adam(optimizer) and
            _covers(model, optimizer)):
        return None, None
    # (an optimiser that has already stepped on its own, or was restored from a checkpoint: its moments and step count are adopted)
    d = _Driver(model, optimizer, criterion_mt, criterion_vse, clip, tfr)
    optimizer._vag_driver = d
    return d, d


def _unfused(existing, fn):
    """A step the fused driver does not serve, on an optimiser that may have one: torch.optim.Adam then steps on the driver's own
    moment buffers (optimizer.state holds views of them), so only the step counter and what derives from the weights cross.
    The parameters' ``_vag_grad`` routing (vagnmt_hip.ops: backward accumulates straight into the flat gradient buffer and hands
    autograd nothing) is lifted for the call, so that ``clip_grad_norm_`` and ``optimizer.step()`` see ordinary ``.grad``s."""
    if existing is None:
        return fn()
    existing.export_steps()
    routed = [(p, p._vag_grad) for _, p in existing.ts.fp.named if hasattr(p, "_vag_grad")]
    for p, _ in routed:
        del p._vag_grad
    try:
        out = fn()
    finally:
        for p, v in routed:
            p._vag_grad = v
    existing.import_steps()
    return out


def _literal_step(model, optimizer, clip, forward):
    """train.py:38-49 on the per-operator HIP path (what the checkout's function does; used when there is no checkout)."""
    model.train()
    optimizer.zero_grad()
    out = forward()
    loss = out[0] if isinstance(out, tuple) else out
    loss.backward()
    _torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    optimizer.step()
    return out


def _as_float(x):
    return float(x.item()) if _torch.is_tensor(x) else float(x)


def train_imagine_beam(input_variable, target_variable, im_variable, input_lengths, model, optimizer, criterion_mt,
                       criterion_vse, loss_weight, teacher_force_ratio, max_length=globals().get("MAX_LENGTH", 40), clip=1):
    """train.py:36-51.  ``loss_weight`` and ``max_length`` are as unused here as there (the model holds its ``loss_w``)."""
    d, existing = _driver(model, optimizer, criterion_mt, criterion_vse, clip, teacher_force_ratio)
    if d is not None:
        return tuple(d.step(input_variable, input_lengths, target_variable, im_variable, clip, teacher_force_ratio))
    if _checkout is not None and hasattr(_checkout, "train_imagine_beam"):
        return _unfused(existing, lambda: _checkout.train_imagine_beam(
            input_variable, target_variable, im_variable, input_lengths, model, optimizer, criterion_mt, criterion_vse,
            loss_weight, teacher_force_ratio, max_length, clip))
    out = _unfused(existing, lambda: _literal_step(model, optimizer, clip, lambda: model(
        input_variable, input_lengths, target_variable, im_variable, teacher_force_ratio, criterion_mt=criterion_mt,
        criterion_vse=criterion_vse)))
    return tuple(_as_float(x) for x in out)


def train_nmt(input_variable, target_variable, input_lengths, model, criterion, optimizer, teacher_force_ratio=0.5):
    """train.py:19-32 (the text-only model; clips at the module constant CLIP, train.py:14)."""
    d, existing = _driver(model, optimizer, criterion, None, CLIP, teacher_force_ratio)
    if d is not None:
        return d.step(input_variable, input_lengths, target_variable, None, CLIP, teacher_force_ratio)[0]
    if _checkout is not None and hasattr(_checkout, "train_nmt"):
        return _unfused(existing, lambda: _checkout.train_nmt(
            input_variable, target_variable, input_lengths, model, criterion, optimizer, teacher_force_ratio))
    out = _unfused(existing, lambda: _literal_step(model, optimizer, CLIP, lambda: model(
        input_variable, input_lengths, target_variable, teacher_force_ratio, criterion=criterion)))
    return _as_float(out)
