"""Checkpoint format.

The reference pickles the whole module (``torch.save(model)``, nmt_multimodal_beam_DE.py:491-520) and keeps no
optimiser state, so training cannot resume.  Here a checkpoint is a plain dict of tensors keyed by the reference's
parameter names -- loadable into the reference's own classes and vice versa -- plus, optionally, the fused optimiser's
state (Adam moments per parameter name, step counter, learning rate) and the dropout generator's {seed, step} words.
Whole-module pickles keep working too (the modules hold no library handles)."""
import torch

FORMAT = "vag-nmt-checkpoint-v1"


def save_checkpoint(path, model, train_step=None, extra=None):
    ckpt = {"format": FORMAT,
            "state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
            "extra": extra or {}}
    rng = getattr(model, "_vag_rng", None)
    if rng is not None:
        ckpt["dropout_rng"] = rng.detach().cpu().clone()
    if train_step is not None:
        if hasattr(train_step, "check") and train_step.fp.flat.is_cuda:
            train_step.check()            # never write a checkpoint over steps the device had to skip without saying so
        if hasattr(train_step, "gather_optimizer_state"):
            train_step.gather_optimizer_state()      # (sharded optimiser: every rank's copy of the moments made whole first)
        fp = train_step.fp
        m, v = {}, {}
        for name, p in fp.named:
            o, k = fp.offsets[name], p.numel()
            m[name] = fp.m[o:o + k].view_as(p).detach().cpu().clone()
            v[name] = fp.v[o:o + k].view_as(p).detach().cpu().clone()
        ckpt["optimizer"] = {"kind": "adam", "m": m, "v": v, "step": int(train_step.step_count.item()),
                             "lr": train_step.lr, "betas": tuple(train_step.betas), "eps": train_step.eps,
                             "weight_decay": train_step.wd, "clip": train_step.clip}
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path, model, train_step=None, map_location="cpu"):
    """Accepts this format, a bare state_dict (reference parameter names), or a pickled module."""
    obj = torch.load(path, map_location=map_location, weights_only=False)
    if isinstance(obj, torch.nn.Module):
        sd, ckpt = obj.state_dict(), {}
    elif isinstance(obj, dict) and obj.get("format") == FORMAT:
        sd, ckpt = obj["state_dict"], obj
    else:
        sd, ckpt = obj, {}
    missing, unexpected = model.load_state_dict(sd, strict=False)      # copies into the (possibly flat-buffer) views
    if unexpected or not set(missing) <= {"decoder.out.weight"}:
        raise RuntimeError("checkpoint does not match the model: missing %s unexpected %s" % (missing, unexpected))
    if "dropout_rng" in ckpt:
        # in place: captured step graphs hold the device address of these two words
        from .state import dropout_rng
        dropout_rng(model, next(model.parameters()).device).copy_(ckpt["dropout_rng"])
    if train_step is not None and "optimizer" in ckpt:
        opt, fp = ckpt["optimizer"], train_step.fp
        with torch.no_grad():
            for name, p in fp.named:
                o, k = fp.offsets[name], p.numel()
                fp.m[o:o + k].copy_(opt["m"][name].reshape(-1))
                fp.v[o:o + k].copy_(opt["v"][name].reshape(-1))
            train_step.step_count.fill_(int(opt["step"]))
        train_step.set_lr(opt["lr"])
    if train_step is not None and hasattr(train_step.backend, "after_optimizer"):
        train_step.backend.after_optimizer()      # derived weights follow the restored parameters
    return ckpt
