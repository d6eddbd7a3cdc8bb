"""Small host-side helpers shared by the drop-in modules."""
import torch


def lengths_tensor(lengths, device):
    """The reference passes a python list of source lengths; the kernels read int32 on the device."""
    if torch.is_tensor(lengths):
        return lengths.to(device=device, dtype=torch.int32)
    return torch.tensor(list(lengths), dtype=torch.int32, device=device)


def dropout_rng(module, device):
    """Device-resident {seed, step} words of the counter-based dropout generator, one per root module.
    Not part of state_dict (reference state_dicts stay loadable)."""
    rng = getattr(module, "_vag_rng", None)
    if rng is None or rng.device != device:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        rng = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        module._vag_rng = rng
    return rng
