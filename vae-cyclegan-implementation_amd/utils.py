"""Checkpoint wire format of the reference (utils.py:17-54), so runs resume across reference <-> this build.

A checkpoint is `torch.save` of
    {'epoch', 'model_state_dict', 'optimizer_states', 'loss', 'args'}
with the reference's state_dict keys and OIHW shapes (spectral-norm `weight_orig / weight_u / weight_v` included) and
`optimizer_states = model.save_optimizer_states()` in torch.optim.Adam's state_dict format (optim.FusedAdam speaks it:
per-parameter `step`, `exp_avg`, `exp_avg_sq`, one param group).  tests/golden/checkpoint_skeleton.json holds the
structure of files written by the reference itself; tests/test_gpu_parity.py compares ours against it and checks that a
resumed run continues bit-identically.  The DoubleAE/DoubleVAE -> Cycle remaps (reference utils.py:57-239) belong to
composites that are not built."""
import os

import torch


def save_checkpoint(model, epoch, loss, args, filename):
    """reference utils.py:17-28.  Tensors are written from the CPU so the file loads on a machine without a GPU
    (the reference's own loader passes map_location anyway)."""
    checkpoint = {
        "epoch": epoch,
        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer_states": _to_cpu(model.save_optimizer_states()),
        "loss": loss,
        "args": vars(args),
    }
    torch.save(checkpoint, filename)
    print(f"Checkpoint saved to {filename}")


def load_checkpoint(model, filename, device):
    """reference utils.py:31-54: restores parameters, then optimizer state(s); returns (epoch, loss)."""
    if not os.path.exists(filename):
        raise FileNotFoundError(f"No checkpoint found at {filename}")
    checkpoint = torch.load(filename, map_location=device, weights_only=False)
    model.load_state_dict(checkpoint["model_state_dict"])
    # the reference configures a default optimizer when none exists yet (utils.py:38-44)
    if getattr(model, "optimizer", None) is None and getattr(model, "optimizer_G", None) is None:
        try:
            model.configure_optimizers()
        except Exception:
            pass
    if "optimizer_states" in checkpoint:
        model.load_optimizer_states(checkpoint["optimizer_states"])
    epoch, loss = checkpoint["epoch"], checkpoint["loss"]
    print(f"Loaded checkpoint from {filename} (epoch {epoch}, loss {loss:.4f})")
    return epoch, loss


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj
