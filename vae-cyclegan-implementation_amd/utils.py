"""Checkpoint wire format of the reference (utils.py:17-54), so runs resume across reference <-> this build.

A checkpoint is `torch.save` of
    {'epoch', 'model_state_dict', 'optimizer_states', 'loss', 'args'}
with the reference's state_dict keys and OIHW shapes (spectral-norm `weight_orig / weight_u / weight_v` included) and
`optimizer_states = model.save_optimizer_states()` in torch.optim.Adam's state_dict format (optim.FusedAdam speaks it:
per-parameter `step`, `exp_avg`, `exp_avg_sq`, one param group).  tests/golden/checkpoint_skeleton.json holds the
structure of files written by the reference itself; tests/test_gpu_parity.py compares ours against it and checks that a
resumed run continues bit-identically.  `load_pretrained_double*_to_cycle*` are the reference's remaps of a pretraining
checkpoint onto a Cycle model (utils.py:57-239)."""
import os

import torch

from . import ops

# extra keys of the last checkpoint `load_checkpoint` read ({} for a file the reference wrote): `best_test_loss`
LAST_EXTRAS = {}


def _rank():
    """Rank of this process in the data-parallel job (0 when single-process)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:
        pass
    return int(os.environ.get("RANK", "0"))


def save_checkpoint(model, epoch, loss, args, filename, best_test_loss=None):
    """reference utils.py:17-28.  Tensors are written from the CPU so the file loads on a machine without a GPU
    (the reference's own loader passes map_location anyway).  Two keys are ADDED to the reference's five (its loader
    reads the ones it knows and ignores the rest): the position of the on-device eps stream (`vcg_eps_rng`) so that a
    resumed run draws the eps it would have drawn, and the best test loss so far (`vcg_best_test_loss`) so that a
    resumed run does not overwrite best_model.pth with a worse model.

    Under data parallelism one rank writes the file but every rank draws from its OWN eps stream
    (`ops.rank_seed(base, rank)`): what is saved is the BASE seed (the writer's seed with its rank offset taken out) and
    the stream offset, which is the same on all ranks (they run identical steps); `load_checkpoint` re-derives each
    rank's seed from the base."""
    rank = _rank()
    seed = int(ops._RNG["seed"])
    checkpoint = {
        "epoch": epoch,
        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizer_states": _to_cpu(model.save_optimizer_states()),
        "loss": loss,
        "args": vars(args),
        "vcg_eps_rng": {"seed": seed, "offset": int(ops._RNG["offset"]), "base_seed": ops.base_seed_of(seed, rank), "rank": rank},
    }
    if best_test_loss is not None:
        checkpoint["vcg_best_test_loss"] = float(best_test_loss)
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
    LAST_EXTRAS.clear()
    if "vcg_eps_rng" in checkpoint:                      # ours: continue the eps stream where the saved run stood
        rng = checkpoint["vcg_eps_rng"]
        # every rank continues ITS stream: re-derive the per-rank seed from the base (a round-2 file has no base_seed: it
        # was written by rank 0, whose seed is rank_seed(base, 0)); loading the writer's seed on every rank would make all
        # ranks draw the same eps for their shards
        base = int(rng["base_seed"]) if "base_seed" in rng else ops.base_seed_of(int(rng["seed"]), 0)
        ops._RNG["seed"] = ops.rank_seed(base, _rank())
        ops._RNG["offset"] = int(rng["offset"])
    if "vcg_best_test_loss" in checkpoint:
        LAST_EXTRAS["best_test_loss"] = float(checkpoint["vcg_best_test_loss"])
    ops.PARAM_EPOCH[0] += 1                              # load_state_dict wrote through .data: drop every weight pack
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


def _split_by_prefix(state_dict, prefixes):
    parts = {p: {} for p in prefixes}
    for key, value in state_dict.items():
        for p in prefixes:
            if key.startswith(p + "."):
                parts[p][key[len(p) + 1:]] = value
                break
    return parts


def load_pretrained_doubleae_to_cycleae(cycleae_model, doubleae_checkpoint_path, device):
    """reference utils.py:57-121: G (A->B) <- encoder + decoder_B, F (B->A) <- encoder + decoder_A (for CycleAE and
    CycleAEGAN, whose generators are plain autoencoders)."""
    if not os.path.exists(doubleae_checkpoint_path):
        raise FileNotFoundError(f"No DoubleAutoencoder checkpoint found at {doubleae_checkpoint_path}")
    print(f"Loading DoubleAutoencoder weights from {doubleae_checkpoint_path}")
    sd = torch.load(doubleae_checkpoint_path, map_location=device, weights_only=False)["model_state_dict"]
    parts = _split_by_prefix(sd, ("encoder", "decoder_A", "decoder_B"))
    cycleae_model.G.encoder.load_state_dict(parts["encoder"])
    cycleae_model.G.decoder.load_state_dict(parts["decoder_B"])
    cycleae_model.F.encoder.load_state_dict(parts["encoder"])
    cycleae_model.F.decoder.load_state_dict(parts["decoder_A"])
    print("Successfully loaded DoubleAutoencoder weights into CycleAE")


def load_pretrained_doublevae_to_cyclevae(cycle_model, doublevae_checkpoint_path, device):
    """reference utils.py:124-239: G <- encoder + VAE blocks B + decoder_B, F <- encoder + VAE blocks A + decoder_A (for
    CycleVAE and CycleVAEGAN), with the reference's check that G and F did not end up swapped."""
    if not os.path.exists(doublevae_checkpoint_path):
        raise FileNotFoundError(f"No DoubleVariationalAutoencoder checkpoint found at {doublevae_checkpoint_path}")
    print(f"Loading DoubleVariationalAutoencoder weights from {doublevae_checkpoint_path}")
    sd = torch.load(doublevae_checkpoint_path, map_location=device, weights_only=False)["model_state_dict"]
    parts = _split_by_prefix(sd, ("encoder", "vae_encoder_block_A", "vae_encoder_block_B", "vae_decoder_block_A",
                                  "vae_decoder_block_B", "decoder_A", "decoder_B"))
    for gen, sfx in ((cycle_model.G, "B"), (cycle_model.F, "A")):
        gen.encoder.load_state_dict(parts["encoder"])
        gen.variational_encoder_block.load_state_dict(parts["vae_encoder_block_" + sfx])
        gen.variational_decoder_block.load_state_dict(parts["vae_decoder_block_" + sfx])
        gen.decoder.load_state_dict(parts["decoder_" + sfx])
        for name, value in gen.decoder.state_dict().items():
            assert torch.equal(value.cpu(), parts["decoder_" + sfx][name].cpu()), \
                f"decoder mismatch at {name} - G and F may be swapped!"
    print("Successfully loaded DoubleVariationalAutoencoder weights into the Cycle model")

