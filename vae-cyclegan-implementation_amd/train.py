#!/usr/bin/env python3
"""Training driver with the reference's `train.py` CLI for the hot path.

Kept from the reference (train.py:588-656): every flag name and default.  Added: `--dataset synthetic`
(on-device U[0,1) batches, the benchmark input), `--latent_dim` (the reference's README mentions it,
its argparse lacks it: create_model always used 64), `--steps_per_epoch`, `--seed`, and the BASELINE
aliases `ae` / `vae_cyclegan` for `--architecture`.  Under `torch.distributed.run` each rank trains
its shard of the global batch and gradients are exchanged by `parallel.GradReducer` (RCCL).

Built beyond the three benchmarked architectures: every architecture of the reference's factory,
`validate`, the checkpoint wire format with `--resume`, the pretraining hand-over (`--pretrained_double*`).
Out of scope, as in SURVEY.md §2: TensorBoard and the reference's PIL / torchvision dataset classes (`Data_Manager.py`).

`create_model` and `train_epoch` mirror the reference's functions of the same name (train.py:43-128).
"""
import argparse
import gc
import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this driver
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")        # kernel arguments in device memory: shorter dispatch gaps (package __init__)

import numpy as np  # noqa: E402
import torch  # noqa: E402

if __package__ in (None, ""):
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module("vae-cyclegan-implementation_amd")
    Networks, ops, parallel, utils, input_pipeline = _pkg.Networks, _pkg.ops, _pkg.parallel, _pkg.utils, _pkg.input_pipeline
else:
    from . import Networks, input_pipeline, ops, parallel, utils

ALIASES = {"ae": "autoencoder", "vae_cyclegan": "cyclevaegan"}
REFERENCE_ARCHS = ["autoencoder", "doubleae", "doublevae", "vae", "aegan", "vaegan", "cycleae", "cyclevae",
                   "cycleaegan", "cyclevaegan"]
BUILT = tuple(REFERENCE_ARCHS)


def create_model(architecture, paired=True, latent_dim=64):
    """reference train.py:43-77 (which never passes latent_dim)."""
    architecture = ALIASES.get(architecture, architecture)
    if architecture == "autoencoder":
        model = Networks.Autoencoder()
        print("Created Autoencoder")
    elif architecture == "vae":
        model = Networks.VariationalAutoencoder(latent_dim=latent_dim)
        print("Created Variational Autoencoder")
    elif architecture == "doubleae":
        model = Networks.DoubleAutoencoder()
        print("Created Double Autoencoder (shared encoder + 2 decoders)")
    elif architecture == "doublevae":
        model = Networks.DoubleVariationalAutoencoder(latent_dim=latent_dim)
        print("Created Double VAE (shared encoder + 2 VAE blocks + 2 decoders)")
    elif architecture == "aegan":
        model = Networks.AEGAN()
        print("Created AE-GAN")
    elif architecture == "vaegan":
        model = Networks.VAEGAN(latent_dim=latent_dim)
        print("Created VAE-GAN")
    elif architecture == "cycleae":
        model = Networks.CycleAE(paired=paired)
        print(f"Created Cycle Autoencoder ({'paired' if paired else 'unpaired'} mode)")
    elif architecture == "cyclevae":
        model = Networks.CycleVAE(latent_dim=latent_dim, paired=paired)
        print(f"Created Cycle VAE ({'paired' if paired else 'unpaired'} mode)")
    elif architecture == "cycleaegan":
        model = Networks.CycleAEGAN(paired=paired)
        print(f"Created Cycle AE-GAN ({'paired' if paired else 'unpaired'} mode)")
    elif architecture == "cyclevaegan":
        model = Networks.CycleVAEGAN(latent_dim=latent_dim, paired=paired)
        print(f"Created Cycle VAE-GAN ({'paired' if paired else 'unpaired'} mode)")
    else:
        raise ValueError(f"Unknown architecture: {architecture}")
    return model


class SyntheticLoader:
    """len()-able iterable of {'x','y'} batches drawn on the device: x, y ~ U[0,1), (B,3,S,S)."""

    def __init__(self, batch_size, image_size, steps, device, seed=1234, rank=0, same_xy=False, epoch=0):
        self.b, self.s, self.steps, self.dev = batch_size, image_size, steps, device
        self.seed, self.rank, self.same_xy, self.epoch = seed, rank, same_xy, epoch

    def __len__(self):
        return self.steps

    def __iter__(self):
        nq = (self.b * 3 * self.s * self.s + 3) // 4
        for i in range(self.steps):
            base = (((self.epoch * self.steps + i) * 64 + self.rank) * 2) * nq
            x = ops.rand_uniform((self.b, 3, self.s, self.s), self.dev, self.seed, base)
            y = x if self.same_xy else ops.rand_uniform((self.b, 3, self.s, self.s), self.dev, self.seed, base + nq)
            yield {"x": x, "y": y}


_FROZEN = [False]


def _freeze_long_lived_objects():
    """Once per process: collect, then move the survivors (modules, parameters, ctypes tables) out of the cyclic GC's way.
    A full collection walks all of them (~30 ms of host time, with the GPU idle behind it); freezing every epoch would pin
    each epoch's garbage for good, so it is done once."""
    if not _FROZEN[0]:
        gc.collect()
        gc.freeze()
        _FROZEN[0] = True


def train_epoch(model, dataloader, device, args, writer=None, epoch=None):
    """reference train.py:80-128: per-batch training_step, metric sums averaged by len(dataloader),
    G_loss as the headline loss.  The reference also runs one extra train-mode forward per batch for a
    visualisation tensor that its caller never uses (:112-117); it is reproduced only with
    --reference_viz_forward (it advances the eps stream and costs a full forward)."""
    model.train()
    total_loss = 0.0
    loss_components = {}
    last_output = last_x = last_y = None
    _freeze_long_lived_objects()
    for batch in dataloader:
        batch["x"] = batch["x"].to(device)
        batch["y"] = batch["y"].to(device)
        metrics = model.training_step(batch)
        total_loss += metrics["G_loss"]
        for key, value in metrics.items():
            loss_components[key] = loss_components.get(key, 0.0) + value
        last_x, last_y = batch["x"], batch["y"]
        if getattr(args, "reference_viz_forward", False):
            with torch.no_grad():
                if isinstance(model, (Networks.Autoencoder, Networks.VariationalAutoencoder)):
                    last_output = model(last_x)[0]
                else:
                    last_output = model(last_x, last_y)[0]
    n = len(dataloader)
    if n > 0:
        return total_loss / n, {k: v / n for k, v in loss_components.items()}, last_output, last_x, last_y
    return float("nan"), {k: float("nan") for k in loss_components}, last_output, last_x, last_y


def validate(model, dataloader, device, args):
    """reference train.py:131-171: model.eval(), forward-only `validation_step` per batch, metrics averaged over
    len(dataloader) with G_loss as the headline; returns (avg_loss, avg_components, last Gx, last Fy, last x, last y).
    Eval mode changes one thing on this path: the discriminators' spectral norm uses the stored u, v as they are."""
    model.eval()
    total_loss = 0.0
    loss_components = {}
    last_Gx = last_Fy = last_x = last_y = None
    with torch.no_grad():
        for batch in dataloader:
            batch["x"] = batch["x"].to(device)
            batch["y"] = batch["y"].to(device)
            metrics = model.validation_step(batch)
            Gx = metrics.pop("Gx")
            Fy = metrics.pop("Fy", None)            # only the Cycle models return it
            total_loss += metrics["G_loss"]
            for key, value in metrics.items():
                loss_components[key] = loss_components.get(key, 0.0) + value
            last_Gx, last_Fy, last_x, last_y = Gx, Fy, batch["x"], batch["y"]
    n = len(dataloader)
    red = getattr(model, "grad_reducer", None)
    if red is not None:
        # data parallel: each rank validated its shard of the test set.  Shards may differ by one batch (or be empty: a test
        # split smaller than the world size), so the ranks pool their per-batch SUMS and batch counts — the same batch-weighted
        # mean the single-process loop computes — instead of averaging per-rank means
        import torch.distributed as dist
        pooled = [None] * red.world
        dist.all_gather_object(pooled, (n, total_loss, loss_components), group=red.group)
        n = sum(p[0] for p in pooled)
        total_loss = sum(p[1] for p in pooled)
        loss_components = {}
        for _, _, comp in pooled:
            for k, v in comp.items():
                loss_components[k] = loss_components.get(k, 0.0) + v
    if n == 0:
        raise ValueError("validate(): the test set is empty (no batch on any rank): lower --test_split or skip validation")
    avg_loss, avg = total_loss / n, {k: v / n for k, v in loss_components.items()}
    return avg_loss, avg, last_Gx, last_Fy, last_x, last_y


def build_parser():
    p = argparse.ArgumentParser(description="Train VAE-CycleGAN models (MI355X-native path)")
    p.add_argument("--architecture", type=str, default="autoencoder", choices=REFERENCE_ARCHS + list(ALIASES))
    p.add_argument("--paired", action="store_true", default=False)
    p.add_argument("--unpaired", dest="paired", action="store_false")
    p.add_argument("--pretrained_doubleae", type=str, default=None)
    p.add_argument("--pretrained_doublevae", type=str, default=None)
    p.add_argument("--data_dir", type=str, default="dataset")
    p.add_argument("--source_modality", type=str, default=None)
    p.add_argument("--target_modality", type=str, default=None)
    p.add_argument("--image_size", type=int, default=256)
    p.add_argument("--test_split", type=float, default=0.1)
    p.add_argument("--dataset", type=str, default="hypersim", choices=["hypersim", "summer2winter", "maps", "synthetic"])
    p.add_argument("--batch_size", type=int, default=5)
    p.add_argument("--epochs", type=int, default=100)
    p.add_argument("--lr", type=float, default=0.0002)
    p.add_argument("--lambda_kl", type=float, default=1e-5)
    p.add_argument("--lambda_gan", type=float, default=1.0)
    p.add_argument("--lambda_identity", type=float, default=5.0)
    p.add_argument("--lambda_cycle", type=float, default=10.0)
    p.add_argument("--lambda_recon", type=float, default=1.0)
    p.add_argument("--output_dir", type=str, default="runs")
    p.add_argument("--save_freq", type=int, default=10)
    p.add_argument("--log_image_freq", type=int, default=5)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--num_workers", type=int, default=1)
    p.add_argument("--no_cuda", action="store_true")
    # additions
    p.add_argument("--latent_dim", type=int, default=64)
    p.add_argument("--steps_per_epoch", type=int, default=20, help="synthetic dataset: batches per epoch")
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--reference_viz_forward", action="store_true")
    return p


def create_dataloaders(args, device, rank, world, epoch_seed):
    """reference train.py:174-357 (create_dataloaders_hypersim / _maps / _summer2winter) on the device-side input pipeline:
    PIL decodes, the MI355X flips, crops, resamples, jitters and converts (input_pipeline.py), from the reference's directory
    layouts (hypersim: the PNG scene tree Data_Manager.py:18-138 reads; its download / HDF5 conversion is out of scope).
    Under data parallelism every rank gets the same `epoch_seed` and takes its shard of the shared per-epoch permutation
    (train and test sets alike): one epoch is one pass over the data, and `validate` averages the metrics over ranks."""
    if args.dataset == "hypersim":
        # reference train.py:174-239: one HypersimDataset with the TRAINING transforms, random_split into train / test
        mods = [m for m in (args.source_modality, args.target_modality) if m]
        if len(mods) == 2 and mods[0] == mods[1]:
            mods = mods[:1]          # the reference keys its images by modality name: two equal names are ONE image, x is y
        if not mods:
            raise ValueError("--dataset hypersim needs --source_modality (and --target_modality)")
        full = input_pipeline.HypersimFolders(os.path.join(args.data_dir, "hypersim"), mods, paired=args.paired or len(mods) == 1)
        kw = dict(num_workers=max(1, args.num_workers), same_xy=len(mods) == 1, rank=rank, world=world)
        order = np.random.RandomState(args.seed).permutation(len(full))
        ntrain = int((1 - args.test_split) * len(full)) if args.test_split > 0 else len(full)
        print(f"Training samples: {ntrain}" + (f", Testing samples: {len(full) - ntrain}" if args.test_split > 0 else ""))
        train = input_pipeline.DeviceInputPipeline(full.subset(order[:ntrain]), args.batch_size, args.image_size, device,
                                                   recipe="hypersim", shuffle=True, seed=epoch_seed, **kw)
        test = None
        if args.test_split > 0 and ntrain < len(full):
            test = input_pipeline.DeviceInputPipeline(full.subset(order[ntrain:]), args.batch_size, args.image_size, device,
                                                      recipe="hypersim", shuffle=False, seed=epoch_seed, **kw)
        return train, test
    root = os.path.join(args.data_dir, args.dataset)
    same_xy = args.architecture in ("autoencoder", "vae")
    test_split = "test" if args.dataset == "summer2winter" else "val"
    train_src = input_pipeline.FolderPairs(root, args.dataset, "train")
    test_src = input_pipeline.FolderPairs(root, args.dataset, test_split)
    print(f"Training samples: {len(train_src)}\nTesting samples: {len(test_src)}")
    kw = dict(num_workers=max(1, args.num_workers), same_xy=same_xy, rank=rank, world=world)
    train = input_pipeline.DeviceInputPipeline(train_src, args.batch_size, args.image_size, device, recipe=args.dataset, shuffle=True,
                                               seed=epoch_seed, **kw)
    test = input_pipeline.DeviceInputPipeline(test_src, args.batch_size, args.image_size, device, recipe="test", shuffle=False,
                                              seed=epoch_seed, **kw)
    return train, test


DATASET_MODALITY_DEFAULTS = {                        # reference train.py:367-377
    "hypersim": ("depth", "normal"),
    "summer2winter": ("summer", "winter"),
    "maps": ("satellite", "map"),
    "synthetic": ("synthetic", "synthetic"),
}


def main(args):
    args.architecture = ALIASES.get(args.architecture, args.architecture)
    # reference train.py:362-377, in its order: the autoencoder / VAE check sees the modalities as given, THEN the
    # per-dataset defaults fill in what was not given (they name the run directory and select hypersim's frames)
    if args.architecture in ("autoencoder", "vae"):
        if args.source_modality != args.target_modality:
            raise ValueError("Source and target modalities should be the same for Autoencoder/VAE architectures.")
    default_source, default_target = DATASET_MODALITY_DEFAULTS[args.dataset]
    if args.source_modality is None:
        args.source_modality = default_source
    if args.target_modality is None:
        args.target_modality = default_target
    if args.dataset in ("summer2winter",):
        args.paired = False                          # reference train.py:380-382: unpaired data forces the unpaired objectives
    if args.no_cuda or not torch.cuda.is_available():
        raise RuntimeError("this path has no CPU implementation: an MI355X is required (the reference's own "
                           "train.py is the CPU path)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device, pg_options=parallel.nccl_options())

    # reference train.py:395-411: a resumed run continues in its checkpoint's directory, a new one gets
    # <architecture>_<timestamp>_<source>_to_<target>_<dataset> and writes its args.json there
    if args.resume:
        if not Path(args.resume).exists():
            raise FileNotFoundError(f"No checkpoint found at {args.resume}")
        output_dir = Path(args.resume).parent
        if rank == 0:
            print(f"Using device: {device}  world size {world}\nResuming run in directory: {output_dir}")
    else:
        output_dir = Path(args.output_dir) / (f"{args.architecture}_{datetime.now().strftime('%m%d_%H%M')}_{args.source_modality}_to_"
                                              f"{args.target_modality}_{args.dataset}")
        if rank == 0:
            output_dir.mkdir(parents=True, exist_ok=True)
            with open(output_dir / "args.json", "w") as f:
                json.dump(vars(args), f, indent=2)
            print(f"Using device: {device}  world size {world}\nOutput directory: {output_dir}")

    torch.manual_seed(args.seed)                     # identical replicas
    ops.manual_seed(ops.rank_seed(args.seed, rank))  # per-rank eps stream
    model = create_model(args.architecture, paired=args.paired, latent_dim=args.latent_dim).to(device)
    model.configure_optimizers(lr=args.lr)
    model.configure_loss(lambda_kl=args.lambda_kl, lambda_gan=args.lambda_gan, lambda_identity=args.lambda_identity,
                         lambda_cycle=args.lambda_cycle, lambda_recon=args.lambda_recon)
    if world > 1:
        parallel.attach(model)
    same_xy = args.architecture in ("autoencoder", "vae")
    # reference train.py:448-463: a pretraining checkpoint initialises the generators of a Cycle model
    if args.pretrained_doubleae is not None and args.pretrained_doublevae is not None:
        raise ValueError("Cannot specify both --pretrained_doubleae and --pretrained_doublevae")
    if args.pretrained_doubleae is not None:
        if args.architecture not in ("cycleae", "cyclevae", "cycleaegan", "cyclevaegan"):
            raise ValueError(f"--pretrained_doubleae can only be used with Cycle architectures, not {args.architecture}")
        utils.load_pretrained_doubleae_to_cycleae(model, args.pretrained_doubleae, device)
    if args.pretrained_doublevae is not None:
        if args.architecture not in ("cyclevae", "cyclevaegan"):
            raise ValueError("--pretrained_doublevae can only be used with CycleVAE or CycleVAEGAN architectures, "
                             f"not {args.architecture}")
        utils.load_pretrained_doublevae_to_cyclevae(model, args.pretrained_doublevae, device)
    start_epoch = 0
    if args.resume:                                  # reference train.py:471-477
        if rank == 0:
            print(f"Resuming from checkpoint: {args.resume}")
        start_epoch, _ = utils.load_checkpoint(model, Path(args.resume), device)
        start_epoch += 1
    if world > 1:
        # identical replicas: AFTER every load above, so that replica identity never depends on each rank having read
        # the same file; the broadcast also invalidates the weight packs of anything a load or a warm-up forward packed
        parallel.broadcast_parameters(model)
    best_test_loss = utils.LAST_EXTRAS.get("best_test_loss", float("inf")) if args.resume else float("inf")
    image_loaders = create_dataloaders(args, device, rank, world, args.seed) if args.dataset != "synthetic" else None
    for epoch in range(start_epoch, args.epochs):
        loader = image_loaders[0] if image_loaders else SyntheticLoader(args.batch_size, args.image_size, args.steps_per_epoch, device,
                                                                       args.seed, rank, same_xy, epoch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_loss, comps, *_ = train_epoch(model, loader, device, args, epoch=epoch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rank == 0:
            ips = args.batch_size * world * len(loader) / dt
            print(f"\nEpoch {epoch + 1}/{args.epochs}\nTrain Loss: {train_loss:.4f}   ({ips:.1f} images/s)")
            for k, v in comps.items():
                print(f"  {k}: {v:.6f}")
        # on the test set (reference train.py:533-537: every log_image_freq epochs); here a held-out synthetic stream.
        # validation_step has no exchange in it: every rank validates its shard of the test set and `validate` averages
        # the metrics over ranks
        if args.log_image_freq > 0 and epoch % args.log_image_freq == 0 and not (image_loaders and image_loaders[1] is None):
            test_loader = image_loaders[1] if image_loaders else SyntheticLoader(
                args.batch_size, args.image_size, max(1, args.steps_per_epoch // 10), device, args.seed + 1, rank, same_xy, epoch)
            test_loss, test_comps, *_ = validate(model, test_loader, device, args)
            if rank == 0:
                print(f"Test Loss: {test_loss:.4f}")
                for k, v in test_comps.items():
                    print(f"  {k}: {v:.6f}")
                if test_loss < best_test_loss:       # reference train.py:565-570
                    best_test_loss = test_loss
                    utils.save_checkpoint(model, epoch, test_loss, args, output_dir / "best_model.pth", best_test_loss)
                    print(f"New best model saved (test_loss: {test_loss:.4f})")
        if rank == 0 and (epoch + 1) % args.save_freq == 0:      # reference train.py:573-575 (replicas are identical)
            utils.save_checkpoint(model, epoch, train_loss, args, output_dir / f"checkpoint_epoch_{epoch + 1}.pth", best_test_loss)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(f"\nTraining completed. Run directory: {output_dir}")


if __name__ == "__main__":
    main(build_parser().parse_args())
