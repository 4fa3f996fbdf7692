#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of one CycleVAEGAN step (256x256, batch 1) against the
oracle in float64, for the HIP path and for the oracle in float32 (= the reference's CPU numerics).
GPU box: python tools/grad_error_profile_gan.py [paired]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
oracle = importlib.import_module("vcg_oracle")
torch.set_num_threads(16)

paired = len(sys.argv) > 1 and sys.argv[1] == "paired"
key = "cvg256_paired" if paired else "cvg256_unpaired"
SEED = 20261003
dev = torch.device("cuda:0")
model = pkg.Networks.CycleVAEGAN(latent_dim=64, paired=paired)
shapes = {f"{key}.{k}": tuple(v.shape) for k, v in model.state_dict().items()}
sd = {k[len(key) + 1:]: torch.from_numpy(v) for k, v in pkg.synth.state_dict_like(shapes, SEED, bias_std=0.02).items()}
model.load_state_dict(sd)
model = model.to(dev).train()
model.configure_optimizers(lr=2e-4)
model.configure_loss(lambda_kl=1e-5, lambda_gan=1.0, lambda_identity=5.0, lambda_cycle=10.0)
x, y = pkg.synth.batch(1, 256, SEED, step=0)
eps = pkg.synth.eps_list(6, (1, 64, 16, 16), SEED, step=0)
pkg.ops.inject_eps([torch.from_numpy(e) for e in eps])
m = model.training_step({"x": torch.from_numpy(x).to(dev), "y": torch.from_numpy(y).to(dev)})
mine = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}


def run(dtype):
    P = {k: v.to(dtype) for k, v in sd.items()}
    mo, _, gg, gd = oracle.cyclevaegan_step(P, {}, torch.from_numpy(x).to(dtype), torch.from_numpy(y).to(dtype),
                                            [torch.from_numpy(e).to(dtype) for e in eps], 2e-4, paired)
    return mo, {k: v.double() for k, v in {**gg, **gd}.items()}


m64, g64 = run(torch.float64)
m32, g32 = run(torch.float32)
print("hip  ", {k: round(v, 6) for k, v in m.items()})
print("f64  ", {k: round(v, 6) for k, v in m64.items()})
rows = []
for n in mine:
    r = g64[n]
    nr = r.norm().item()
    if nr < 1e-12:
        continue
    rows.append((n, ((mine[n] - r).norm() / nr).item(), ((g32[n] - r).norm() / nr).item(), ((mine[n] - g32[n]).norm() / nr).item()))
print(f"{'parameter':52s} {'hip-f64':>9s} {'cpu32-f64':>10s} {'hip-cpu32':>10s}")
for n, a, b, c in rows:
    if n.endswith("weight") or n.endswith("weight_orig"):
        print(f"{n:52s} {a:9.2e} {b:10.2e} {c:10.2e}")
for net in ("G.", "F.", "DX.", "DY."):
    sel = [(a, b) for n, a, b, c in rows if n.startswith(net) and (n.endswith("weight") or n.endswith("weight_orig"))]
    ta = torch.tensor([s[0] for s in sel])
    tb = torch.tensor([s[1] for s in sel])
    print(f"{net:4s} weights: hip median {ta.median().item():.2e} max {ta.max().item():.2e} | cpu32 median {tb.median().item():.2e} max {tb.max().item():.2e}")
