#!/usr/bin/env python3
"""Diagnostic: three CycleVAEGAN steps at the headline config (batch 8, 256x256) printed as JSON, to compare two builds
of the library (e.g. the diagnostic build with and without VCG_NO_WINOGRAD=1).  Round 1: all 18 metrics of the first
step agree to 6e-6 between the Winograd and the direct paths; later steps of the GAN decorrelate (Adam's first updates
are sign-like, so rounding noise in near-zero gradients moves parameters by +-lr)."""
import importlib, os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
if os.environ.get("VCG_LIBRARY"):
    pkg._native.LIB_PATH = os.path.abspath(os.environ["VCG_LIBRARY"])
ops, N = pkg.ops, pkg.Networks
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = N.CycleVAEGAN(latent_dim=64, paired=False).to(dev).train()
m.configure_optimizers(lr=2e-4); m.configure_loss(lambda_kl=1e-5, lambda_gan=1.0, lambda_identity=5.0, lambda_cycle=10.0, lambda_recon=1.0)
ops.manual_seed(4321)
B, S = 8, 256
nq = (B * 3 * S * S + 3) // 4
x = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=0)); y = ops.to_nhwc(ops.rand_uniform((B, 3, S, S), dev, seed=1234, offset=nq))
out = []
for i in range(3):
    out.append(m.training_step({"x": x, "y": y}))
print(json.dumps(out))
