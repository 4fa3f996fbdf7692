#!/usr/bin/env python3
"""Diagnostic (not product, not a test): per-parameter gradient error of the HIP path against the
oracle evaluated in float64, in backward order, for one VAE step.  Shows where along the backward
chain rounding error enters.  Usage (GPU box): python tools/grad_error_profile.py [vae|ae] [size] [batch]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("vae-cyclegan-implementation_amd")
oracle = importlib.import_module("vcg_oracle")

arch = sys.argv[1] if len(sys.argv) > 1 else "vae"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SEED = 20261003
key = "vae64" if arch == "vae" else "ae64"
dev = torch.device("cuda:0")

model = pkg.Networks.VariationalAutoencoder(64) if arch == "vae" else pkg.Networks.Autoencoder()
shapes = {f"{key}.{k}": tuple(v.shape) for k, v in model.state_dict().items()}
sd = {k[len(key) + 1:]: torch.from_numpy(v) for k, v in pkg.synth.state_dict_like(shapes, SEED, bias_std=0.02).items()}
model.load_state_dict(sd)
model = model.to(dev).train()
model.configure_optimizers(lr=2e-4)
model.configure_loss(lambda_kl=1e-5)
x, _ = pkg.synth.batch(B, S, SEED, step=0)
eps = pkg.synth.eps_list(1, (B, 64, S // 16, S // 16), SEED, step=0)[0]
if arch == "vae":
    pkg.ops.inject_eps([torch.from_numpy(eps)])
m = model.training_step({"x": torch.from_numpy(x).to(dev), "y": torch.from_numpy(x).to(dev)})
mine = {n: p.grad.detach().cpu().double() for n, p in model.named_parameters()}


def run_oracle(dtype):
    P = {k: v.to(dtype) for k, v in sd.items()}
    xb = torch.from_numpy(x).to(dtype)
    if arch == "vae":
        mo, _, g = oracle.vae_step(P, {}, xb, xb, torch.from_numpy(eps).to(dtype), 2e-4, 1e-5)
    else:
        mo, _, g = oracle.autoencoder_step(P, {}, xb, xb, 2e-4)
    return mo, {k: v.double() for k, v in g.items()}


m64, g64 = run_oracle(torch.float64)
m32, g32 = run_oracle(torch.float32)
print("metrics hip", m, "\nmetrics f64", m64)
print(f"{'parameter (backward order)':58s} {'hip vs f64':>11s} {'cpu32 vs f64':>13s} {'|g|':>10s}")
for n in reversed(list(mine)):
    ref = g64[n]
    nr = ref.norm().item()
    e1 = ((mine[n] - ref).norm() / max(nr, 1e-30)).item()
    e2 = ((g32[n] - ref).norm() / max(nr, 1e-30)).item()
    print(f"{n:58s} {e1:11.2e} {e2:13.2e} {nr:10.3e}")

# ---- activation-gradient profile through the encoder (where does the error enter?) ---------------------
if arch == "vae":
    store = {}
    model2 = pkg.Networks.VariationalAutoencoder(64)
    model2.load_state_dict(sd)
    model2 = model2.to(dev).train()
    def make_hook(i):
        def fwd_hook(mod, inp, out):
            def grad_hook(g):
                store[i] = pkg.ops.to_nchw_contiguous(g).cpu().double()
            if out.requires_grad:
                out.register_hook(grad_hook)
        return fwd_hook
    for i, layer in enumerate(model2.encoder.model):
        layer.register_forward_hook(make_hook(i))
    pkg.ops.inject_eps([torch.from_numpy(eps)])
    xb = torch.from_numpy(x).to(dev)
    out, mu, lv = model2(xb)
    loss = pkg.ops.weighted_sum([pkg.ops.l1_loss(out, pkg.ops.to_nhwc(xb)), pkg.ops.kl_loss(mu, lv)], [1.0, 1e-5])
    loss.backward()

    def oracle_acts(dtype):
        P = {k: v.to(dtype).requires_grad_(False) for k, v in sd.items()}
        h = torch.from_numpy(x).to(dtype)
        acts = []
        h = oracle.casb(h, P, "encoder.model.0.", 1, 3, "ReLU", True); h.requires_grad_(True); acts.append(h)
        for i in (1, 2, 3, 4):
            h = oracle.d_block(h, P, f"encoder.model.{i}."); h.retain_grad(); acts.append(h)
        h = oracle.r_block(h, P, "encoder.model.5."); h.retain_grad(); acts.append(h)
        z, mu_, lv_ = oracle.variational_encoder_block(h, P, "variational_encoder_block.", torch.from_numpy(eps).to(dtype))
        o = oracle.decoder(oracle.s_conv(z, P, "variational_decoder_block.conv."), P, "decoder.")
        l = oracle.l1(o, torch.from_numpy(x).to(dtype)) + 1e-5 * oracle.kl_loss(mu_, lv_)
        l.backward()
        return [a.grad.double() for a in acts]
    a64, a32 = oracle_acts(torch.float64), oracle_acts(torch.float32)
    print("\ngradient w.r.t. the OUTPUT of each encoder layer")
    for i in reversed(range(6)):
        r = a64[i]
        print(f"encoder.model.{i} out: hip vs f64 {((store[i] - r).norm() / r.norm()).item():.2e}   cpu32 vs f64 {((a32[i] - r).norm() / r.norm()).item():.2e}"
              f"   |g| {r.norm().item():.3e}  max|g|/rms {(r.abs().max() / r.pow(2).mean().sqrt()).item():.1f}")

    # ---- which channels of encoder.model.3 carry the error? (bias grad = per-channel sum of dt) ----------
    P64 = {k: v.double() for k, v in sd.items()}
    h = torch.from_numpy(x).double()
    h = oracle.casb(h, P64, "encoder.model.0.", 1, 3, "ReLU", True)
    for i in (1, 2):
        h = oracle.d_block(h, P64, f"encoder.model.{i}.")
    pre = torch.relu(oracle.reflect_conv(torch.nn.functional.pixel_unshuffle(h, 2), P64["encoder.model.3.conv.weight"], P64["encoder.model.3.conv.bias"]))
    var = pre.var(dim=(2, 3), unbiased=False)          # (N, C)
    mean = pre.mean(dim=(2, 3))
    alive = (pre > 0).sum(dim=(2, 3))
    for lname in ("encoder.model.3.conv.bias", "encoder.model.2.conv.bias"):
        err = (mine[lname] - g64[lname]).abs()
        print(f"\n{lname}: total rel err {(err.norm() / g64[lname].norm()).item():.2e}; top channels by |error|")
        if lname.endswith("model.3.conv.bias"):
            for c in err.argsort(descending=True)[:8].tolist():
                print(f"  c={c:4d} hip {mine[lname][c].item(): .6e} f64 {g64[lname][c].item(): .6e} cpu32 {g32[lname][c].item(): .6e} | per-sample var {var[:, c].tolist()} mean {mean[:, c].tolist()} alive {alive[:, c].tolist()}")
        e_sorted = err.sort(descending=True).values
        print(f"  share of squared error in top 8 channels: {(e_sorted[:8].pow(2).sum() / err.pow(2).sum()).item():.3f}")

    # ---- confirm: is it a ReLU-mask flip?  compare the HIP pre-IN activation of model.3 with fp64 ----------
    acts = {}
    m3 = model2.encoder.model[3]
    xin = {}
    model2.encoder.model[2].register_forward_hook(lambda mod, inp, out: xin.__setitem__("x", out.detach()))
    pkg.ops.inject_eps([torch.from_numpy(eps)])
    with torch.no_grad():
        model2(xb)
        spec = m3._spec
        import copy
        plain = copy.copy(spec)
        plain.norm = False
        t3 = pkg.ops.conv_block(xin["x"], m3.conv.weight, m3.conv.bias, plain)      # conv + bias + ReLU only
    t3 = pkg.ops.to_nchw_contiguous(t3).cpu().double()
    flips = ((t3 > 0) != (pre > 0))
    print(f"\nReLU mask disagreements HIP-fp32 vs fp64 at encoder.model.3: {int(flips.sum())} of {flips.numel()}")
    for n_, c_, h_, w_ in flips.nonzero().tolist()[:10]:
        raw = oracle.reflect_conv(torch.nn.functional.pixel_unshuffle(h, 2), P64["encoder.model.3.conv.weight"], P64["encoder.model.3.conv.bias"])
        print(f"  (n={n_}, c={c_}, h={h_}, w={w_}): fp64 pre-activation {raw[n_, c_, h_, w_].item(): .3e}, HIP post-ReLU {t3[n_, c_, h_, w_].item(): .3e}")
