#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference)
on deterministic synthetic tensors from `synth.py`.

Runs only in the build container (the reference never travels).  The fixtures hold inputs'
recipe (seed + names; the tensors are rebuilt by synth) and the reference's OUTPUTS: full
tensors for atom-sized cases, metric dicts, strided slices and per-tensor checksums for the
full-size training steps.

    python tests/golden/make_golden.py [atoms] [steps] [validation] [cycleaegan] ... [vae1024] [train_epoch] [checkpoint]      # writes atoms.npz, steps.npz, validation.npz next to this file
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("VCG_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
synth = importlib.import_module("vae-cyclegan-implementation_amd.synth")
sys.path.insert(0, HERE)
from cases import ATOM_CASES, LAMBDAS, LR, SEED  # noqa: E402
from cases import checksum as cases_checksum  # noqa: E402


def import_reference():
    # Networks.py:51 imports torchvision.transforms but never uses it; the module is absent here.
    for name in ("torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REF)
    import Networks  # noqa
    import Losses  # noqa
    return sys.modules["Networks"], sys.modules["Losses"]


def load_synth_params(module, seed, bias_std, prefix=""):
    shapes = {prefix + k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.state_dict_like(shapes, seed, bias_std=bias_std)
    module.load_state_dict({k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items()})
    return sd


class EpsInjector:
    """Replace torch.randn_like (Networks.py:225) by a queue of prepared eps tensors."""

    def __init__(self, eps):
        self.eps = [torch.from_numpy(e) for e in eps]
        self.i = 0

    def __enter__(self):
        self.orig = torch.randn_like

        def fake(t, **kw):
            e = self.eps[self.i]
            self.i += 1
            assert e.shape == t.shape, (e.shape, t.shape)
            return e
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def checksum(t):
    return cases_checksum(t.detach().cpu().numpy())


# ----------------------------------------------------------------------------- atoms
def gen_atoms(N, out):
    for name, (cls, args, kwargs, xshape, scale) in ATOM_CASES.items():
        mod = getattr(N, cls)(*args, **kwargs)
        load_synth_params(mod, SEED, 0.1, prefix=name + ".")
        x = torch.from_numpy(synth.normal(xshape, SEED, name + "/x") * scale).requires_grad_(True)
        y = mod(x)
        g = torch.from_numpy(synth.normal(tuple(y.shape), SEED, name + "/g"))
        y.backward(g)
        out[name + "/y"] = y.detach().numpy()
        out[name + "/dx"] = x.grad.numpy()
        for pn, p in mod.named_parameters():
            out[name + "/d." + pn] = p.grad.numpy()
        print("atom", name, tuple(y.shape))

    # VAE bottleneck with injected eps; input scaled so part of logvar sits on the +-10 clamp
    name = "veb"
    mod = N.VariationalEncoderBlock(16, 8)
    load_synth_params(mod, SEED, 0.1, prefix=name + ".")
    x = torch.from_numpy(synth.normal((2, 16, 4, 6), SEED, name + "/x") * 6.0).requires_grad_(True)
    eps = synth.normal((2, 8, 4, 6), SEED, name + "/eps")
    with EpsInjector([eps]):
        z, mu, lv = mod(x)
    gz = torch.from_numpy(synth.normal(tuple(z.shape), SEED, name + "/gz"))
    gm = torch.from_numpy(synth.normal(tuple(z.shape), SEED, name + "/gm"))
    gl = torch.from_numpy(synth.normal(tuple(z.shape), SEED, name + "/gl"))
    ((z * gz).sum() + (mu * gm).sum() + (lv * gl).sum()).backward()
    out[name + "/z"], out[name + "/mu"], out[name + "/logvar"] = z.detach().numpy(), mu.detach().numpy(), lv.detach().numpy()
    out[name + "/dx"] = x.grad.numpy()
    for pn, p in mod.named_parameters():
        out[name + "/d." + pn] = p.grad.numpy()
    out[name + "/clamped_fraction"] = np.array([(lv.detach().abs() >= 10).float().mean().item()])
    print("atom veb, clamped fraction", out[name + "/clamped_fraction"])

    # losses
    a = torch.from_numpy(synth.normal((2, 3, 8, 8), SEED, "loss/a")).requires_grad_(True)
    b = torch.from_numpy(synth.normal((2, 3, 8, 8), SEED, "loss/b"))
    Lm = sys.modules["Losses"]
    l = Lm.TranslationLoss()(a, b)
    l.backward()
    out["loss/l1"], out["loss/l1_da"] = np.array([l.item()]), a.grad.numpy()
    mu = torch.from_numpy(synth.normal((2, 8, 4, 4), SEED, "loss/mu")).requires_grad_(True)
    lv = torch.from_numpy(synth.normal((2, 8, 4, 4), SEED, "loss/lv") * 8.0).requires_grad_(True)
    k = Lm.KLDivergenceLoss()(mu, lv)
    k.backward()
    out["loss/kl"], out["loss/kl_dmu"], out["loss/kl_dlv"] = np.array([k.item()]), mu.grad.numpy(), lv.grad.numpy()
    d1 = torch.from_numpy(synth.normal((5,), SEED, "loss/d1")).requires_grad_(True)
    d2 = torch.from_numpy(synth.normal((5,), SEED, "loss/d2")).requires_grad_(True)
    tot, real, fake = Lm.GANLossGenerator()(d1, d2)
    tot.backward()
    out["loss/gan_g"] = np.array([tot.item(), real.item(), fake.item()])
    out["loss/gan_g_d1"], out["loss/gan_g_d2"] = d1.grad.numpy().copy(), d2.grad.numpy().copy()
    d1.grad = None
    d2.grad = None
    tot, real, fake = Lm.GANLossDiscriminator()(d1, d2)
    tot.backward()
    out["loss/gan_d"] = np.array([tot.item(), real.item(), fake.item()])
    out["loss/gan_d_d1"], out["loss/gan_d_d2"] = d1.grad.numpy().copy(), d2.grad.numpy().copy()

    # discriminator (needs 256x256): scalar outputs, grads as checksums, spectral-norm buffers
    name = "disc"
    mod = N.Discriminator()
    mod.train()
    load_synth_params(mod, SEED, 0.05, prefix=name + ".")
    x = torch.from_numpy(synth.uniform((2, 3, 256, 256), SEED, name + "/x")).requires_grad_(True)
    o = mod(x)
    g = torch.from_numpy(synth.normal((2,), SEED, name + "/g"))
    o.backward(g)
    out[name + "/y"] = o.detach().numpy()
    out[name + "/dx_slice"] = x.grad[:, :, ::32, ::32].numpy()
    out[name + "/dx_ck"] = checksum(x.grad)
    for pn, p in mod.named_parameters():
        out[name + "/dck." + pn] = checksum(p.grad)
    sd = mod.state_dict()
    out[name + "/u"] = sd["model.4.weight_u"].numpy()
    out[name + "/v_ck"] = checksum(sd["model.4.weight_v"])
    print("atom disc", o.detach().numpy())


# ----------------------------------------------------------------------------- steps
def param_checksums(model, out, key):
    """Post-step parameters AND the gradients that produced the step: training_step leaves them in
    p.grad (the reference zeroes grads at the START of a step: Networks.py:375, :944, :1994, :2025).
    Gradients are the well-conditioned parity target; Adam's first update is sign(g)*lr, so an
    element whose gradient is at rounding-noise level moves by +-lr in either implementation."""
    for pn, p in model.state_dict().items():
        out[f"{key}/ck.{pn}"] = checksum(p)
    for pn, p in model.named_parameters():
        if p.grad is not None:
            out[f"{key}/gck.{pn}"] = checksum(p.grad)


def fp64_truth(ctor, key, batch, eps, out):
    """The same first step in float64: its gradients are the exact values the fp32 paths approximate.
    Stored as gck64.*; |gck - gck64| is the REFERENCE's own fp32 error, which calibrates how close
    another fp32 implementation can be expected to land (tests/conftest.py: assert_grad_checksum)."""
    model = ctor()
    load_synth_params(model, SEED, 0.02, prefix=key + ".")
    model = model.double()
    model.configure_optimizers(lr=LR)
    model.configure_loss(**LAMBDAS)
    model.train()
    with EpsInjector(eps):
        model.training_step({k: v.double() for k, v in batch.items()})
    for pn, p in model.named_parameters():
        if p.grad is not None:
            out[f"{key}@step1/gck64.{pn}"] = checksum(p.grad)


def gen_steps(N, out, meta):
    torch.set_num_threads(8)
    # Autoencoder, cfg1-shaped (64x64), batch 2, two steps
    key = "ae64"
    model = N.Autoencoder()
    load_synth_params(model, SEED, 0.02, prefix=key + ".")
    model.configure_optimizers(lr=LR)
    model.configure_loss(**LAMBDAS)
    model.train()
    ms = []
    for step in range(2):
        x, _ = synth.batch(2, 64, SEED, step=step)
        xb = torch.from_numpy(x)
        if step == 0:
            with torch.no_grad():
                out[key + "/out0"] = model(xb)[:, :, ::4, ::4].numpy()
        ms.append(model.training_step({"x": xb, "y": xb}))
        if step == 0:
            param_checksums(model, out, key + "@step1")
            fp64_truth(N.Autoencoder, key, {"x": xb, "y": xb}, [], out)
    meta[key] = ms
    param_checksums(model, out, key)
    print(key, ms)

    # VAE latent 64, 64x64, batch 2, two steps
    key = "vae64"
    model = N.VariationalAutoencoder(latent_dim=64)
    load_synth_params(model, SEED, 0.02, prefix=key + ".")
    model.configure_optimizers(lr=LR)
    model.configure_loss(**LAMBDAS)
    model.train()
    ms = []
    for step in range(2):
        x, _ = synth.batch(2, 64, SEED, step=step)
        xb = torch.from_numpy(x)
        eps = synth.eps_list(1, (2, 64, 4, 4), SEED, step=step)
        if step == 0:
            with torch.no_grad(), EpsInjector(eps):
                o, mu, lv = model(xb)
                out[key + "/out0"] = o[:, :, ::4, ::4].numpy()
                out[key + "/mu0"], out[key + "/logvar0"] = mu.numpy(), lv.numpy()
        with EpsInjector(eps):
            ms.append(model.training_step({"x": xb, "y": xb}))
        if step == 0:
            param_checksums(model, out, key + "@step1")
            fp64_truth(lambda: N.VariationalAutoencoder(latent_dim=64), key, {"x": xb, "y": xb}, eps, out)
    meta[key] = ms
    param_checksums(model, out, key)
    print(key, ms)

    # CycleVAEGAN at 256x256, batch 1: unpaired (the summer2winter configuration) two steps, paired one step
    for key, paired, nsteps in (("cvg256_unpaired", False, 2), ("cvg256_paired", True, 1)):
        model = N.CycleVAEGAN(latent_dim=64, paired=paired)
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.train()
        ms = []
        for step in range(nsteps):
            x, y = synth.batch(1, 256, SEED, step=step)
            xb, yb = torch.from_numpy(x), torch.from_numpy(y)
            eps = synth.eps_list(6, (1, 64, 16, 16), SEED, step=step)
            if step == 0:
                with torch.no_grad(), EpsInjector(eps):
                    fw = model(xb, yb)
                    for nm, t in zip(("Gx", "FGx", "Fy", "GFy"), fw[:4]):
                        out[f"{key}/{nm}0"] = t[:, :, ::16, ::16].numpy()
                    out[key + "/mu_x0"] = fw[4][:, ::8].numpy()
                    out[key + "/logvar_x0"] = fw[5][:, ::8].numpy()
                    out[key + "/D0"] = torch.stack([fw[12], fw[13], fw[14], fw[15]]).numpy()
                # the no-grad forward above ran the spectral-norm power iteration once; harmless (fixed point)
            with EpsInjector(eps):
                ms.append(model.training_step({"x": xb, "y": yb}))
            if step == 0:
                param_checksums(model, out, key + "@step1")     # single-step parity snapshot
                fp64_truth(lambda: N.CycleVAEGAN(latent_dim=64, paired=paired), key, {"x": xb, "y": yb}, eps, out)
            print(key, step, ms[-1])
        meta[key] = ms
        param_checksums(model, out, key)


def gen_validation(N, out, meta):
    """`validation_step` in eval mode, as `train.validate` runs it (reference train.py:131-171: model.eval(), no_grad):
    forward-only metrics plus the Gx / Fy images.  Eval mode matters for the discriminators: spectral_norm does no
    power iteration and normalises by sigma = u . (W v) with the STORED u, v (here the synthetic, non-converged ones)."""
    torch.set_num_threads(8)
    for key, ctor, S, B in (("ae64", N.Autoencoder, 64, 2), ("vae64", lambda: N.VariationalAutoencoder(latent_dim=64), 64, 2)):
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)          # VariationalAutoencoder.validation_step insists on one (Networks.py:963)
        model.configure_loss(**LAMBDAS)
        model.eval()
        x, y = synth.batch(B, S, SEED, step=7)
        eps = synth.eps_list(1, (B, 64, S // 16, S // 16), SEED, step=7)
        with EpsInjector(eps):
            m = model.validation_step({"x": torch.from_numpy(x), "y": torch.from_numpy(y)})
        out[key + "/Gx"] = m.pop("Gx")[:, :, ::4, ::4].numpy()
        meta[key] = m
        print(key, m)
    for key, paired in (("cvg256_unpaired", False), ("cvg256_paired", True)):
        model = N.CycleVAEGAN(latent_dim=64, paired=paired)
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.eval()
        x, y = synth.batch(1, 256, SEED, step=7)
        eps = synth.eps_list(6, (1, 64, 16, 16), SEED, step=7)
        u0 = model.DX.model[4].weight_u.clone()
        with EpsInjector(eps):
            m = model.validation_step({"x": torch.from_numpy(x), "y": torch.from_numpy(y)})
        assert torch.equal(u0, model.DX.model[4].weight_u), "eval mode must leave the spectral-norm vectors alone"
        out[key + "/Gx"] = m.pop("Gx")[:, :, ::16, ::16].numpy()
        out[key + "/Fy"] = m.pop("Fy")[:, :, ::16, ::16].numpy()
        meta[key] = m
        print(key, m)


def gen_cycleaegan(N, out, meta):
    """CycleAEGAN (Networks.py:1618-1869) at 256x256, batch 1: one training step (unpaired and paired)
    from synthetic parameters, and `validation_step` in eval mode on another batch."""
    torch.set_num_threads(8)
    for key, paired in (("cag256_unpaired", False), ("cag256_paired", True)):
        model = N.CycleAEGAN(paired=paired)
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.eval()
        x, y = synth.batch(1, 256, SEED, step=7)
        m = model.validation_step({"x": torch.from_numpy(x), "y": torch.from_numpy(y)})
        out[key + "/val_Gx"] = m.pop("Gx")[:, :, ::16, ::16].numpy()
        out[key + "/val_Fy"] = m.pop("Fy")[:, :, ::16, ::16].numpy()
        meta[key + "/validation"] = m
        model.train()
        x, y = synth.batch(1, 256, SEED, step=0)
        xb, yb = torch.from_numpy(x), torch.from_numpy(y)
        with torch.no_grad():
            fw = model(xb, yb)
            assert len(fw) == 10
            for nm, t in zip(("Gx", "FGx", "Fy", "GFy"), fw[:4]):
                out[f"{key}/{nm}0"] = t[:, :, ::16, ::16].numpy()
            out[key + "/D0"] = torch.stack([fw[4], fw[5], fw[6], fw[7]]).numpy()
        meta[key] = [model.training_step({"x": xb, "y": yb})]
        param_checksums(model, out, key + "@step1")
        fp64_truth(lambda: N.CycleAEGAN(paired=paired), key, {"x": xb, "y": yb}, [], out)
        print(key, meta[key], meta[key + "/validation"])


def gen_cycle_nogan(N, out, meta):
    """CycleAE (Networks.py:1350-1480) and CycleVAE (:1482-1616), SURVEY.md §8f.3: no discriminator, so 64x64 inputs
    do (batch 2).  One training step from synthetic parameters (fp32 + fp64 gradients) and the eval-mode validation step,
    unpaired and paired.  CycleVAE draws eps four times per forward: G(x), F(G(x)), F(y), G(F(y)) (:1489-1494)."""
    torch.set_num_threads(8)
    for name, ctor, nv in (("cae64", lambda p: N.CycleAE(paired=p), 0), ("cve64", lambda p: N.CycleVAE(latent_dim=64, paired=p), 4)):
        for paired in (False, True):
            key = f"{name}_{'paired' if paired else 'unpaired'}"
            model = ctor(paired)
            load_synth_params(model, SEED, 0.02, prefix=key + ".")
            model.configure_optimizers(lr=LR)
            model.configure_loss(**LAMBDAS)
            model.eval()
            x, y = (torch.from_numpy(a) for a in synth.batch(2, 64, SEED, step=7))
            with EpsInjector(synth.eps_list(nv, (2, 64, 4, 4), SEED, step=7)):
                m = model.validation_step({"x": x, "y": y})
            out[key + "/val_Gx"] = m.pop("Gx")[:, :, ::4, ::4].numpy()
            out[key + "/val_Fy"] = m.pop("Fy")[:, :, ::4, ::4].numpy()
            meta[key + "/validation"] = m
            model.train()
            x, y = (torch.from_numpy(a) for a in synth.batch(2, 64, SEED, step=0))
            eps = synth.eps_list(nv, (2, 64, 4, 4), SEED, step=0)
            with EpsInjector(eps):
                meta[key] = [model.training_step({"x": x, "y": y})]
            param_checksums(model, out, key + "@step1")
            fp64_truth(lambda: ctor(paired), key, {"x": x, "y": y}, eps, out)
            print(key, meta[key], meta[key + "/validation"])


def gen_double(N, out, meta):
    """DoubleAutoencoder (Networks.py:415-606) and DoubleVariationalAutoencoder (:608-852), SURVEY.md §8f.3, at 64x64,
    batch 2: one training step and the eval-mode validation step.  The VAE draws eps twice per forward (block A, block
    B) and once more per translation in validation_step."""
    torch.set_num_threads(8)
    for key, ctor, ntrain, nval in (("dae64", N.DoubleAutoencoder, 0, 0), ("dve64", lambda: N.DoubleVariationalAutoencoder(latent_dim=64), 2, 4)):
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.eval()
        x, y = (torch.from_numpy(a) for a in synth.batch(2, 64, SEED, step=7))
        with EpsInjector(synth.eps_list(nval, (2, 64, 4, 4), SEED, step=7)):
            m = model.validation_step({"x": x, "y": y})
        out[key + "/val_Gx"] = m.pop("Gx")[:, :, ::4, ::4].numpy()
        out[key + "/val_Fy"] = m.pop("Fy")[:, :, ::4, ::4].numpy()
        meta[key + "/validation"] = m
        model.train()
        x, y = (torch.from_numpy(a) for a in synth.batch(2, 64, SEED, step=0))
        eps = synth.eps_list(ntrain, (2, 64, 4, 4), SEED, step=0)
        with EpsInjector(eps):
            meta[key] = [model.training_step({"x": x, "y": y})]
        param_checksums(model, out, key + "@step1")
        fp64_truth(ctor, key, {"x": x, "y": y}, eps, out)
        print(key, meta[key], meta[key + "/validation"])


def gen_single_gan(N, out, meta):
    """AEGAN (Networks.py:991-1188) and VAEGAN (:1190-1348), SURVEY.md §8f.3: one generator, one discriminator; 256x256,
    batch 1.  One training step and the eval-mode validation step.  VAEGAN draws eps for G(x) then G(y) (:1203-1208)."""
    torch.set_num_threads(8)
    for key, ctor, ne in (("aeg256", N.AEGAN, 0), ("vag256", lambda: N.VAEGAN(latent_dim=64), 2)):
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.eval()
        x, y = (torch.from_numpy(a) for a in synth.batch(1, 256, SEED, step=7))
        with EpsInjector(synth.eps_list(ne, (1, 64, 16, 16), SEED, step=7)):
            m = model.validation_step({"x": x, "y": y})
        out[key + "/val_Gx"] = m.pop("Gx")[:, :, ::16, ::16].numpy()
        meta[key + "/validation"] = m
        model.train()
        x, y = (torch.from_numpy(a) for a in synth.batch(1, 256, SEED, step=0))
        eps = synth.eps_list(ne, (1, 64, 16, 16), SEED, step=0)
        with EpsInjector(eps):
            meta[key] = [model.training_step({"x": x, "y": y})]
        param_checksums(model, out, key + "@step1")
        fp64_truth(ctor, key, {"x": x, "y": y}, eps, out)
        print(key, meta[key], meta[key + "/validation"])


def gen_steps_fp64(N, meta):
    """The two unpaired CycleVAEGAN steps of `steps` again in float64: what the step-2 metrics would be without fp32
    rounding.  |fp32 - fp64| of the REFERENCE calibrates how far a second GAN step of another fp32 implementation may
    land (Adam's first update is lr * sign(g): wherever a gradient element is rounding noise — the reference's own fp32
    gradient of D's spectral-normed weight is 77 % away from its fp64 one — the two runs take opposite steps)."""
    torch.set_num_threads(8)
    key = "cvg256_unpaired"
    model = N.CycleVAEGAN(latent_dim=64, paired=False)
    load_synth_params(model, SEED, 0.02, prefix=key + ".")
    model = model.double()
    model.configure_optimizers(lr=LR)
    model.configure_loss(**LAMBDAS)
    model.train()
    ms = []
    for step in range(2):
        x, y = synth.batch(1, 256, SEED, step=step)
        eps = synth.eps_list(6, (1, 64, 16, 16), SEED, step=step)
        with EpsInjector([e.astype(np.float64) for e in eps]):
            ms.append(model.training_step({"x": torch.from_numpy(x).double(), "y": torch.from_numpy(y).double()}))
        print(key, "fp64 step", step, ms[-1])
    meta[key] = ms


def gen_dp_batch2(N, out, meta):
    """SURVEY.md §8e's DDP parity fixture: the reference's CycleVAEGAN.training_step (Networks.py:1973-2078) on a batch of TWO
    256 x 256 pairs — what a 2-rank x batch-1 data-parallel step must reproduce in its rank-averaged metrics and its
    world-averaged gradients (tests/test_gpu_parity.py::test_two_rank_step_equals_the_big_batch_step).  Same parameters
    ("dp." keys), images (synth.batch(2, 256, SEED, step=3)) and eps (step=3) as that test; fp32 and fp64 runs."""
    torch.set_num_threads(8)
    key = "dp"
    x, y = synth.batch(2, 256, SEED, step=3)
    eps = synth.eps_list(6, (2, 64, 16, 16), SEED, step=3)
    for dt, tag in ((torch.float32, ""), (torch.float64, "_fp64")):
        model = N.CycleVAEGAN(latent_dim=64, paired=False)
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model = model.to(dt)
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.train()
        with EpsInjector([e.astype(np.float64 if dt == torch.float64 else np.float32) for e in eps]):
            m = model.training_step({"x": torch.from_numpy(x).to(dt), "y": torch.from_numpy(y).to(dt)})
        meta[key + tag] = [m]
        for pn, p in model.named_parameters():
            if p.grad is not None:
                out[f"{key}@step1/{'gck64' if tag else 'gck'}.{pn}"] = checksum(p.grad)
        print(key + tag, m)


def gen_vae1024(N, out, meta):
    """BASELINE.json configs[2] is `vae` with latent_dim 1024 (the reference CLI cannot reach it; the class can: Networks.py:856).
    64x64, batch 2, ONE training step from synthetic parameters (fp32 + fp64 gradients) and the eval-mode validation
    step: this is the only fixture in which the 1024 -> 1024 bare convs of the bottleneck (Networks.py:214-237) run."""
    torch.set_num_threads(8)
    key = "vae1024"
    ctor = lambda: N.VariationalAutoencoder(latent_dim=1024)   # noqa: E731
    model = ctor()
    load_synth_params(model, SEED, 0.02, prefix=key + ".")
    model.configure_optimizers(lr=LR)
    model.configure_loss(**LAMBDAS)
    model.eval()
    x, y = (torch.from_numpy(a) for a in synth.batch(2, 64, SEED, step=7))
    with EpsInjector(synth.eps_list(1, (2, 1024, 4, 4), SEED, step=7)):
        m = model.validation_step({"x": x, "y": y})
    out[key + "/val_Gx"] = m.pop("Gx")[:, :, ::4, ::4].numpy()
    meta[key + "/validation"] = m
    model.train()
    x, _ = synth.batch(2, 64, SEED, step=0)
    xb = torch.from_numpy(x)
    eps = synth.eps_list(1, (2, 1024, 4, 4), SEED, step=0)
    with torch.no_grad(), EpsInjector(eps):
        o, mu, lv = model(xb)
        out[key + "/out0"] = o[:, :, ::4, ::4].numpy()
        out[key + "/mu0"], out[key + "/logvar0"] = mu[:, ::16].numpy(), lv[:, ::16].numpy()
    with EpsInjector(eps):
        meta[key] = [model.training_step({"x": xb, "y": xb})]
    param_checksums(model, out, key + "@step1")
    fp64_truth(ctor, key, {"x": xb, "y": xb}, eps, out)
    print(key, meta[key], meta[key + "/validation"])


def gen_headline(N, out, meta):
    """ONE reference training_step at each BASELINE.json configuration itself (VERDICT r3 missing #2): the planners of the
    HIP path pick other tiles, split-K factors, stream-K runs and Winograd gates at batch 8 / 16 than at the batch-1 / 2
    fixtures above, so the benchmarked shapes are pinned end to end by the reference's own numbers:
      cvg256_b8    CycleVAEGAN(latent 64, unpaired), 8 x 3 x 256 x 256          (configs[3] / [4]; Networks.py:1973-2078)
      ae256_b16    Autoencoder, 16 x 3 x 256 x 256                               (configs[1]; :334-384)
      vae1024_b16  VariationalAutoencoder(latent_dim=1024), 16 x 3 x 256 x 256   (configs[2]; :918-953)
      ae64_b4      Autoencoder, 4 x 3 x 64 x 64                                  (configs[0])
    Kept per config: the metric dict, strided output slices of a no-grad forward, post-step parameter checksums and the
    gradient checksums in fp32 and (a second, float64 run) fp64.  ~10 minutes and ~35 GB in the build container."""
    import time
    torch.set_num_threads(8)
    cfgs = (("ae64_b4", N.Autoencoder, 4, 64, 0, None, False),
            ("ae256_b16", N.Autoencoder, 16, 256, 0, None, False),
            ("vae1024_b16", lambda: N.VariationalAutoencoder(latent_dim=1024), 16, 256, 1, 1024, False),
            ("cvg256_b8", lambda: N.CycleVAEGAN(latent_dim=64, paired=False), 8, 256, 6, 64, True))
    for key, ctor, B, S, ne, lat, xy in cfgs:
        t0 = time.time()
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.train()
        x, y = synth.batch(B, S, SEED, step=0)
        xb = torch.from_numpy(x)
        yb = torch.from_numpy(y) if xy else xb
        eps = synth.eps_list(ne, (B, lat, S // 16, S // 16), SEED, step=0) if ne else []
        st = S // 16
        with torch.no_grad(), EpsInjector(eps):
            if xy:
                fw = model(xb, yb)
                for nm, t in zip(("Gx", "FGx", "Fy", "GFy"), fw[:4]):
                    out[f"{key}/{nm}0"] = t[:, :, ::st, ::st].numpy()
                out[key + "/mu_x0"] = fw[4][:, ::8, ::2, ::2].numpy()
                out[key + "/logvar_x0"] = fw[5][:, ::8, ::2, ::2].numpy()
                out[key + "/D0"] = torch.stack([fw[12], fw[13], fw[14], fw[15]]).numpy()
            else:
                fw = model(xb)
                o = fw[0] if isinstance(fw, tuple) else fw
                out[key + "/out0"] = o[:, :, ::st, ::st].numpy()
                if isinstance(fw, tuple):
                    out[key + "/mu0"] = fw[1][:, ::64, ::2, ::2].numpy()
                    out[key + "/logvar0"] = fw[2][:, ::64, ::2, ::2].numpy()
        with EpsInjector(eps):
            meta[key] = [model.training_step({"x": xb, "y": yb})]
        param_checksums(model, out, key + "@step1")
        del model
        print(key, "fp32", meta[key], f"{time.time() - t0:.0f} s", flush=True)
        fp64_truth(ctor, key, {"x": xb, "y": yb}, [e.astype(np.float64) for e in eps], out)
        print(key, "fp64 done", f"{time.time() - t0:.0f} s", flush=True)


def import_reference_train():
    """/root/reference/train.py needs two more stubs than Networks.py: torch.utils.tensorboard (absent here) and the
    torchvision names Data_Manager.py touches at import time (none: it only imports the module)."""
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    import_reference()
    import train as ref_train  # noqa
    return sys.modules["train"]


def gen_train_epoch(N, out, meta):
    """The reference's own `train_epoch` (train.py:80-128) on a list of two synthetic batches: the averaged metric tuple and
    a slice of `last_output`.  Its extra train-mode forward per batch (:112-117) draws eps too, so the stream of the SECOND
    batch's training_step starts after it: VAE eps order = step0, viz0, step1, viz1; CycleVAEGAN 6 + 6 per batch."""
    import argparse
    T = import_reference_train()
    torch.set_num_threads(8)
    args = argparse.Namespace()
    for key, ctor, S, B, ne, xy in (("ae64", N.Autoencoder, 64, 2, 0, False),
                                    ("vae64", lambda: N.VariationalAutoencoder(latent_dim=64), 64, 2, 1, False),
                                    ("cvg256_unpaired", lambda: N.CycleVAEGAN(latent_dim=64, paired=False), 256, 1, 6, True)):
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        batches = []
        for step in range(2):
            x, y = synth.batch(B, S, SEED, step=step)
            batches.append({"x": torch.from_numpy(x), "y": torch.from_numpy(y if xy else x)})
        lat = (B, 64, S // 16, S // 16)
        eps = synth.eps_list(4 * ne, lat, SEED, step=100)       # [batch 0: step, viz][batch 1: step, viz]
        with EpsInjector(eps) as inj:
            avg_loss, comps, last_output, last_x, last_y = T.train_epoch(model, batches, torch.device("cpu"), args)
        assert inj.i == 4 * ne, (inj.i, ne)
        st = 16 if S == 256 else 4
        lo = last_output if last_output.dim() == 4 else last_output[None]      # Autoencoder: model(x)[0] is ONE image
        out[f"{key}/last_output"] = lo[:, :, ::st, ::st].numpy()
        meta[key] = {"avg_loss": avg_loss, "components": comps, "last_output_shape": list(last_output.shape),
                     "eps_draws": inj.i}
        # the same epoch in float64: `last_output` is a forward AFTER two Adam updates, and Adam's early updates are
        # lr * sign(g) — every gradient element at rounding-noise level steps the other way in another fp32 run — so the
        # reference's own fp32 `last_output` sits tens of percent from its fp64 one.  The fp64 copy calibrates the test.
        model = ctor()
        load_synth_params(model, SEED, 0.02, prefix=key + ".")
        model = model.double()
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        b64 = [{k: v.double() for k, v in b.items()} for b in batches]
        with EpsInjector([e.astype(np.float64) for e in eps]):
            avg64, comps64, lo64, _, _ = T.train_epoch(model, b64, torch.device("cpu"), args)
        lo64 = lo64 if lo64.dim() == 4 else lo64[None]
        out[f"{key}/last_output64"] = lo64[:, :, ::st, ::st].numpy()
        meta[key]["avg_loss64"], meta[key]["components64"] = avg64, comps64
        print("train_epoch", key, meta[key])


def gen_checkpoint_skeleton(N):
    """Structure of the checkpoints the REFERENCE writes (utils.py:17-28: torch.save of {epoch, model_state_dict,
    optimizer_states, loss, args}) after one training step: key names, shapes and dtypes of every tensor, the optimizer
    state layout.  Values are pinned elsewhere (steps.npz); files of 0.3-1.7 GB cannot be committed, their skeleton can."""
    import argparse
    import tempfile
    utils = importlib.import_module("utils")          # /root/reference/utils.py
    torch.set_num_threads(8)

    def describe(v):
        if isinstance(v, torch.Tensor):
            return {"tensor": list(v.shape), "dtype": str(v.dtype)}
        if isinstance(v, dict):
            return {str(k): describe(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [describe(x) for x in v]
        return {"py": type(v).__name__, "value": v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)}

    skel = {}
    for key, ctor, S in (("autoencoder", N.Autoencoder, 64), ("vae", lambda: N.VariationalAutoencoder(latent_dim=64), 64),
                         ("cyclevaegan", lambda: N.CycleVAEGAN(latent_dim=64, paired=False), 256)):
        model = ctor()
        model.configure_optimizers(lr=LR)
        model.configure_loss(**LAMBDAS)
        model.train()
        x, y = synth.batch(1, S, SEED, step=0)
        m = model.training_step({"x": torch.from_numpy(x), "y": torch.from_numpy(y if key == "cyclevaegan" else x)})
        args = argparse.Namespace(architecture=key, lr=LR, batch_size=1)
        with tempfile.TemporaryDirectory() as d:
            fn = os.path.join(d, "ck.pth")
            utils.save_checkpoint(model, 3, m["G_loss"], args, fn)
            ck = torch.load(fn, map_location="cpu", weights_only=False)
        skel[key] = describe(ck)
        skel[key]["loss"]["value"] = None           # a run-dependent number
        print(key, "checkpoint keys:", list(ck), "state_dict tensors:", len(ck["model_state_dict"]))
    return skel


def main():
    N, _ = import_reference()
    torch.manual_seed(0)
    atoms, steps, meta = {}, {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS,
                                  "torch": torch.__version__, "reference": "Baverne/VAE-CYCLEGAN-Implementation"}
    which = sys.argv[1:] or ["atoms", "steps", "validation", "cycleaegan", "cycle_nogan", "double", "single_gan", "steps_fp64", "vae1024", "train_epoch",
                                 "dp_batch2", "headline", "checkpoint"]
    if "atoms" in which:
        gen_atoms(N, atoms)
        np.savez_compressed(os.path.join(HERE, "atoms.npz"), **atoms)
    if "steps" in which:
        gen_steps(N, steps, meta)
        np.savez_compressed(os.path.join(HERE, "steps.npz"), **steps)
        with open(os.path.join(HERE, "steps_meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
    if "validation" in which:
        val, vmeta = {}, {"seed": SEED, "lambdas": LAMBDAS, "torch": torch.__version__, "batch_step": 7}
        gen_validation(N, val, vmeta)
        np.savez_compressed(os.path.join(HERE, "validation.npz"), **val)
        with open(os.path.join(HERE, "validation_meta.json"), "w") as f:
            json.dump(vmeta, f, indent=1)
    if "cycleaegan" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_cycleaegan(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "cycleaegan.npz"), **arr)
        with open(os.path.join(HERE, "cycleaegan_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "cycle_nogan" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_cycle_nogan(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "cycle_nogan.npz"), **arr)
        with open(os.path.join(HERE, "cycle_nogan_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "double" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_double(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "double.npz"), **arr)
        with open(os.path.join(HERE, "double_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "single_gan" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_single_gan(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "single_gan.npz"), **arr)
        with open(os.path.join(HERE, "single_gan_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "steps_fp64" in which:
        cmeta = {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_steps_fp64(N, cmeta)
        with open(os.path.join(HERE, "steps_fp64_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "vae1024" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__}
        gen_vae1024(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "vae1024.npz"), **arr)
        with open(os.path.join(HERE, "vae1024_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "train_epoch" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__, "eps_stream_step": 100}
        gen_train_epoch(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "train_epoch.npz"), **arr)
        with open(os.path.join(HERE, "train_epoch_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "dp_batch2" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__, "batch": 2, "batch_step": 3}
        gen_dp_batch2(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "dp_batch2.npz"), **arr)
        with open(os.path.join(HERE, "dp_batch2_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "headline" in which:
        arr, cmeta = {}, {"seed": SEED, "lr": LR, "lambdas": LAMBDAS, "torch": torch.__version__, "batch_step": 0}
        gen_headline(N, arr, cmeta)
        np.savez_compressed(os.path.join(HERE, "headline.npz"), **arr)
        with open(os.path.join(HERE, "headline_meta.json"), "w") as f:
            json.dump(cmeta, f, indent=1)
    if "checkpoint" in which:
        with open(os.path.join(HERE, "checkpoint_skeleton.json"), "w") as f:
            json.dump(gen_checkpoint_skeleton(N), f, indent=0)
    for fn in ("atoms.npz", "steps.npz", "steps_meta.json", "validation.npz", "validation_meta.json", "cycleaegan.npz",
               "cycleaegan_meta.json", "cycle_nogan.npz", "cycle_nogan_meta.json", "double.npz",
               "double_meta.json", "single_gan.npz", "single_gan_meta.json", "vae1024.npz", "vae1024_meta.json", "train_epoch.npz",
               "train_epoch_meta.json", "dp_batch2.npz", "dp_batch2_meta.json", "headline.npz", "headline_meta.json", "checkpoint_skeleton.json"):
        p = os.path.join(HERE, fn)
        if os.path.exists(p):
            print(fn, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
