"""ORACLE — test infrastructure, not product code.

A CPU restatement, in plain functional fp32 PyTorch, of the reference's training-step hot path
(Baverne/VAE-CYCLEGAN-Implementation: Networks.py, Losses.py, and torch.optim.Adam as called from
them).  Each function cites the reference lines it follows.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module; the
product (`vae-cyclegan-implementation_amd/`) never does.

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference itself, produced in the build container by
`tests/golden/make_golden.py` (which imports /root/reference) and committed as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks the oracle against them.

Parameters are a flat dict {state_dict name: tensor}, NCHW / OIHW exactly as the reference's
state_dict, so fixtures and checkpoints map one to one.
"""
import math

import torch
import torch.nn.functional as F

IN_EPS = 1e-5


# ----------------------------------------------------------------------------- atoms
def reflect_conv(x, w, b, stride=1, pad=1):
    """nn.Conv2d(..., padding_mode='reflect')  — Networks.py:60,87,101,104,122,136,145"""
    if pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    return F.conv2d(x, w, b, stride=stride)


def instance_norm(x):
    """nn.InstanceNorm2d(C): eps 1e-5, biased variance, no affine, no running stats — Networks.py:61.
    Calls the functional the reference's module calls, so the CPU numerics are the reference's
    (torch's CPU kernel accumulates the statistics and the backward reductions in double; a
    hand-written fp32 mean/var differs from it by up to 1e-2 in the gradients of near-dead
    channels, where rstd reaches 1/sqrt(eps) = 316)."""
    return F.instance_norm(x, None, None, None, None, True, 0.1, IN_EPS)


def _act(x, name):
    if name == "ReLU":
        return F.relu(x)
    if name == "LeakyReLU":
        return F.leaky_relu(x, 0.2)
    if name == "Identity":
        return x
    if name == "Tanh":
        return torch.tanh(x)
    if name == "Sigmoid":
        return torch.sigmoid(x)
    raise NotImplementedError(name)


def casb(x, P, pre, stride=1, pad=3, activation="ReLU", use_norm=True):
    """CaSb.forward: conv -> [IN] -> act — Networks.py:76-81"""
    x = reflect_conv(x, P[pre + "conv.weight"], P[pre + "conv.bias"], stride, pad)
    if use_norm:
        x = instance_norm(x)
    return _act(x, activation)


def d_block(x, P, pre):
    """D.forward: PixelUnshuffle(2) -> conv3x3 -> ReLU -> IN — Networks.py:91-96"""
    x = F.pixel_unshuffle(x, 2)
    x = reflect_conv(x, P[pre + "conv.weight"], P[pre + "conv.bias"])
    return instance_norm(F.relu(x))


def r_block(x, P, pre):
    """R.forward: conv -> ReLU -> IN -> conv -> IN -> + x — Networks.py:108-116"""
    h = reflect_conv(x, P[pre + "conv1.weight"], P[pre + "conv1.bias"])
    h = instance_norm(F.relu(h))
    h = reflect_conv(h, P[pre + "conv2.weight"], P[pre + "conv2.bias"])
    return instance_norm(h) + x


def u_block(x, P, pre):
    """U.forward: PixelShuffle(2) -> conv3x3 -> ReLU -> IN — Networks.py:126-131"""
    x = F.pixel_shuffle(x, 2)
    x = reflect_conv(x, P[pre + "conv.weight"], P[pre + "conv.bias"])
    return instance_norm(F.relu(x))


def s_conv(x, P, pre):
    """S.forward / L.forward: bare conv3x3 — Networks.py:138-140, 147-149"""
    return reflect_conv(x, P[pre + "conv.weight"], P[pre + "conv.bias"])


# ----------------------------------------------------------------------------- molecules
def encoder(x, P, pre):
    """Encoder — Networks.py:158-163"""
    x = casb(x, P, pre + "model.0.", 1, 3, "ReLU", True)
    for i in (1, 2, 3, 4):
        x = d_block(x, P, pre + f"model.{i}.")
    return r_block(x, P, pre + "model.5.")


def decoder(x, P, pre):
    """Decoder — Networks.py:187-192"""
    x = r_block(x, P, pre + "model.0.")
    for i in (1, 2, 3, 4):
        x = u_block(x, P, pre + f"model.{i}.")
    return casb(x, P, pre + "model.5.", 1, 3, "Identity", False)


def variational_encoder_block(x, P, pre, eps):
    """VariationalEncoderBlock.forward — Networks.py:219-227 (eps replaces torch.randn_like at :225)"""
    mu = s_conv(x, P, pre + "muConv.")
    logvar = s_conv(s_conv(x, P, pre + "logvarConv.0."), P, pre + "logvarConv.1.")
    logvar = torch.clamp(logvar, min=-10, max=10)
    std = torch.exp(0.5 * logvar)
    return mu + eps * std, mu, logvar


def autoencoder_forward(x, P, pre=""):
    """Autoencoder.forward — Networks.py:302-305"""
    return decoder(encoder(x, P, pre + "encoder."), P, pre + "decoder.")


def vae_forward(x, P, pre, eps):
    """VariationalAutoencoder.forward — Networks.py:885-890"""
    enc = encoder(x, P, pre + "encoder.")
    z, mu, logvar = variational_encoder_block(enc, P, pre + "variational_encoder_block.", eps)
    dec = s_conv(z, P, pre + "variational_decoder_block.conv.")
    return decoder(dec, P, pre + "decoder."), mu, logvar


def spectral_norm_weight(P, pre, training=True):
    """torch.nn.utils.spectral_norm's compute_weight (one power iteration, eps 1e-12) on
    nn.Conv2d(512, 1, 16) — Networks.py:248.  Returns (W/sigma, u_new, v_new)."""
    w = P[pre + "weight_orig"]
    u, v = P[pre + "weight_u"], P[pre + "weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.detach().t(), u), dim=0, eps=1e-12)
            u = F.normalize(torch.mv(wm.detach(), v), dim=0, eps=1e-12)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma, u, v


def discriminator(x, P, pre, training=True, sn_state=None):
    """Discriminator.forward — Networks.py:244-248, :267-269"""
    x = casb(x, P, pre + "model.0.", 2, 1, "LeakyReLU", False)
    for i in (1, 2, 3):
        x = casb(x, P, pre + f"model.{i}.", 2, 1, "LeakyReLU", True)
    w, u, v = spectral_norm_weight(P, pre + "model.4.", training)
    if sn_state is not None:
        sn_state[pre + "model.4.weight_u"] = u.detach()
        sn_state[pre + "model.4.weight_v"] = v.detach()
    return F.conv2d(x, w, P[pre + "model.4.bias"]).view(-1, 1).squeeze(1)


# ----------------------------------------------------------------------------- losses
def l1(a, b):
    """nn.L1Loss() — Losses.py:21-24"""
    return (a - b).abs().mean()


def cycle_loss(x, y, FGx, GFy):
    """CycleConsistencyLoss.forward — Losses.py:36-39"""
    return l1(FGx, x) + l1(GFy, y)


def identity_loss(x, y, Fx, Gy):
    """IdentityLoss.forward — Losses.py:55-65"""
    return l1(Fx, x) + l1(Gy, y)


def gan_loss_generator(d_real, d_fake):
    """GANLossGenerator.forward — Losses.py:78-83"""
    real = (d_real ** 2).mean()
    fake = ((d_fake - 1.0) ** 2).mean()
    return real + fake, real, fake


def gan_loss_discriminator(d_real, d_fake):
    """GANLossDiscriminator.forward — Losses.py:97-102"""
    real = ((d_real - 1.0) ** 2).mean()
    fake = (d_fake ** 2).mean()
    return real + fake, real, fake


def kl_loss(mu, logvar):
    """KLDivergenceLoss.forward — Losses.py:115-121"""
    logvar = torch.clamp(logvar, min=-10, max=10)
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


# ----------------------------------------------------------------------------- optimizer
def adam_update(P, grads, state, names, lr, betas=(0.5, 0.999), eps=1e-8):
    """torch.optim.Adam.step (single-tensor path, no weight decay / amsgrad) for the named parameters;
    call sites Networks.py:312, 894, 1928-1935, steps at :377, :946, :2022, :2044."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    with torch.no_grad():
        for n in names:
            g = grads.get(n)
            if g is None:
                continue
            m = state.setdefault("exp_avg", {}).setdefault(n, torch.zeros_like(P[n]))
            v = state.setdefault("exp_avg_sq", {}).setdefault(n, torch.zeros_like(P[n]))
            m.lerp_(g, 1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / bc2_sqrt).add_(eps)
            P[n] = P[n] - step_size * (m / denom)


def _leaf_params(P, names):
    Q = dict(P)
    for n in names:
        Q[n] = P[n].detach().clone().requires_grad_(True)
    return Q


def _grads(loss, Q, names):
    gs = torch.autograd.grad(loss, [Q[n] for n in names], allow_unused=True)
    return {n: g for n, g in zip(names, gs)}


def trainable_names(P, prefixes=("",)):
    return [n for n in P if not (n.endswith("weight_u") or n.endswith("weight_v")) and any(n.startswith(p) for p in prefixes)]


# ----------------------------------------------------------------------------- training steps
def autoencoder_step(P, state, x, y, lr):
    """Autoencoder.training_step — Networks.py:334-384.  Returns (metrics, output, grads)."""
    names = trainable_names(P)
    Q = _leaf_params(P, names)
    out = autoencoder_forward(x, Q)
    loss = l1(out, y)
    grads = _grads(loss, Q, names)
    adam_update(P, grads, state, names, lr)
    v = loss.item()
    return {"G_loss": v, "loss_trans": v, "total_loss": v}, out.detach(), grads


def vae_step(P, state, x, y, eps, lr, lambda_kl=1e-5):
    """VariationalAutoencoder.training_step — Networks.py:918-953"""
    names = trainable_names(P)
    Q = _leaf_params(P, names)
    out, mu, logvar = vae_forward(x, Q, "", eps)
    loss_trans = l1(out, y)
    loss_kl = kl_loss(mu, logvar)
    G_loss = loss_trans + lambda_kl * loss_kl
    grads = _grads(G_loss, Q, names)
    adam_update(P, grads, state, names, lr)
    return ({"G_loss": G_loss.item(), "loss_trans": loss_trans.item(), "loss_kl": loss_kl.item()},
            {"Gx": out.detach(), "mu": mu.detach(), "logvar": logvar.detach()}, grads)


def cyclevaegan_step(P, state, x, y, eps6, lr, paired=False, lambda_cycle=10.0, lambda_gan=1.0, lambda_kl=1e-5,
                     lambda_identity=5.0):
    """CycleVAEGAN.training_step — Networks.py:1973-2078, with forward :1909-1924 (six VAE forwards in the
    reference's order, so eps6[i] is the i-th randn_like draw) and the discriminator re-forward :2028-2035.
    `state` holds {"G": adam state, "D": adam state}.  Spectral-norm u/v in P are updated in place
    (torch does so in every training-mode forward)."""
    g_names = trainable_names(P, ("F.", "G."))
    d_names = trainable_names(P, ("DX.", "DY."))
    Q = _leaf_params(P, g_names + d_names)
    sn = {}
    Gx, mu_x, lv_x = vae_forward(x, Q, "G.", eps6[0])
    Gy, _, _ = vae_forward(y, Q, "G.", eps6[1])
    FGx, mu_FGx, lv_FGx = vae_forward(Gx, Q, "F.", eps6[2])
    Fy, mu_y, lv_y = vae_forward(y, Q, "F.", eps6[3])
    Fx, _, _ = vae_forward(x, Q, "F.", eps6[4])
    GFy, mu_GFy, lv_GFy = vae_forward(Fy, Q, "G.", eps6[5])

    def run_d(inp, pre):
        out = discriminator(inp, Q, pre, True, sn)
        Q[pre + "model.4.weight_u"], Q[pre + "model.4.weight_v"] = sn[pre + "model.4.weight_u"], sn[pre + "model.4.weight_v"]
        return out

    DYGx = run_d(Gx, "DY.")
    DXFy = run_d(Fy, "DX.")
    DXx = run_d(x, "DX.")
    DYy = run_d(y, "DY.")
    loss_cycle = cycle_loss(x, y, FGx, GFy)
    _, g_x_real, g_x_fake = gan_loss_generator(DXx, DXFy)
    _, g_y_real, g_y_fake = gan_loss_generator(DYy, DYGx)
    loss_gan_g_fake = g_x_fake + g_y_fake
    loss_kl = kl_loss(mu_x, lv_x) + kl_loss(mu_FGx, lv_FGx) + kl_loss(mu_y, lv_y) + kl_loss(mu_GFy, lv_GFy)
    G_loss = lambda_cycle * loss_cycle + lambda_gan * loss_gan_g_fake + lambda_kl * loss_kl
    if paired:
        loss_identity = identity_loss(x, y, Fx, Gy)
        G_loss = G_loss + lambda_identity * loss_identity
    g_grads = _grads(G_loss, Q, g_names)
    adam_update(P, g_grads, state.setdefault("G", {}), g_names, lr)

    DYGx_d = run_d(Gx.detach(), "DY.")
    DXFy_d = run_d(Fy.detach(), "DX.")
    DXx_d = run_d(x, "DX.")
    DYy_d = run_d(y, "DY.")
    d_x, d_x_real, d_x_fake = gan_loss_discriminator(DXx_d, DXFy_d)
    d_y, d_y_real, d_y_fake = gan_loss_discriminator(DYy_d, DYGx_d)
    D_loss = d_x + d_y
    d_grads = _grads(D_loss, Q, d_names)
    adam_update(P, d_grads, state.setdefault("D", {}), d_names, lr)
    for k, v in sn.items():
        P[k] = v

    m = {
        "total_loss": G_loss.item() + D_loss.item(), "G_loss": G_loss.item(), "D_loss": D_loss.item(),
        "D_loss_x_real": d_x_real.item(), "D_loss_x_fake": d_x_fake.item(),
        "D_loss_y_real": d_y_real.item(), "D_loss_y_fake": d_y_fake.item(),
        "loss_cycle": loss_cycle.item(), "loss_gan_g": loss_gan_g_fake.item(),
        "loss_gan_g_x_real": g_x_real.item(), "loss_gan_g_x_fake": g_x_fake.item(),
        "loss_gan_g_y_real": g_y_real.item(), "loss_gan_g_y_fake": g_y_fake.item(),
        "loss_kl": loss_kl.item(),
        "d_x_real_mean": DXx_d.mean().item(), "d_x_fake_mean": DXFy_d.mean().item(),
        "d_y_real_mean": DYy_d.mean().item(), "d_y_fake_mean": DYGx_d.mean().item(),
    }
    if paired:
        m["loss_identity"] = loss_identity.item()
    outs = {"Gx": Gx.detach(), "FGx": FGx.detach(), "Fy": Fy.detach(), "GFy": GFy.detach(),
            "mu_x": mu_x.detach(), "logvar_x": lv_x.detach()}
    return m, outs, g_grads, d_grads


# ----------------------------------------------------------------------------- validation (forward only, eval mode)
def autoencoder_validation(P, x, y):
    """Autoencoder.validation_step — Networks.py:386-413"""
    with torch.no_grad():
        out = autoencoder_forward(x, P)
        v = l1(out, y).item()
    return {"G_loss": v, "total_loss": v, "loss_trans": v}, {"Gx": out}


def vae_validation(P, x, y, eps, lambda_kl=1e-5):
    """VariationalAutoencoder.validation_step — Networks.py:955-988 (randn_like is drawn in eval mode too, :225)"""
    with torch.no_grad():
        out, mu, logvar = vae_forward(x, P, "", eps)
        loss_trans = l1(out, y)
        loss_kl = kl_loss(mu, logvar)
        G_loss = loss_trans + lambda_kl * loss_kl
    return {"G_loss": G_loss.item(), "loss_trans": loss_trans.item(), "loss_kl": loss_kl.item()}, {"Gx": out}


def cyclevaegan_validation(P, x, y, eps6, paired=False, lambda_cycle=10.0, lambda_gan=1.0, lambda_kl=1e-5,
                           lambda_identity=5.0):
    """CycleVAEGAN.validation_step — Networks.py:2080-2150 under model.eval() (train.py:133): the six VAE forwards of
    :1909-1924 in the reference's order, discriminators with the STORED spectral-norm vectors (no power iteration in
    eval mode, torch/nn/utils/spectral_norm.py compute_weight(do_power_iteration=False))."""
    with torch.no_grad():
        Gx, mu_x, lv_x = vae_forward(x, P, "G.", eps6[0])
        Gy, _, _ = vae_forward(y, P, "G.", eps6[1])
        FGx, mu_FGx, lv_FGx = vae_forward(Gx, P, "F.", eps6[2])
        Fy, mu_y, lv_y = vae_forward(y, P, "F.", eps6[3])
        Fx, _, _ = vae_forward(x, P, "F.", eps6[4])
        GFy, mu_GFy, lv_GFy = vae_forward(Fy, P, "G.", eps6[5])
        DYGx = discriminator(Gx, P, "DY.", False)
        DXFy = discriminator(Fy, P, "DX.", False)
        DXx = discriminator(x, P, "DX.", False)
        DYy = discriminator(y, P, "DY.", False)
        loss_cycle = cycle_loss(x, y, FGx, GFy)
        _, g_x_real, g_x_fake = gan_loss_generator(DXx, DXFy)
        _, g_y_real, g_y_fake = gan_loss_generator(DYy, DYGx)
        loss_gan_g_fake = g_x_fake + g_y_fake
        loss_kl = kl_loss(mu_x, lv_x) + kl_loss(mu_FGx, lv_FGx) + kl_loss(mu_y, lv_y) + kl_loss(mu_GFy, lv_GFy)
        G_loss = lambda_cycle * loss_cycle + lambda_gan * loss_gan_g_fake + lambda_kl * loss_kl
        if paired:
            loss_identity = identity_loss(x, y, Fx, Gy)
            G_loss = G_loss + lambda_identity * loss_identity
        d_x, d_x_real, d_x_fake = gan_loss_discriminator(DXx, DXFy)
        d_y, d_y_real, d_y_fake = gan_loss_discriminator(DYy, DYGx)
        D_loss = d_x + d_y
    m = {
        "total_loss": G_loss.item() + D_loss.item(), "G_loss": G_loss.item(), "D_loss": D_loss.item(),
        "D_loss_x_real": d_x_real.item(), "D_loss_x_fake": d_x_fake.item(),
        "D_loss_y_real": d_y_real.item(), "D_loss_y_fake": d_y_fake.item(),
        "loss_cycle": loss_cycle.item(), "loss_gan_g": loss_gan_g_fake.item(),
        "loss_gan_g_x_real": g_x_real.item(), "loss_gan_g_x_fake": g_x_fake.item(),
        "loss_gan_g_y_real": g_y_real.item(), "loss_gan_g_y_fake": g_y_fake.item(),
        "loss_kl": loss_kl.item(),
    }
    if paired:
        m["loss_identity"] = loss_identity.item()
    return m, {"Gx": Gx, "Fy": Fy}


# ----------------------------------------------------------------------------- CycleAEGAN
def _cycleaegan_losses(Q, x, y, paired, lambda_cycle, lambda_gan, lambda_identity, run_d):
    """Shared by the training and validation steps: forward :1654-1665 and the generator objective :1733-1753 — the
    WHOLE LSGAN generator loss (real + fake terms) enters G_loss, unlike CycleVAEGAN's fake-only sum."""
    Gx = autoencoder_forward(x, Q, "G.")
    Gy = autoencoder_forward(y, Q, "G.")
    FGx = autoencoder_forward(Gx, Q, "F.")
    Fy = autoencoder_forward(y, Q, "F.")
    Fx = autoencoder_forward(x, Q, "F.")
    GFy = autoencoder_forward(Fy, Q, "G.")
    DYGx, DXFy, DXx, DYy = run_d(Gx, "DY."), run_d(Fy, "DX."), run_d(x, "DX."), run_d(y, "DY.")
    loss_cycle = cycle_loss(x, y, FGx, GFy)
    g_x, g_x_real, g_x_fake = gan_loss_generator(DXx, DXFy)
    g_y, g_y_real, g_y_fake = gan_loss_generator(DYy, DYGx)
    loss_gan_g = g_x + g_y
    G_loss = lambda_cycle * loss_cycle + lambda_gan * loss_gan_g
    loss_identity = None
    if paired:
        loss_identity = identity_loss(x, y, Fx, Gy)
        G_loss = G_loss + lambda_identity * loss_identity
    m = {"loss_cycle": loss_cycle, "loss_gan_g": loss_gan_g, "loss_gan_g_x_real": g_x_real, "loss_gan_g_x_fake": g_x_fake,
         "loss_gan_g_y_real": g_y_real, "loss_gan_g_y_fake": g_y_fake}
    if paired:
        m["loss_identity"] = loss_identity
    return G_loss, m, (Gx, FGx, Fy, GFy), (DYGx, DXFy, DXx, DYy)


def cycleaegan_step(P, state, x, y, lr, paired=False, lambda_cycle=10.0, lambda_gan=1.0, lambda_identity=5.0):
    """CycleAEGAN.training_step — Networks.py:1711-1812 (generator update, then discriminators re-run on detached
    generator outputs :1763-1771).  Spectral-norm u/v in P are updated in place."""
    g_names = trainable_names(P, ("F.", "G."))
    d_names = trainable_names(P, ("DX.", "DY."))
    Q = _leaf_params(P, g_names + d_names)
    sn = {}

    def run_d(inp, pre):
        out = discriminator(inp, Q, pre, True, sn)
        Q[pre + "model.4.weight_u"], Q[pre + "model.4.weight_v"] = sn[pre + "model.4.weight_u"], sn[pre + "model.4.weight_v"]
        return out

    G_loss, m, (Gx, FGx, Fy, GFy), _ = _cycleaegan_losses(Q, x, y, paired, lambda_cycle, lambda_gan, lambda_identity, run_d)
    g_grads = _grads(G_loss, Q, g_names)
    adam_update(P, g_grads, state.setdefault("G", {}), g_names, lr)
    DYGx_d, DXFy_d, DXx_d, DYy_d = run_d(Gx.detach(), "DY."), run_d(Fy.detach(), "DX."), run_d(x, "DX."), run_d(y, "DY.")
    d_x, d_x_real, d_x_fake = gan_loss_discriminator(DXx_d, DXFy_d)
    d_y, d_y_real, d_y_fake = gan_loss_discriminator(DYy_d, DYGx_d)
    D_loss = d_x + d_y
    d_grads = _grads(D_loss, Q, d_names)
    adam_update(P, d_grads, state.setdefault("D", {}), d_names, lr)
    for k, v in sn.items():
        P[k] = v
    out = {"total_loss": G_loss.item() + D_loss.item(), "G_loss": G_loss.item(), "D_loss": D_loss.item(),
           "D_loss_x_real": d_x_real.item(), "D_loss_x_fake": d_x_fake.item(),
           "D_loss_y_real": d_y_real.item(), "D_loss_y_fake": d_y_fake.item()}
    out.update({k: v.item() for k, v in m.items()})
    out.update({"d_x_real_mean": DXx_d.mean().item(), "d_x_fake_mean": DXFy_d.mean().item(),
                "d_y_real_mean": DYy_d.mean().item(), "d_y_fake_mean": DYGx_d.mean().item()})
    outs = {"Gx": Gx.detach(), "FGx": FGx.detach(), "Fy": Fy.detach(), "GFy": GFy.detach()}
    return out, outs, g_grads, d_grads


def cycleaegan_validation(P, x, y, paired=False, lambda_cycle=10.0, lambda_gan=1.0, lambda_identity=5.0):
    """CycleAEGAN.validation_step — Networks.py:1814-1869 under model.eval(): stored spectral-norm vectors."""
    with torch.no_grad():
        G_loss, m, (Gx, _, Fy, _), (DYGx, DXFy, DXx, DYy) = _cycleaegan_losses(
            P, x, y, paired, lambda_cycle, lambda_gan, lambda_identity, lambda inp, pre: discriminator(inp, P, pre, False))
        d_x, d_x_real, d_x_fake = gan_loss_discriminator(DXx, DXFy)
        d_y, d_y_real, d_y_fake = gan_loss_discriminator(DYy, DYGx)
        D_loss = d_x + d_y
    out = {"total_loss": G_loss.item() + D_loss.item(), "G_loss": G_loss.item(), "D_loss": D_loss.item(),
           "D_loss_x_real": d_x_real.item(), "D_loss_x_fake": d_x_fake.item(),
           "D_loss_y_real": d_y_real.item(), "D_loss_y_fake": d_y_fake.item()}
    out.update({k: v.item() for k, v in m.items()})
    return out, {"Gx": Gx, "Fy": Fy}


# ----------------------------------------------------------------------------- CycleAE / CycleVAE
def _cycle_nogan_losses(Q, x, y, eps4, paired, lambda_cycle, lambda_kl):
    """CycleAE.forward + losses (Networks.py:1363-1368, :1413-1427) when eps4 is None, else CycleVAE's (:1489-1494,
    :1541-1559): total = lambda_cycle * cycle (+ lambda_kl * sum of four KLs) (+ L1(G(x), y) + L1(F(y), x) when paired)."""
    m = {}
    if eps4 is None:
        Gx = autoencoder_forward(x, Q, "G.")
        FGx = autoencoder_forward(Gx, Q, "F.")
        Fy = autoencoder_forward(y, Q, "F.")
        GFy = autoencoder_forward(Fy, Q, "G.")
        total = lambda_cycle * cycle_loss(x, y, FGx, GFy)
    else:
        Gx, mu_x, lv_x = vae_forward(x, Q, "G.", eps4[0])
        FGx, mu_FGx, lv_FGx = vae_forward(Gx, Q, "F.", eps4[1])
        Fy, mu_y, lv_y = vae_forward(y, Q, "F.", eps4[2])
        GFy, mu_GFy, lv_GFy = vae_forward(Fy, Q, "G.", eps4[3])
        m["loss_kl"] = kl_loss(mu_x, lv_x) + kl_loss(mu_FGx, lv_FGx) + kl_loss(mu_y, lv_y) + kl_loss(mu_GFy, lv_GFy)
        total = lambda_cycle * cycle_loss(x, y, FGx, GFy) + lambda_kl * m["loss_kl"]
    m["loss_cycle"] = cycle_loss(x, y, FGx, GFy)
    if paired:
        m["loss_trans"] = l1(Gx, y) + l1(Fy, x)
        total = total + m["loss_trans"]
    return total, m, Gx, Fy


def cycle_nogan_step(P, state, x, y, eps4, lr, paired=False, lambda_cycle=10.0, lambda_kl=1e-5):
    """CycleAE.training_step (Networks.py:1397-1439; eps4 None) / CycleVAE.training_step (:1525-1572)."""
    names = trainable_names(P)
    Q = _leaf_params(P, names)
    total, m, Gx, Fy = _cycle_nogan_losses(Q, x, y, eps4, paired, lambda_cycle, lambda_kl)
    grads = _grads(total, Q, names)
    adam_update(P, grads, state, names, lr)
    out = {k: v.item() for k, v in m.items()}
    out["total_loss"] = out["G_loss"] = total.item()
    return out, {"Gx": Gx.detach(), "Fy": Fy.detach()}, grads


def cycle_nogan_validation(P, x, y, eps4, paired=False, lambda_cycle=10.0, lambda_kl=1e-5):
    """CycleAE.validation_step (Networks.py:1441-1480) / CycleVAE.validation_step (:1574-1616)."""
    with torch.no_grad():
        total, m, Gx, Fy = _cycle_nogan_losses(P, x, y, eps4, paired, lambda_cycle, lambda_kl)
    out = {k: v.item() for k, v in m.items()}
    out["total_loss"] = out["G_loss"] = total.item()
    return out, {"Gx": Gx, "Fy": Fy}


# ----------------------------------------------------------------------------- DoubleAutoencoder / DoubleVAE
def _double_forward(Q, x, y, eps):
    """DoubleAutoencoder.forward (Networks.py:446-462) when eps is None, else DoubleVariationalAutoencoder.forward
    (:663-690; eps[0] for block A on enc(x), eps[1] for block B on enc(y))."""
    ex, ey = encoder(x, Q, "encoder."), encoder(y, Q, "encoder.")
    if eps is None:
        return decoder(ex, Q, "decoder_A."), decoder(ey, Q, "decoder_B."), None
    zx, mu_x, lv_x = variational_encoder_block(ex, Q, "vae_encoder_block_A.", eps[0])
    zy, mu_y, lv_y = variational_encoder_block(ey, Q, "vae_encoder_block_B.", eps[1])
    Gx = decoder(s_conv(zx, Q, "vae_decoder_block_A.conv."), Q, "decoder_A.")
    Gy = decoder(s_conv(zy, Q, "vae_decoder_block_B.conv."), Q, "decoder_B.")
    return Gx, Gy, (mu_x, lv_x, mu_y, lv_y)


def _double_losses(Q, x, y, eps, lambda_kl):
    Gx, Gy, lat = _double_forward(Q, x, y, eps)
    m = {"loss_recon_A": l1(Gx, x), "loss_recon_B": l1(Gy, y)}
    total = m["loss_recon_A"] + m["loss_recon_B"]
    if lat is not None:
        m["loss_kl_A"], m["loss_kl_B"] = kl_loss(lat[0], lat[1]), kl_loss(lat[2], lat[3])
        m["loss_kl"] = m["loss_kl_A"] + m["loss_kl_B"]
        total = total + lambda_kl * m["loss_kl"]
    return total, m


def double_step(P, state, x, y, eps, lr, lambda_kl=1e-5):
    """DoubleAutoencoder.training_step (Networks.py:502-541; eps None) / DoubleVariationalAutoencoder.training_step (:764-808)."""
    names = trainable_names(P)
    Q = _leaf_params(P, names)
    total, m = _double_losses(Q, x, y, eps, lambda_kl)
    grads = _grads(total, Q, names)
    adam_update(P, grads, state, names, lr)
    out = {k: v.item() for k, v in m.items()}
    out["G_loss"] = out["total_loss"] = total.item()
    return out, grads


def double_validation(P, x, y, eps, lambda_kl=1e-5):
    """validation_step (:543-578 / :810-852): the losses of the two reconstructions, and as images the TRANSLATIONS
    A->B = decoder_B(enc(x)) and B->A = decoder_A(enc(y)); the VAE draws eps[2], eps[3] for them."""
    with torch.no_grad():
        total, m = _double_losses(P, x, y, eps, lambda_kl)
        ex, ey = encoder(x, P, "encoder."), encoder(y, P, "encoder.")
        if eps is None:
            Gx, Fy = decoder(ex, P, "decoder_B."), decoder(ey, P, "decoder_A.")
        else:
            zx, _, _ = variational_encoder_block(ex, P, "vae_encoder_block_B.", eps[2])
            Gx = decoder(s_conv(zx, P, "vae_decoder_block_B.conv."), P, "decoder_B.")
            zy, _, _ = variational_encoder_block(ey, P, "vae_encoder_block_A.", eps[3])
            Fy = decoder(s_conv(zy, P, "vae_decoder_block_A.conv."), P, "decoder_A.")
    out = {k: v.item() for k, v in m.items()}
    out["G_loss"] = out["total_loss"] = total.item()
    return out, {"Gx": Gx, "Fy": Fy}


# ----------------------------------------------------------------------------- AEGAN / VAEGAN
def _single_gan_losses(Q, x, y, eps, run_d, lambda_gan, lambda_identity, lambda_kl, lambda_recon):
    """AEGAN (Networks.py:1023-1028, :1090-1096; eps None, lambda_recon is 1 there) / VAEGAN (:1203-1208, :1266-1275;
    eps[0] for G(x), eps[1] for G(y), KL of the x branch only)."""
    m = {}
    if eps is None:
        Gx, Gy = autoencoder_forward(x, Q, "G."), autoencoder_forward(y, Q, "G.")
    else:
        Gx, mu, logvar = vae_forward(x, Q, "G.", eps[0])
        Gy, _, _ = vae_forward(y, Q, "G.", eps[1])
        m["loss_kl"] = kl_loss(mu, logvar)
    DGx, Dy = run_d(Gx), run_d(y)
    m["loss_trans"], m["loss_identity"] = l1(Gx, y), l1(Gy, y)
    m["gan_g"], m["gan_g_real"], m["gan_g_fake"] = gan_loss_generator(Dy, DGx)
    G_loss = (1.0 if eps is None else lambda_recon) * m["loss_trans"] + lambda_gan * m["gan_g"] + lambda_identity * m["loss_identity"]
    if eps is not None:
        G_loss = G_loss + lambda_kl * m["loss_kl"]
    return G_loss, m, Gx, DGx, Dy


def _single_gan_metrics(variational, training, G_loss, D_loss, d_real, d_fake, m, extra=None):
    """the reference's metric names: AEGAN :1120-1130 / :1166-1178, VAEGAN :1289-1299 / :1334-1344"""
    v = {k: t.item() for k, t in m.items()}
    if not variational:
        out = {"G_loss": G_loss.item(), "D_loss": D_loss.item(), "D_loss_real": d_real.item(), "D_loss_fake": d_fake.item(),
               "loss_trans": v["loss_trans"], "loss_gan_g": v["gan_g"]}
        if training:
            out.update({"loss_identity": v["loss_identity"], **extra})
        else:
            out = {"total_loss": G_loss.item() + D_loss.item(), **out, "loss_gan_g_real": v["gan_g_real"],
                   "loss_gan_g_fake": v["gan_g_fake"], "loss_identity": v["loss_identity"]}
        return out
    if training:
        return {"G_loss": G_loss.item(), "D_loss": D_loss.item(), "loss_gan_disc_real": d_real.item(), "loss_gan_disc_fake": d_fake.item(),
                "loss_trans": v["loss_trans"], "loss_gan_real": v["gan_g_real"], "loss_gan_fake": v["gan_g_fake"],
                "loss_identity": v["loss_identity"], "loss_kl": v["loss_kl"]}
    return {"total_loss": G_loss.item() + D_loss.item(), "G_loss": G_loss.item(), "D_loss": D_loss.item(),
            "loss_trans": v["loss_trans"], "loss_gan_real": v["gan_g_real"], "loss_gan_fake": v["gan_g_fake"],
            "loss_identity": v["loss_identity"], "loss_kl": v["loss_kl"]}


def single_gan_step(P, state, x, y, eps, lr, lambda_gan=1.0, lambda_identity=5.0, lambda_kl=1e-5, lambda_recon=1.0):
    """AEGAN.training_step (Networks.py:1068-1136: D re-run on the detached G(x) after the generator update) /
    VAEGAN.training_step (:1254-1308: D loss on the same pass with DGx detached).  Both give the D gradients of the
    pre-update discriminator on the pre-update G(x), which is what is computed here."""
    g_names, d_names = trainable_names(P, ("G.",)), trainable_names(P, ("D.",))
    Q = _leaf_params(P, g_names + d_names)
    sn = {}

    def run_d(inp):
        out = discriminator(inp, Q, "D.", True, sn)
        Q["D.model.4.weight_u"], Q["D.model.4.weight_v"] = sn["D.model.4.weight_u"], sn["D.model.4.weight_v"]
        return out

    G_loss, m, Gx, _, _ = _single_gan_losses(Q, x, y, eps, run_d, lambda_gan, lambda_identity, lambda_kl, lambda_recon)
    g_grads = _grads(G_loss, Q, g_names)
    adam_update(P, g_grads, state.setdefault("G", {}), g_names, lr)
    DGx_d, Dy_d = run_d(Gx.detach()), run_d(y)
    D_loss, d_real, d_fake = gan_loss_discriminator(Dy_d, DGx_d)
    # VAEGAN detaches the discriminator OUTPUT of the fake branch (`DGx.detach()`, :1277), not its input: the fake term
    # is a constant there and only (1 - D(y))^2 reaches the discriminator's parameters.  Reproduced as written.
    d_grads = _grads(d_real if eps is not None else D_loss, Q, d_names)
    adam_update(P, d_grads, state.setdefault("D", {}), d_names, lr)
    for k, v in sn.items():
        P[k] = v
    extra = {"d_y_mean": Dy_d.mean().item(), "d_gx_mean": DGx_d.mean().item()}
    return _single_gan_metrics(eps is not None, True, G_loss, D_loss, d_real, d_fake, m, extra), g_grads, d_grads


def single_gan_validation(P, x, y, eps, lambda_gan=1.0, lambda_identity=5.0, lambda_kl=1e-5, lambda_recon=1.0):
    """AEGAN.validation_step (Networks.py:1138-1188) / VAEGAN.validation_step (:1310-1348) under model.eval()."""
    with torch.no_grad():
        G_loss, m, Gx, DGx, Dy = _single_gan_losses(P, x, y, eps, lambda inp: discriminator(inp, P, "D.", False),
                                                     lambda_gan, lambda_identity, lambda_kl, lambda_recon)
        D_loss, d_real, d_fake = gan_loss_discriminator(Dy, DGx)
    return _single_gan_metrics(eps is not None, False, G_loss, D_loss, d_real, d_fake, m), {"Gx": Gx}

