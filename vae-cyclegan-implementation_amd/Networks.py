"""MI355X-native networks behind the reference's `Networks.py` module surface.

Class names, constructor signatures, attribute names, `forward` return order, metric keys
and state_dict keys/shapes follow the reference (file:line cited per class) so a
reference checkpoint loads and `train.py` drives these classes unchanged.  Underneath,
every block is a fused HIP launch sequence (`ops.conv_block`): implicit-GEMM / Winograd conv on the
16-bit matrix pipe with split fp32 operands (two fp16 pieces, fp32-level rounding) and a bias/activation epilogue,
two-stage InstanceNorm statistics, and a normalise(+activation)(+residual)(+PixelShuffle) store.  torch modules (`nn.Conv2d`,
`spectral_norm`) are used only as parameter containers — their forwards are never run.

Tensors crossing module boundaries keep the logical (N, C, H, W) shape and are stored
NHWC (pitch 4 for 3-channel images); NCHW-contiguous inputs are converted on entry.
"""
import math

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops
from .Losses import (CycleConsistencyLoss, GANLossDiscriminator, GANLossGenerator, IdentityLoss,
                     KLDivergenceLoss, TranslationLoss)
from .optim import FusedAdam

_ACTS = {"ReLU": ops.ACT_RELU, "LeakyReLU": ops.ACT_LEAKY, "Identity": ops.ACT_NONE, "Tanh": ops.ACT_TANH, "Sigmoid": ops.ACT_SIGMOID}


def _kaiming_relu_init(module):
    """Kaiming-normal fan_out (gain sqrt 2), zero bias: reference Networks.py:168-178, 1893-1903."""
    if isinstance(module, nn.Conv2d):
        nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
        if module.bias is not None:
            nn.init.zeros_(module.bias)


# --------------------------------------------------------------------------- atoms
class CaSb(nn.Module):
    """reflect conv -> [InstanceNorm] -> activation  (reference Networks.py:57-81)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=3, activation="ReLU", use_norm=True):
        super().__init__()
        if activation not in ("ReLU", "LeakyReLU", "Tanh", "Sigmoid", "Identity"):
            raise NotImplementedError("Activation not implemented")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              padding_mode="reflect")
        self.norm = nn.InstanceNorm2d(out_channels)
        self.activation_name = activation
        self.use_norm = use_norm
        act = _ACTS.get(activation)
        if use_norm:
            self._spec = ops.ConvSpec(in_channels, out_channels, kernel_size, stride, padding, True, 1,
                                      ops.ACT_NONE, True, act if act is not None else 0)
        else:
            self._spec = ops.ConvSpec(in_channels, out_channels, kernel_size, stride, padding, True, 1,
                                      act if act is not None else 0, False)

    def forward(self, x, defer_out=False):
        return ops.conv_block(x, self.conv.weight, self.conv.bias, self._spec, defer=defer_out)


class D(nn.Module):
    """PixelUnshuffle(2) -> reflect conv3x3 -> ReLU -> InstanceNorm  (reference Networks.py:83-96).
    The unshuffle is folded into the conv's gather addresses (K ordered (kh,kw,i,j,c))."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.PixelUnshuffle = nn.PixelUnshuffle(downscale_factor=2)
        self.conv = nn.Conv2d(in_channels * 4, out_channels, kernel_size=3, stride=1, padding=1, padding_mode="reflect")
        self.norm = nn.InstanceNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=False)
        self._spec = ops.ConvSpec(in_channels * 4, out_channels, 3, 1, 1, True, 2, ops.ACT_RELU, True)

    def forward(self, x, defer_out=False):
        """`defer_out` (internal, Encoder.forward): hand the raw conv output and its statistics to the next block's gather
        instead of writing the normalised tensor (ops.conv_block, "deferred InstanceNorm")."""
        return ops.conv_block(x, self.conv.weight, self.conv.bias, self._spec, defer=defer_out)


class R(nn.Module):
    """x + IN(conv(IN(ReLU(conv(x)))))  (reference Networks.py:98-116)."""

    def __init__(self, out_channels):
        super().__init__()
        c = out_channels
        self.conv1 = nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1, padding_mode="reflect")
        self.norm1 = nn.InstanceNorm2d(c)
        self.activation1 = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(c, c, kernel_size=3, stride=1, padding=1, padding_mode="reflect")
        self.norm2 = nn.InstanceNorm2d(c)
        self._spec1 = ops.ConvSpec(c, c, 3, 1, 1, True, 1, ops.ACT_RELU, True)
        self._spec2 = ops.ConvSpec(c, c, 3, 1, 1, True, 1, ops.ACT_NONE, True)

    def forward(self, x):
        x = ops.to_nhwc(x)
        # conv1's InstanceNorm is applied inside conv2's input gather where conv2's geometry has one (the 1024-channel Winograd
        # layers of the training sizes): h is then never written (the fused Conv + IN + act hand-off)
        n, _, hh, ww = x.shape
        defer = ops.consumer_takes_deferred(self._spec2, n, hh, ww, torch.is_grad_enabled())
        h = ops.conv_block(x, self.conv1.weight, self.conv1.bias, self._spec1, defer=defer)
        return ops.conv_block(h, self.conv2.weight, self.conv2.bias, self._spec2, residual=x)


class U(nn.Module):
    """PixelShuffle(2) -> reflect conv3x3 -> ReLU -> InstanceNorm  (reference Networks.py:118-131).

    `pre_shuffled`: the producer already stored its output through the shuffle.
    `shuffle_out`: store this block's output through the NEXT block's PixelShuffle."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.PixelShuffle = nn.PixelShuffle(upscale_factor=2)
        self.conv = nn.Conv2d(in_channels // 4, out_channels, kernel_size=3, stride=1, padding=1, padding_mode="reflect")
        self.norm = nn.InstanceNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=False)
        self._spec = ops.ConvSpec(in_channels // 4, out_channels, 3, 1, 1, True, 1, ops.ACT_RELU, True)
        self._spec_shuf = None
        if out_channels % 16 == 0:
            self._spec_shuf = ops.ConvSpec(in_channels // 4, out_channels, 3, 1, 1, True, 1, ops.ACT_RELU, True,
                                           ops.ACT_NONE, True)

    def forward(self, x, pre_shuffled=False, shuffle_out=False):
        if not pre_shuffled:
            x = ops.pixel_shuffle(x)
        spec = self._spec_shuf if shuffle_out else self._spec
        if spec is None:
            raise RuntimeError("shuffle_out needs out_channels % 16 == 0")
        return ops.conv_block(x, self.conv.weight, self.conv.bias, spec)


class S(nn.Module):
    """bare reflect conv3x3  (reference Networks.py:133-140)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, padding_mode="reflect")
        self._spec = ops.ConvSpec(in_channels, out_channels, 3, 1, 1, True, 1)

    def forward(self, x):
        return ops.conv_block(x, self.conv.weight, self.conv.bias, self._spec)


class L(S):
    """bare reflect conv3x3  (reference Networks.py:142-149)."""


# --------------------------------------------------------------------------- molecules
class Encoder(nn.Module):
    """3 -> 1024 channels at 1/16 resolution  (reference Networks.py:154-181)."""

    def __init__(self):
        super().__init__()
        self.model = nn.Sequential(CaSb(3, 64, kernel_size=7, stride=1), D(64, 128), D(128, 256), D(256, 512),
                                   D(512, 1024), R(1024))
        self.apply(self._init_weights)

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def forward(self, x):
        out = ops.to_nhwc(x)
        layers = list(self.model)
        for i, layer in enumerate(layers):
            nxt = layers[i + 1] if i + 1 < len(layers) else None
            if isinstance(layer, (CaSb, D)) and isinstance(nxt, D):
                # the next D block normalises this block's raw output in its own gather where its geometry can (the Winograd
                # input transform: D2..D4 at the training sizes) — the normalised tensor is then never written
                n, _, hh, ww = out.shape
                ho, wo = layer._spec.out_hw(hh, ww)
                defer = layer._spec.norm and ops.consumer_takes_deferred(nxt._spec, n, ho, wo, torch.is_grad_enabled())
                out = layer(out, defer_out=defer)
            else:
                out = layer(out)
        return out


class Decoder(nn.Module):
    """mirror of Encoder  (reference Networks.py:183-211).  U->U hand-offs are stored pre-shuffled."""

    def __init__(self):
        super().__init__()
        self.model = nn.Sequential(R(1024), U(1024, 512), U(512, 256), U(256, 128), U(128, 64),
                                   CaSb(64, 3, kernel_size=7, stride=1, activation="Identity", use_norm=False))
        self.apply(self._init_weights)

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def forward(self, x):
        layers = list(self.model)
        out = ops.to_nhwc(x)
        pre = False
        for i, layer in enumerate(layers):
            if isinstance(layer, U):
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                fuse = isinstance(nxt, U) and layer._spec_shuf is not None
                out = layer(out, pre_shuffled=pre, shuffle_out=fuse)
                pre = fuse
            else:
                out = layer(out)
                pre = False
        return out


class VariationalEncoderBlock(nn.Module):
    """mu / logvar convs, clamp, z = mu + eps * exp(0.5 logvar)  (reference Networks.py:214-227)."""

    def __init__(self, in_channels, latent_dim=64):
        super().__init__()
        self.muConv = L(in_channels, latent_dim)
        self.logvarConv = nn.Sequential(S(in_channels, latent_dim), S(latent_dim, latent_dim))
        self.latent_dim = latent_dim
        # mu and the first logvar convolution read the same map: one convolution with 2 x latent outputs (ops.FusedConvPair;
        # the parameters, state_dict names and optimizer entries stay the reference's)
        self._pair = None
        if latent_dim % 4 == 0:
            object.__setattr__(self, "_pair", ops.FusedConvPair(self.muConv.conv, self.logvarConv[0].conv,
                                                                ops.ConvSpec(in_channels, 2 * latent_dim, 3, 1, 1, True, 1)))

    def forward(self, x):
        x = ops.to_nhwc(x)
        if ops.FUSE_MU_LOGVAR and self._pair is not None and x.is_cuda:
            mu, lv0 = ops.conv_pair(x, self._pair)
            logvar = self.logvarConv[1](lv0)
            eps = ops.next_eps(mu.shape, mu.device)
            z, logvar = ops.reparameterize(mu, logvar, eps)
            return z, mu, logvar
        mu = self.muConv(x)
        logvar = self.logvarConv(x)
        eps = ops.next_eps(mu.shape, mu.device)
        z, logvar = ops.reparameterize(mu, logvar, eps)
        return z, mu, logvar


class VariationalDecoderBlock(nn.Module):
    """latent -> 1024 channels  (reference Networks.py:230-237)."""

    def __init__(self, latent_dim=64, out_channels=1024):
        super().__init__()
        self.conv = S(latent_dim, out_channels)

    def forward(self, z):
        return self.conv(z)


class Discriminator(nn.Module):
    """4x (conv4x4 s2 [+IN] + LeakyReLU 0.2) + spectral-normed 16x16 conv -> one scalar per image
    (reference Networks.py:240-269)."""

    def __init__(self):
        super().__init__()
        self.model = nn.Sequential(
            CaSb(3, 64, kernel_size=4, stride=2, padding=1, activation="LeakyReLU", use_norm=False),
            CaSb(64, 128, kernel_size=4, stride=2, padding=1, activation="LeakyReLU"),
            CaSb(128, 256, kernel_size=4, stride=2, padding=1, activation="LeakyReLU"),
            CaSb(256, 512, kernel_size=4, stride=2, padding=1, activation="LeakyReLU"),
            spectral_norm(nn.Conv2d(512, 1, kernel_size=16, stride=1, padding=0)),
        )
        self.apply(self._init_weights)

    def _init_weights(self, module):
        if isinstance(module, nn.Conv2d):
            nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="leaky_relu", a=0.2)
            if module.bias is not None:
                nn.init.zeros_(module.bias)

    def forward(self, x):
        x = ops.to_nhwc(x)
        layers = list(self.model)
        for blk in layers[:-1]:
            x = blk(x)
        head = layers[-1]
        return ops.fullmap_sn_conv(x, head.weight_orig, head.bias, head.weight_u, head.weight_v, self.training)


# --------------------------------------------------------------------------- composites
def _metrics_to_host(named, reducer=None):
    """One device->host copy for all the step's scalars (the reference does one .item() each).
    Under data parallelism the scalars are first averaged over ranks (= the global-batch means)."""
    keys = list(named.keys())
    vec = torch.stack([named[k].detach().reshape(()) for k in keys])
    if reducer is not None:
        vec = reducer.average_metrics(vec)
    return dict(zip(keys, vec.tolist()))


def _backward_and_step(loss, optimizer, reducer):
    """zero_grad -> backward -> [data-parallel gradient exchange, its buckets launched from inside the backward] -> step."""
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin(optimizer)
    ops.backward_overlapped(loss)
    if reducer is not None:
        reducer.start(optimizer)
        reducer.finish(optimizer)
    optimizer.step()


class _OptimizerStatesMixin:
    _opt_names = ("optimizer",)

    def save_optimizer_states(self):
        out = {}
        for name in self._opt_names:
            opt = getattr(self, name)
            if opt is None:
                raise ValueError("Optimizer has not been configured yet." if len(self._opt_names) == 1
                                 else "Optimizers have not been configured yet.")
            out[name] = opt.state_dict()
        return out

    def load_optimizer_states(self, states):
        for name in self._opt_names:
            if getattr(self, name) is None:
                raise ValueError("Optimizer has not been configured yet." if len(self._opt_names) == 1
                                 else "Optimizers have not been configured yet.")
        for name in self._opt_names:
            if name not in states:
                raise KeyError(f"{name} state not found in states")
            getattr(self, name).load_state_dict(states[name])


class Autoencoder(_OptimizerStatesMixin, nn.Module):
    """Encoder -> Decoder, L1 loss, one Adam  (reference Networks.py:276-413)."""

    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder()
        self.optimizer = None
        self.grad_reducer = None
        self.loss_fn = None
        self.apply(self._init_weights)

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def forward(self, x):
        return self.decoder(self.encoder(x))

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999), decoder_only=False):
        params = self.decoder.parameters() if decoder_only else self.parameters()
        self.optimizer = FusedAdam(params, lr=lr, betas=betas)
        return self.optimizer

    def configure_loss(self, **kwargs):
        self.loss_fn = TranslationLoss()

    def training_step(self, batch):
        if self.loss_fn is None:
            raise ValueError("Loss function has not been configured yet.")
        if self.optimizer is None:
            raise ValueError("Optimizer has not been configured yet.")
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        output = self(x)
        loss_trans = self.loss_fn(output, y)
        value = float(loss_trans.detach())             # the reference's isnan/isinf guard syncs here too (:357)
        bad = math.isnan(value) or math.isinf(value)
        red = self.grad_reducer
        if red is not None:
            # data parallel: the decision is collective — a rank that skipped alone would never join the gradient
            # exchange the others enter, and the replicas would diverge
            bad = red.any_rank(bad)
        if bad:
            print("NaN or Inf detected in loss during training step; skipping the update.")
            self.optimizer.zero_grad()
            return {"nan_detected": True, "G_loss": float("nan"), "loss_trans": float("nan"), "total_loss": float("nan")}
        _backward_and_step(loss_trans, self.optimizer, red)
        if red is not None:
            value = _metrics_to_host({"loss_trans": loss_trans}, red)["loss_trans"]      # the global-batch mean, as every other model logs
        return {"G_loss": value, "loss_trans": value, "total_loss": value}

    def validation_step(self, batch):
        if self.loss_fn is None:
            raise ValueError("Loss function has not been configured yet.")
        with torch.no_grad():
            x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
            output = self(x)
            value = float(self.loss_fn(output, y))
            return {"G_loss": value, "total_loss": value, "loss_trans": value, "Gx": output}


class VariationalAutoencoder(_OptimizerStatesMixin, nn.Module):
    """Encoder -> VAE bottleneck -> Decoder, L1 + lambda_kl * KL  (reference Networks.py:855-988)."""

    def __init__(self, latent_dim=64):
        super().__init__()
        self.encoder = Encoder()
        self.variational_encoder_block = VariationalEncoderBlock(in_channels=1024, latent_dim=latent_dim)
        self.variational_decoder_block = VariationalDecoderBlock(latent_dim=latent_dim, out_channels=1024)
        self.decoder = Decoder()
        self.optimizer = None
        self.grad_reducer = None
        self.loss_trans_fn = None
        self.loss_kl_fn = None
        self.lambda_kl = 0
        self.apply(self._init_weights)

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def forward(self, x):
        encoded = self.encoder(x)
        z, mu, logvar = self.variational_encoder_block(encoded)
        Gx = self.decoder(self.variational_decoder_block(z))
        return Gx, mu, logvar

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999)):
        self.optimizer = FusedAdam(self.parameters(), lr=lr, betas=betas)
        return self.optimizer

    def configure_loss(self, **kwargs):
        self.loss_trans_fn = TranslationLoss()
        self.loss_kl_fn = KLDivergenceLoss()
        self.lambda_kl = kwargs.get("lambda_kl", 1e-5)

    def _check_configured(self):
        if self.optimizer is None:
            raise ValueError("Optimizer has not been configured yet.")
        if self.loss_trans_fn is None:
            raise ValueError("Translation loss function has not been configured yet.")
        if self.loss_kl_fn is None:
            raise ValueError("KL divergence loss function has not been configured yet.")

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        output, mu, logvar = self(x)
        loss_trans = self.loss_trans_fn(output, y)
        loss_kl = self.loss_kl_fn(mu, logvar)
        G_loss = ops.weighted_sum([loss_trans, loss_kl], [1.0, self.lambda_kl])
        return output, G_loss, loss_trans, loss_kl

    def training_step(self, batch):
        self._check_configured()
        _, G_loss, loss_trans, loss_kl = self._losses(batch)
        _backward_and_step(G_loss, self.optimizer, self.grad_reducer)
        return _metrics_to_host({"G_loss": G_loss, "loss_trans": loss_trans, "loss_kl": loss_kl}, self.grad_reducer)

    def validation_step(self, batch):
        self._check_configured()
        with torch.no_grad():
            output, G_loss, loss_trans, loss_kl = self._losses(batch)
            m = _metrics_to_host({"G_loss": G_loss, "loss_trans": loss_trans, "loss_kl": loss_kl})
            m["Gx"] = output
            return m


def _vae_pair(vae_a, a, ticket_a, vae_b, b, ticket_b, fork):
    """vae_a(a) on the caller's stream and vae_b(b) on `fork`'s second stream, issued alternately in half-generator pieces
    (encoder | bottleneck + decoder): the second stream has work after a quarter of the host time a whole generator takes to
    issue, and — autograd replays in reverse issue order — the backward alternates between the two chains at the same
    granularity instead of one whole generator at a time (CycleVAEGAN step 32.54 -> 31.89 ms; block by block, through generator
    coroutines, measured no better: 32.4 vs 32.3).  VCG_DIR_INTERLEAVE=0: whole generators."""
    def tail(vae, enc, ticket):                  # VariationalAutoencoder.forward after its encoder
        with ops.use_ticket(ticket):
            z, mu, lv = vae.variational_encoder_block(enc)
        return vae.decoder(vae.variational_decoder_block(z)), mu, lv

    if not ops.DIR_INTERLEAVE:
        with ops.use_ticket(ticket_a):
            ra = vae_a(a)
        with fork.second(), ops.use_ticket(ticket_b):
            rb = vae_b(b)
        return ra, rb
    ea = vae_a.encoder(a)
    with fork.second():
        eb = vae_b.encoder(b)
    ra = tail(vae_a, ea, ticket_a)
    with fork.second():
        rb = tail(vae_b, eb, ticket_b)
    return ra, rb


def _ae_pair(ae_a, a, ae_b, b, fork):
    """The same for two plain autoencoders."""
    if not ops.DIR_INTERLEAVE:
        ra = ae_a(a)
        with fork.second():
            rb = ae_b(b)
        return ra, rb
    ea = ae_a.encoder(a)
    with fork.second():
        eb = ae_b.encoder(b)
    ra = ae_a.decoder(ea)
    with fork.second():
        rb = ae_b.decoder(eb)
    return ra, rb


class CycleVAEGAN(nn.Module):
    """Two VAEs (G: X->Y, F: Y->X) + two discriminators; cycle + LSGAN + KL (+identity if paired);
    alternating G then D update  (reference Networks.py:1872-2150).

    Differences from the reference are work that cannot change any result:
      * unpaired mode does not compute G(y), F(x) inside training_step (their outputs feed only
        the identity loss, reference :2016-2018); their eps draws are still consumed;
      * the discriminators run ONCE per step: the G-phase backward takes only the data gradient
        and the D-phase backward takes the weight gradients from the same saved activations
        (the reference recomputes four D forwards at :2032-2035 with unchanged D weights, and
        discards the D weight gradients of its G-phase backward at :2025).
    """

    def __init__(self, latent_dim=64, paired=True):
        super().__init__()
        self.F = VariationalAutoencoder(latent_dim)
        self.G = VariationalAutoencoder(latent_dim)
        self.DX = Discriminator()
        self.DY = Discriminator()
        self.paired = paired
        self.apply(self._init_weights)
        self.debug_mode = False
        self.debug_info = {}
        self.optimizer_G = None
        self.optimizer_D = None
        self.loss_cycle = None
        self.loss_gan_gen = None
        self.loss_gan_disc = None
        self.loss_identity = None
        self.loss_kl = None
        # data-parallel hook: set by parallel.attach(); called as reducer(phase, optimizer)
        self.grad_reducer = None

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def enable_debug_mode(self, enabled=True):
        self.debug_mode = enabled

    def forward(self, x, y):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        Gx, mu_x, logvar_x = self.G(x)
        Gy, _, _ = self.G(y)
        FGx, mu_FGx, logvar_FGx = self.F(Gx)
        Fy, mu_y, logvar_y = self.F(y)
        Fx, _, _ = self.F(x)
        GFy, mu_GFy, logvar_GFy = self.G(Fy)
        DYGx = self.DY(Gx)
        DXFy = self.DX(Fy)
        DXx = self.DX(x)
        DYy = self.DY(y)
        return (Gx, FGx, Fy, GFy, mu_x, logvar_x, mu_FGx, logvar_FGx, mu_y, logvar_y, mu_GFy, logvar_GFy,
                DYGx, DXFy, DXx, DYy, Gy, Fx)

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999)):
        self.optimizer_G = FusedAdam(list(self.F.parameters()) + list(self.G.parameters()), lr=lr, betas=betas)
        self.optimizer_D = FusedAdam(list(self.DX.parameters()) + list(self.DY.parameters()), lr=lr, betas=betas)
        return self.optimizer_G, self.optimizer_D

    def save_optimizer_states(self):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        return {"optimizer_G": self.optimizer_G.state_dict(), "optimizer_D": self.optimizer_D.state_dict()}

    def load_optimizer_states(self, states):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        for name in ("optimizer_G", "optimizer_D"):
            if name not in states:
                raise KeyError(f"{name} state not found in states")
        self.optimizer_G.load_state_dict(states["optimizer_G"])
        self.optimizer_D.load_state_dict(states["optimizer_D"])

    def configure_loss(self, **kwargs):
        self.loss_cycle = CycleConsistencyLoss()
        self.loss_gan_gen = GANLossGenerator()
        self.loss_gan_disc = GANLossDiscriminator()
        if self.paired:
            self.loss_identity = IdentityLoss()
        self.loss_kl = KLDivergenceLoss()
        self.lambda_gan = kwargs.get("lambda_gan", 1.0)
        self.lambda_identity = kwargs.get("lambda_identity", 5.0)
        self.lambda_cycle = kwargs.get("lambda_cycle", 10.0)
        self.lambda_kl = kwargs.get("lambda_kl", 1e-5)

    def _check_configured(self, need_opt=True):
        if self.loss_cycle is None or self.loss_gan_gen is None or self.loss_gan_disc is None or self.loss_kl is None:
            raise ValueError("Loss functions have not been configured yet.")
        if self.paired and self.loss_identity is None:
            raise ValueError("Identity loss not configured for paired mode.")
        if need_opt and (self.optimizer_G is None or self.optimizer_D is None):
            raise ValueError("Optimizers have not been configured yet.")

    def _skip_vae(self, ref_shape_src, vae):
        """Advance the eps stream past a VAE forward that is not computed."""
        n, _, h, w = ref_shape_src.shape
        lat = vae.variational_encoder_block.latent_dim
        ops.next_eps((n, lat, h // 16, w // 16), ref_shape_src.device, skip=True)

    def _forward_two_streams(self, x, y):
        """The unpaired forward with the two translation directions on two streams (ops.DirectionFork): x -> G -> F -> DY on the
        caller's stream, y -> F -> G -> DX on the second one, issued interleaved so that both have work from the start.  Same
        results bit for bit: the eps draws are reserved in the reference's call order (G(x), [G(y)], F(G(x)), F(y), [F(x)],
        G(F(y))), each discriminator still sees its two inputs in the reference's order (spectral norm's power iteration
        advances per call: DY: G(x) then y; DX: F(y) then x)."""
        n, _, h, w = x.shape
        shp = (n, self.G.variational_encoder_block.latent_dim, h // 16, w // 16)
        tk = ops.eps_tickets([(shp, False), (shp, True), (shp, False), (shp, False), (shp, True), (shp, False)], x.device)
        ops.premeasure(x)
        ops.premeasure(y)
        fork = ops.DirectionFork(x.device)
        (Gx, mu_x, lv_x), (Fy, mu_y, lv_y) = _vae_pair(self.G, x, tk[0], self.F, y, tk[3], fork)
        (FGx, mu_FGx, lv_FGx), (GFy, mu_GFy, lv_GFy) = _vae_pair(self.F, Gx, tk[2], self.G, Fy, tk[5], fork)
        DYGx = self.DY(Gx)                       # alternately again (32.64 -> 32.42 ms over three A/B pairs)
        with fork.second():
            DXFy = self.DX(Fy)
        DYy = self.DY(y)
        with fork.second():
            DXx = self.DX(x)
        fork.join()
        return (Gx, mu_x, lv_x, FGx, mu_FGx, lv_FGx, Fy, mu_y, lv_y, GFy, mu_GFy, lv_GFy, DYGx, DXFy, DXx, DYy)

    def _generator_losses(self, x, y, two_streams=False):
        """Forward of both generators and everything G_loss needs (reference :1997-2018).  `two_streams`: see
        `_forward_two_streams` (training_step / validation_step ask for it when the weight gradients overlap too)."""
        Gy = Fx = None
        if two_streams and not self.paired and x.is_cuda:
            (Gx, mu_x, lv_x, FGx, mu_FGx, lv_FGx, Fy, mu_y, lv_y, GFy, mu_GFy, lv_GFy, DYGx, DXFy, DXx,
             DYy) = self._forward_two_streams(x, y)
        else:
            Gx, mu_x, lv_x = self.G(x)
            if self.paired:
                Gy, _, _ = self.G(y)
            else:
                self._skip_vae(y, self.G)
            FGx, mu_FGx, lv_FGx = self.F(Gx)
            Fy, mu_y, lv_y = self.F(y)
            if self.paired:
                Fx, _, _ = self.F(x)
            else:
                self._skip_vae(x, self.F)
            GFy, mu_GFy, lv_GFy = self.G(Fy)
            DYGx = self.DY(Gx)
            DXFy = self.DX(Fy)
            DXx = self.DX(x)
            DYy = self.DY(y)

        t = {}
        t["loss_cycle"] = self.loss_cycle(x, y, FGx, GFy)
        t["loss_gan_g_x_fake"], t["d_x_fake_mean"] = ops.mse_const(DXFy, 1.0)
        t["loss_gan_g_y_fake"], t["d_y_fake_mean"] = ops.mse_const(DYGx, 1.0)
        t["loss_gan_g_x_real"], t["d_x_real_mean"] = ops.mse_const(DXx, 0.0)
        t["loss_gan_g_y_real"], t["d_y_real_mean"] = ops.mse_const(DYy, 0.0)
        t["loss_gan_g"] = ops.weighted_sum([t["loss_gan_g_x_fake"], t["loss_gan_g_y_fake"]], [1.0, 1.0])
        t["loss_kl"] = ops.weighted_sum([self.loss_kl(mu_x, lv_x), self.loss_kl(mu_FGx, lv_FGx),
                                         self.loss_kl(mu_y, lv_y), self.loss_kl(mu_GFy, lv_GFy)], [1.0] * 4)
        terms = [t["loss_cycle"], t["loss_gan_g"], t["loss_kl"]]
        weights = [self.lambda_cycle, self.lambda_gan, self.lambda_kl]
        if self.paired:
            t["loss_identity"] = self.loss_identity(x, y, Fx, Gy)
            terms.append(t["loss_identity"])
            weights.append(self.lambda_identity)
        t["G_loss"] = ops.weighted_sum(terms, weights)
        # discriminator objective on the SAME discriminator outputs (reference :2038-2040)
        t["D_loss_x_real"], _ = ops.mse_const(DXx, 1.0)
        t["D_loss_x_fake"], _ = ops.mse_const(DXFy, 0.0)
        t["D_loss_y_real"], _ = ops.mse_const(DYy, 1.0)
        t["D_loss_y_fake"], _ = ops.mse_const(DYGx, 0.0)
        t["D_loss"] = ops.weighted_sum([t["D_loss_x_real"], t["D_loss_x_fake"], t["D_loss_y_real"], t["D_loss_y_fake"]],
                                       [1.0] * 4)
        return t, Gx, Fy

    _METRIC_KEYS = ("G_loss", "D_loss", "D_loss_x_real", "D_loss_x_fake", "D_loss_y_real", "D_loss_y_fake",
                    "loss_cycle", "loss_gan_g", "loss_gan_g_x_real", "loss_gan_g_x_fake", "loss_gan_g_y_real",
                    "loss_gan_g_y_fake", "loss_kl")
    _MEAN_KEYS = ("d_x_real_mean", "d_x_fake_mean", "d_y_real_mean", "d_y_fake_mean")

    def _metrics(self, t, with_means):
        keys = list(self._METRIC_KEYS) + (list(self._MEAN_KEYS) if with_means else [])
        if self.paired:
            keys.append("loss_identity")
        host = _metrics_to_host({k: t[k] for k in keys}, self.grad_reducer if with_means else None)
        out = {"total_loss": host["G_loss"] + host["D_loss"]}
        out.update(host)
        return out

    def training_step(self, batch):
        self._check_configured()
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        g_params = self.optimizer_G.params
        d_params = self.optimizer_D.params

        red = self.grad_reducer
        self.optimizer_G.zero_grad()
        t, _, _ = self._generator_losses(x, y, two_streams=ops.two_directions())
        if red is not None:
            red.begin(self.optimizer_G)          # F+G buckets are all-reduced from inside the backward as they complete
        # generator gradients reach F and G only (the discriminators contribute their data gradient)
        with ops.no_wgrad(d_params):
            ops.backward_overlapped(t["G_loss"], inputs=g_params, retain_graph=True)
        if red is not None:
            red.start(self.optimizer_G)          # whatever is left; it runs under the D backward below
        # discriminator gradients from the same activations reach DX and DY only — what detaching
        # G(x), F(y) achieves in the reference (:2028-2029).  Neither this backward nor D_loss reads a
        # generator parameter, so running it before optimizer_G.step() changes nothing.
        self.optimizer_D.zero_grad()
        if red is not None:
            red.begin(self.optimizer_D)
        with ops.no_dgrad([self.DX.model[0]._spec, self.DY.model[0]._spec]):
            ops.backward_overlapped(t["D_loss"], inputs=d_params)
        if red is not None:
            red.start(self.optimizer_D)
            red.finish(self.optimizer_G)
        self.optimizer_G.step()
        if red is not None:
            red.finish(self.optimizer_D)
        self.optimizer_D.step()
        return self._metrics(t, with_means=True)

    def validation_step(self, batch):
        self._check_configured(need_opt=False)
        with torch.no_grad():
            paired = self.paired
            x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
            t, Gx, Fy = self._generator_losses(x, y)
            m = self._metrics(t, with_means=False)
            m["Gx"] = Gx.detach()
            m["Fy"] = Fy.detach()
            assert paired == self.paired
            return m


class CycleAEGAN(CycleVAEGAN):
    """Two plain autoencoders (G: X->Y, F: Y->X) + two discriminators; cycle + LSGAN (+identity if paired), alternating
    G then D update  (reference Networks.py:1618-1869).  CycleVAEGAN's wiring minus the VAE block: no KL term and no
    eps draws, `forward` returns 10 tensors, and the generator objective carries the WHOLE LSGAN generator loss
    (`loss_gan_g = loss_gan_g_x + loss_gan_g_y`, real + fake terms, :1745-1748) where CycleVAEGAN takes only the fake
    terms — the real terms have no gradient into F and G, but they are part of `G_loss` and of the `loss_gan_g` metric.
    The step itself (one discriminator pass serving both phases, exchange hooks, side stream) is CycleVAEGAN's."""

    def __init__(self, paired=True):
        nn.Module.__init__(self)
        self.F = Autoencoder()
        self.G = Autoencoder()
        self.DX = Discriminator()
        self.DY = Discriminator()
        self.paired = paired
        self.apply(self._init_weights)
        self.debug_mode = False
        self.debug_info = {}
        self.optimizer_G = None
        self.optimizer_D = None
        self.loss_cycle = None
        self.loss_gan_gen = None
        self.loss_gan_disc = None
        self.loss_identity = None
        self.grad_reducer = None

    def forward(self, x, y):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        Gx = self.G(x)
        Gy = self.G(y)
        FGx = self.F(Gx)
        Fy = self.F(y)
        Fx = self.F(x)
        GFy = self.G(Fy)
        return Gx, FGx, Fy, GFy, self.DY(Gx), self.DX(Fy), self.DX(x), self.DY(y), Gy, Fx

    def configure_loss(self, **kwargs):
        self.loss_cycle = CycleConsistencyLoss()
        self.loss_gan_gen = GANLossGenerator()
        self.loss_gan_disc = GANLossDiscriminator()
        if self.paired:
            self.loss_identity = IdentityLoss()
        self.lambda_gan = kwargs.get("lambda_gan", 1.0)
        self.lambda_identity = kwargs.get("lambda_identity", 5.0)
        self.lambda_cycle = kwargs.get("lambda_cycle", 10.0)

    def _check_configured(self, need_opt=True):
        if self.loss_cycle is None or self.loss_gan_gen is None or self.loss_gan_disc is None:
            raise ValueError("Loss functions have not been configured yet.")
        if self.paired and self.loss_identity is None:
            raise ValueError("Identity loss not configured for paired mode.")
        if need_opt and (self.optimizer_G is None or self.optimizer_D is None):
            raise ValueError("Optimizers have not been configured yet.")

    def _generator_losses(self, x, y, two_streams=False):
        """reference :1733-1753; G(y), F(x) feed only the identity loss and are skipped when unpaired.  `two_streams`: the two
        translation directions on two streams (CycleVAEGAN._forward_two_streams; no eps here)."""
        if two_streams and not self.paired and x.is_cuda:
            ops.premeasure(x)
            ops.premeasure(y)
            fork = ops.DirectionFork(x.device)
            Gx, Fy = _ae_pair(self.G, x, self.F, y, fork)
            FGx, GFy = _ae_pair(self.F, Gx, self.G, Fy, fork)
            DYGx = self.DY(Gx)
            with fork.second():
                DXFy = self.DX(Fy)
            DYy = self.DY(y)
            with fork.second():
                DXx = self.DX(x)
            fork.join()
        else:
            Gx = self.G(x)
            FGx = self.F(Gx)
            Fy = self.F(y)
            GFy = self.G(Fy)
            DYGx = self.DY(Gx)
            DXFy = self.DX(Fy)
            DXx = self.DX(x)
            DYy = self.DY(y)
        t = {}
        t["loss_cycle"] = self.loss_cycle(x, y, FGx, GFy)
        t["loss_gan_g_x_fake"], t["d_x_fake_mean"] = ops.mse_const(DXFy, 1.0)
        t["loss_gan_g_y_fake"], t["d_y_fake_mean"] = ops.mse_const(DYGx, 1.0)
        t["loss_gan_g_x_real"], t["d_x_real_mean"] = ops.mse_const(DXx, 0.0)
        t["loss_gan_g_y_real"], t["d_y_real_mean"] = ops.mse_const(DYy, 0.0)
        t["loss_gan_g"] = ops.weighted_sum([t["loss_gan_g_x_real"], t["loss_gan_g_x_fake"], t["loss_gan_g_y_real"],
                                            t["loss_gan_g_y_fake"]], [1.0] * 4)
        terms, weights = [t["loss_cycle"], t["loss_gan_g"]], [self.lambda_cycle, self.lambda_gan]
        if self.paired:
            t["loss_identity"] = self.loss_identity(x, y, self.F(x), self.G(y))
            terms.append(t["loss_identity"])
            weights.append(self.lambda_identity)
        t["G_loss"] = ops.weighted_sum(terms, weights)
        t["D_loss_x_real"], _ = ops.mse_const(DXx, 1.0)
        t["D_loss_x_fake"], _ = ops.mse_const(DXFy, 0.0)
        t["D_loss_y_real"], _ = ops.mse_const(DYy, 1.0)
        t["D_loss_y_fake"], _ = ops.mse_const(DYGx, 0.0)
        t["D_loss"] = ops.weighted_sum([t["D_loss_x_real"], t["D_loss_x_fake"], t["D_loss_y_real"], t["D_loss_y_fake"]],
                                       [1.0] * 4)
        return t, Gx, Fy

    _METRIC_KEYS = tuple(k for k in CycleVAEGAN._METRIC_KEYS if k != "loss_kl")


class _CycleNoGAN(_OptimizerStatesMixin, nn.Module):
    """Two generators G: X->Y, F: Y->X trained on the cycle loss alone (+ KL for VAEs, + translation loss when paired),
    one Adam over both — the shared body of CycleAE and CycleVAE (reference Networks.py:1350-1616)."""

    def _init_common(self, paired):
        self.paired = paired
        self.optimizer = None        # (CycleVAE.__init__ in the reference leaves this attribute unset until configured)
        self.grad_reducer = None
        self.loss_cycle = None
        self.loss_trans = None
        self.loss_kl = None
        self.lambda_cycle = 0
        self.lambda_kl = 0

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999)):
        self.optimizer = FusedAdam(self.parameters(), lr=lr, betas=betas)
        return self.optimizer

    def _check_configured(self, need_opt=True):
        if self.loss_cycle is None or (self._variational and self.loss_kl is None):
            raise ValueError("Loss functions have not been configured yet.")
        if self.paired and self.loss_trans is None:
            raise ValueError("Translation loss not configured for paired mode.")
        if need_opt and self.optimizer is None:
            raise ValueError("Optimizer has not been configured yet.")

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        fw = self._fwd(x, y, ops.two_directions())     # (a bare model(x, y) stays on one stream: its caller may use a plain backward)
        Gx, FGx, Fy, GFy = fw[:4]
        t = {"loss_cycle": self.loss_cycle(x, y, FGx, GFy)}
        terms, weights = [t["loss_cycle"]], [self.lambda_cycle]
        if self._variational:
            t["loss_kl"] = ops.weighted_sum([self.loss_kl(fw[4 + 2 * i], fw[5 + 2 * i]) for i in range(4)], [1.0] * 4)
            terms.append(t["loss_kl"])
            weights.append(self.lambda_kl)
        if self.paired:
            t["loss_trans"] = ops.weighted_sum([self.loss_trans(Gx, y), self.loss_trans(Fy, x)], [1.0, 1.0])
            terms.append(t["loss_trans"])
            weights.append(1.0)
        t["G_loss"] = ops.weighted_sum(terms, weights)
        return t, Gx, Fy

    def training_step(self, batch):
        self._check_configured()
        t, _, _ = self._losses(batch)
        _backward_and_step(t["G_loss"], self.optimizer, self.grad_reducer)
        host = _metrics_to_host(t, self.grad_reducer)
        m = self._ordered(host)
        if self.paired:
            m["loss_trans"] = host["loss_trans"]
        return m

    def _ordered(self, host):
        """the reference's key order (:1421-1433, :1547-1561): total_loss, loss_cycle, [loss_kl], G_loss, [loss_trans]"""
        m = {"total_loss": host["G_loss"], "loss_cycle": host["loss_cycle"]}
        if self._variational:
            m["loss_kl"] = host["loss_kl"]
        m["G_loss"] = host["G_loss"]
        return m

    def validation_step(self, batch):
        self._check_configured(need_opt=False)
        with torch.no_grad():
            t, Gx, Fy = self._losses(batch)
            host = _metrics_to_host(t)
            m = self._ordered(host)
            m["Gx"], m["Fy"] = Gx.detach(), Fy.detach()
            if self.paired:
                m["loss_trans"] = host["loss_trans"]
            return m


class CycleAE(_CycleNoGAN):
    """reference Networks.py:1350-1480: cycle loss over two plain autoencoders (+ L1(G(x), y) + L1(F(y), x) when paired)."""
    _variational = False

    def __init__(self, paired=True):
        super().__init__()
        self.F = Autoencoder()
        self.G = Autoencoder()
        self._init_common(paired)

    def forward(self, x, y):
        return self._fwd(x, y, False)

    def _fwd(self, x, y, two_streams):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        if two_streams and x.is_cuda:            # the two translation directions on two streams (CycleVAEGAN._forward_two_streams)
            ops.premeasure(x)
            ops.premeasure(y)
            fork = ops.DirectionFork(x.device)
            Gx, Fy = _ae_pair(self.G, x, self.F, y, fork)
            FGx, GFy = _ae_pair(self.F, Gx, self.G, Fy, fork)
            fork.join()
            return Gx, FGx, Fy, GFy
        Gx = self.G(x)
        FGx = self.F(Gx)
        Fy = self.F(y)
        return Gx, FGx, Fy, self.G(Fy)

    def configure_loss(self, **kwargs):
        self.loss_cycle = CycleConsistencyLoss()
        if self.paired:
            self.loss_trans = TranslationLoss()
        self.lambda_cycle = kwargs.get("lambda_cycle", 10.0)


class CycleVAE(_CycleNoGAN):
    """reference Networks.py:1482-1616: the same over two VAEs, plus the four KL terms; eps is drawn in the order
    G(x), F(G(x)), F(y), G(F(y)) (:1489-1494)."""
    _variational = True

    def __init__(self, latent_dim=64, paired=True):
        super().__init__()
        self.F = VariationalAutoencoder(latent_dim)
        self.G = VariationalAutoencoder(latent_dim)
        self._init_common(paired)

    def forward(self, x, y):
        return self._fwd(x, y, False)

    def _fwd(self, x, y, two_streams):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        if two_streams and x.is_cuda:            # as CycleVAEGAN._forward_two_streams; eps reserved in the order G(x), F(G(x)), F(y), G(F(y))
            n, _, h, w = x.shape
            shp = (n, self.G.variational_encoder_block.latent_dim, h // 16, w // 16)
            tk = ops.eps_tickets([(shp, False)] * 4, x.device)
            ops.premeasure(x)
            ops.premeasure(y)
            fork = ops.DirectionFork(x.device)
            (Gx, mu_x, logvar_x), (Fy, mu_y, logvar_y) = _vae_pair(self.G, x, tk[0], self.F, y, tk[2], fork)
            (FGx, mu_FGx, logvar_FGx), (GFy, mu_GFy, logvar_GFy) = _vae_pair(self.F, Gx, tk[1], self.G, Fy, tk[3], fork)
            fork.join()
            return Gx, FGx, Fy, GFy, mu_x, logvar_x, mu_FGx, logvar_FGx, mu_y, logvar_y, mu_GFy, logvar_GFy
        Gx, mu_x, logvar_x = self.G(x)
        FGx, mu_FGx, logvar_FGx = self.F(Gx)
        Fy, mu_y, logvar_y = self.F(y)
        GFy, mu_GFy, logvar_GFy = self.G(Fy)
        return Gx, FGx, Fy, GFy, mu_x, logvar_x, mu_FGx, logvar_FGx, mu_y, logvar_y, mu_GFy, logvar_GFy

    def configure_loss(self, **kwargs):
        self.loss_cycle = CycleConsistencyLoss()
        if self.paired:
            self.loss_trans = TranslationLoss()
        self.loss_kl = KLDivergenceLoss()
        self.lambda_kl = kwargs.get("lambda_kl", 1e-5)
        self.lambda_cycle = kwargs.get("lambda_cycle", 10.0)


class DoubleAutoencoder(_OptimizerStatesMixin, nn.Module):
    """One shared encoder, decoder_A reconstructs the source and decoder_B the target modality — the pretraining model
    for CycleAE (reference Networks.py:415-606).  The encoder runs twice per step, so its weight gradients accumulate
    from both uses (the backward kernels add into the flat gradient buffer)."""

    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.decoder_A = Decoder()
        self.decoder_B = Decoder()
        self.optimizer = None
        self.grad_reducer = None
        self.loss_fn = None

    def forward(self, x, y):
        return self._fwd(x, y, False)

    def _fwd(self, x, y, two_streams):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        if two_streams and x.is_cuda:            # the two modalities on two streams (ops.DirectionFork; the shared encoder's
            ops.premeasure(x)                    # gradients meet on the one weight-gradient stream)
            ops.premeasure(y)
            fork = ops.DirectionFork(x.device)
            a = self.decoder_A(self.encoder(x))
            with fork.second():
                b = self.decoder_B(self.encoder(y))
            fork.join()
            return a, b
        return self.decoder_A(self.encoder(x)), self.decoder_B(self.encoder(y))

    def translate_A_to_B(self, x):
        return self.decoder_B(self.encoder(ops.to_nhwc(x)))

    def translate_B_to_A(self, y):
        return self.decoder_A(self.encoder(ops.to_nhwc(y)))

    def create_cycle_ae(self):
        """reference :580-606: G (A->B) = encoder + decoder_B, F (B->A) = encoder + decoder_A."""
        cycle_ae = CycleAE().to(next(self.parameters()).device)
        cycle_ae.G.encoder.load_state_dict(self.encoder.state_dict())
        cycle_ae.G.decoder.load_state_dict(self.decoder_B.state_dict())
        cycle_ae.F.encoder.load_state_dict(self.encoder.state_dict())
        cycle_ae.F.decoder.load_state_dict(self.decoder_A.state_dict())
        return cycle_ae

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999)):
        self.optimizer = FusedAdam(self.parameters(), lr=lr, betas=betas)
        return self.optimizer

    def configure_loss(self, **kwargs):
        self.loss_fn = TranslationLoss()

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        Gx, Gy = self._fwd(x, y, ops.two_directions())
        t = {"loss_recon_A": self.loss_fn(Gx, x), "loss_recon_B": self.loss_fn(Gy, y)}
        t["G_loss"] = ops.weighted_sum([t["loss_recon_A"], t["loss_recon_B"]], [1.0, 1.0])
        return t, x, y

    def training_step(self, batch):
        if self.loss_fn is None:
            raise ValueError("Loss function has not been configured yet.")
        if self.optimizer is None:
            raise ValueError("Optimizer has not been configured yet.")
        t, _, _ = self._losses(batch)
        _backward_and_step(t["G_loss"], self.optimizer, self.grad_reducer)
        h = _metrics_to_host(t, self.grad_reducer)
        return {"G_loss": h["G_loss"], "loss_recon_A": h["loss_recon_A"], "loss_recon_B": h["loss_recon_B"], "total_loss": h["G_loss"]}

    def validation_step(self, batch):
        if self.loss_fn is None:
            raise ValueError("Loss function has not been configured yet.")
        with torch.no_grad():
            t, x, y = self._losses(batch)
            h = _metrics_to_host(t)
            return {"G_loss": h["G_loss"], "total_loss": h["G_loss"], "loss_recon_A": h["loss_recon_A"],
                    "loss_recon_B": h["loss_recon_B"], "Gx": self.translate_A_to_B(x), "Fy": self.translate_B_to_A(y)}


class DoubleVariationalAutoencoder(_OptimizerStatesMixin, nn.Module):
    """Shared encoder, one VAE bottleneck and one decoder per modality — the pretraining model for CycleVAE / CycleVAEGAN
    (reference Networks.py:608-852).  eps draws per forward: block A on enc(x), then block B on enc(y); validation adds one
    per translation."""

    def __init__(self, latent_dim=64):
        super().__init__()
        self.encoder = Encoder()
        self.vae_encoder_block_A = VariationalEncoderBlock(in_channels=1024, latent_dim=latent_dim)
        self.vae_encoder_block_B = VariationalEncoderBlock(in_channels=1024, latent_dim=latent_dim)
        self.vae_decoder_block_A = VariationalDecoderBlock(latent_dim=latent_dim, out_channels=1024)
        self.vae_decoder_block_B = VariationalDecoderBlock(latent_dim=latent_dim, out_channels=1024)
        self.decoder_A = Decoder()
        self.decoder_B = Decoder()
        self.optimizer = None
        self.grad_reducer = None
        self.loss_trans_fn = None
        self.loss_kl_fn = None
        self.lambda_kl = 0
        self.apply(self._init_weights)

    def _init_weights(self, module):
        _kaiming_relu_init(module)

    def forward(self, x, y):
        return self._fwd(x, y, False)

    def _fwd(self, x, y, two_streams):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        if two_streams and x.is_cuda:            # the two modalities on two streams; eps: block A's draw, then block B's
            n, _, h, w = x.shape
            shp = (n, self.vae_encoder_block_A.latent_dim, h // 16, w // 16)
            tk = ops.eps_tickets([(shp, False)] * 2, x.device)
            ops.premeasure(x)
            ops.premeasure(y)
            fork = ops.DirectionFork(x.device)
            with ops.use_ticket(tk[0]):
                z_x, mu_x, logvar_x = self.vae_encoder_block_A(self.encoder(x))
            Gx = self.decoder_A(self.vae_decoder_block_A(z_x))
            with fork.second():
                with ops.use_ticket(tk[1]):
                    z_y, mu_y, logvar_y = self.vae_encoder_block_B(self.encoder(y))
                Gy = self.decoder_B(self.vae_decoder_block_B(z_y))
            fork.join()
            return Gx, Gy, mu_x, logvar_x, mu_y, logvar_y
        encoded_x = self.encoder(x)
        encoded_y = self.encoder(y)
        z_x, mu_x, logvar_x = self.vae_encoder_block_A(encoded_x)
        z_y, mu_y, logvar_y = self.vae_encoder_block_B(encoded_y)
        Gx = self.decoder_A(self.vae_decoder_block_A(z_x))
        Gy = self.decoder_B(self.vae_decoder_block_B(z_y))
        return Gx, Gy, mu_x, logvar_x, mu_y, logvar_y

    def translate_A_to_B(self, x):
        z, _, _ = self.vae_encoder_block_B(self.encoder(ops.to_nhwc(x)))
        return self.decoder_B(self.vae_decoder_block_B(z))

    def translate_B_to_A(self, y):
        z, _, _ = self.vae_encoder_block_A(self.encoder(ops.to_nhwc(y)))
        return self.decoder_A(self.vae_decoder_block_A(z))

    def create_cycle_vae(self):
        """reference :701-736: G = encoder + VAE blocks B + decoder_B, F = encoder + VAE blocks A + decoder_A."""
        cycle_vae = CycleVAE(latent_dim=self.vae_encoder_block_A.latent_dim).to(next(self.parameters()).device)
        for gen, sfx in ((cycle_vae.G, "B"), (cycle_vae.F, "A")):
            gen.encoder.load_state_dict(self.encoder.state_dict())
            gen.variational_encoder_block.load_state_dict(getattr(self, "vae_encoder_block_" + sfx).state_dict())
            gen.variational_decoder_block.load_state_dict(getattr(self, "vae_decoder_block_" + sfx).state_dict())
            gen.decoder.load_state_dict(getattr(self, "decoder_" + sfx).state_dict())
        return cycle_vae

    def configure_optimizers(self, lr=1e-4, betas=(0.5, 0.999)):
        self.optimizer = FusedAdam(self.parameters(), lr=lr, betas=betas)
        return self.optimizer

    def configure_loss(self, **kwargs):
        self.loss_trans_fn = TranslationLoss()
        self.loss_kl_fn = KLDivergenceLoss()
        self.lambda_kl = kwargs.get("lambda_kl", 1e-5)

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        Gx, Gy, mu_x, logvar_x, mu_y, logvar_y = self._fwd(x, y, ops.two_directions())
        t = {"loss_recon_A": self.loss_trans_fn(Gx, x), "loss_recon_B": self.loss_trans_fn(Gy, y),
             "loss_kl_A": self.loss_kl_fn(mu_x, logvar_x), "loss_kl_B": self.loss_kl_fn(mu_y, logvar_y)}
        t["loss_kl"] = ops.weighted_sum([t["loss_kl_A"], t["loss_kl_B"]], [1.0, 1.0])
        t["G_loss"] = ops.weighted_sum([t["loss_recon_A"], t["loss_recon_B"], t["loss_kl"]], [1.0, 1.0, self.lambda_kl])
        return t, x, y

    def training_step(self, batch):
        if self.loss_trans_fn is None or self.loss_kl_fn is None:
            raise ValueError("Loss functions have not been configured yet.")
        if self.optimizer is None:
            raise ValueError("Optimizer has not been configured yet.")
        t, _, _ = self._losses(batch)
        _backward_and_step(t["G_loss"], self.optimizer, self.grad_reducer)
        h = _metrics_to_host(t, self.grad_reducer)
        return {"G_loss": h["G_loss"], "loss_recon_A": h["loss_recon_A"], "loss_recon_B": h["loss_recon_B"], "loss_kl": h["loss_kl"],
                "loss_kl_A": h["loss_kl_A"], "loss_kl_B": h["loss_kl_B"], "total_loss": h["G_loss"]}

    def validation_step(self, batch):
        if self.loss_trans_fn is None or self.loss_kl_fn is None:
            raise ValueError("Loss functions have not been configured yet.")
        with torch.no_grad():
            t, x, y = self._losses(batch)
            h = _metrics_to_host(t)
            return {"G_loss": h["G_loss"], "total_loss": h["G_loss"], "loss_recon_A": h["loss_recon_A"],
                    "loss_recon_B": h["loss_recon_B"], "loss_kl": h["loss_kl"], "loss_kl_A": h["loss_kl_A"],
                    "loss_kl_B": h["loss_kl_B"], "Gx": self.translate_A_to_B(x), "Fy": self.translate_B_to_A(y)}


class _SingleGAN(nn.Module):
    """One generator G: X->Y and one discriminator D on Y, alternating G / D updates — the shared step of AEGAN and VAEGAN
    (reference Networks.py:991-1348).  As in CycleVAEGAN the discriminator runs once per step: the G phase takes its data
    gradient, the D phase its weight gradients from the same activations.  VAEGAN is written that way in the reference
    (`DGx.detach()`, `retain_graph`, :1277-1287); AEGAN re-runs D on the detached G(x) after the generator update
    (:1105-1108), which reproduces the same outputs because that update does not touch D."""

    def configure_optimizers(self, lr=2e-4, betas=(0.5, 0.999)):
        self.optimizer_G = FusedAdam(self.G.parameters(), lr=lr, betas=betas)
        self.optimizer_D = FusedAdam(self.D.parameters(), lr=lr, betas=betas)
        return self.optimizer_G, self.optimizer_D

    def save_optimizer_states(self):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        return {"optimizer_G": self.optimizer_G.state_dict(), "optimizer_D": self.optimizer_D.state_dict()}

    def load_optimizer_states(self, states):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        for name in ("optimizer_G", "optimizer_D"):
            if name not in states:
                raise KeyError(f"{name} state not found in states")
        self.optimizer_G.load_state_dict(states["optimizer_G"])
        self.optimizer_D.load_state_dict(states["optimizer_D"])

    def _gan_terms(self, t, DGx, Dy):
        """LSGAN terms on one discriminator pass: generator (real -> 0, fake -> 1, Losses.py:67-83) and discriminator
        (real -> 1, fake -> 0, :86-102) objectives, plus the output means AEGAN logs."""
        t["gan_g_real"], t["d_y_mean"] = ops.mse_const(Dy, 0.0)
        t["gan_g_fake"], t["d_gx_mean"] = ops.mse_const(DGx, 1.0)
        t["gan_g"] = ops.weighted_sum([t["gan_g_real"], t["gan_g_fake"]], [1.0, 1.0])
        t["D_loss_real"], _ = ops.mse_const(Dy, 1.0)
        t["D_loss_fake"], _ = ops.mse_const(DGx, 0.0)
        t["D_loss"] = ops.weighted_sum([t["D_loss_real"], t["D_loss_fake"]], [1.0, 1.0])

    def _alternating_step(self, t):
        g_params, d_params = self.optimizer_G.params, self.optimizer_D.params
        red = self.grad_reducer
        self.optimizer_G.zero_grad()
        if red is not None:
            red.begin(self.optimizer_G)
        with ops.no_wgrad(d_params):
            ops.backward_overlapped(t["G_loss"], inputs=g_params, retain_graph=True)
        if red is not None:
            red.start(self.optimizer_G)
        self.optimizer_D.zero_grad()
        if red is not None:
            red.begin(self.optimizer_D)
        with ops.no_dgrad([self.D.model[0]._spec]):
            ops.backward_overlapped(t.get("D_loss_backward", t["D_loss"]), inputs=d_params)
        if red is not None:
            red.start(self.optimizer_D)
            red.finish(self.optimizer_G)
        self.optimizer_G.step()
        if red is not None:
            red.finish(self.optimizer_D)
        self.optimizer_D.step()


class AEGAN(_SingleGAN):
    """Autoencoder generator + discriminator: L1(G(x), y) + lambda_gan * LSGAN + lambda_identity * L1(G(y), y)
    (reference Networks.py:991-1188)."""

    def __init__(self):
        super().__init__()
        self.G = Autoencoder()
        self.D = Discriminator()
        self.apply(self._init_weights_)
        self.optimizer_G = None
        self.optimizer_D = None
        self.grad_reducer = None
        self.loss_trans_fn = None
        self.loss_gan_gen_fn = None
        self.loss_gan_disc_fn = None
        self.loss_identity_fn = None
        self.lambda_gan = 0
        self.lambda_identity = 0

    def _init_weights_(self, module):
        _kaiming_relu_init(module)

    def forward(self, x, y):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        Gx = self.G(x)
        Gy = self.G(y)
        return Gx, Gy, self.D(Gx), self.D(y)

    def configure_loss(self, **kwargs):
        self.loss_trans_fn = TranslationLoss()
        self.loss_gan_gen_fn = GANLossGenerator()
        self.loss_gan_disc_fn = GANLossDiscriminator()
        self.loss_identity_fn = TranslationLoss()
        self.lambda_gan = kwargs.get("lambda_gan", 1.0)
        self.lambda_identity = kwargs.get("lambda_identity", 5.0)

    def _check_configured(self):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        if self.loss_trans_fn is None:
            raise ValueError("Translation loss function has not been configured yet.")
        if self.loss_gan_gen_fn is None:
            raise ValueError("GAN generator loss function has not been configured yet.")
        if self.loss_gan_disc_fn is None:
            raise ValueError("GAN discriminator loss function has not been configured yet.")
        if self.loss_identity_fn is None:
            raise ValueError("Identity loss function has not been configured yet.")

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        Gx, Gy, DGx, Dy = self(x, y)
        t = {"loss_trans": self.loss_trans_fn(Gx, y), "loss_identity": self.loss_identity_fn(Gy, y)}
        self._gan_terms(t, DGx, Dy)
        t["G_loss"] = ops.weighted_sum([t["loss_trans"], t["gan_g"], t["loss_identity"]], [1.0, self.lambda_gan, self.lambda_identity])
        return t, Gx

    def training_step(self, batch):
        self._check_configured()
        t, _ = self._losses(batch)
        self._alternating_step(t)
        h = _metrics_to_host(t, self.grad_reducer)
        return {"G_loss": h["G_loss"], "D_loss": h["D_loss"], "D_loss_real": h["D_loss_real"], "D_loss_fake": h["D_loss_fake"],
                "loss_trans": h["loss_trans"], "loss_gan_g": h["gan_g"], "loss_identity": h["loss_identity"],
                "d_y_mean": h["d_y_mean"], "d_gx_mean": h["d_gx_mean"]}

    def validation_step(self, batch):
        self._check_configured()            # the reference's AEGAN wants its optimizers even to validate (:1145)
        with torch.no_grad():
            t, Gx = self._losses(batch)
            h = _metrics_to_host(t)
            return {"total_loss": h["G_loss"] + h["D_loss"], "G_loss": h["G_loss"], "D_loss": h["D_loss"],
                    "D_loss_real": h["D_loss_real"], "D_loss_fake": h["D_loss_fake"], "loss_trans": h["loss_trans"],
                    "loss_gan_g": h["gan_g"], "loss_gan_g_real": h["gan_g_real"], "loss_gan_g_fake": h["gan_g_fake"],
                    "loss_identity": h["loss_identity"], "Gx": Gx}


class VAEGAN(_SingleGAN):
    """VAE generator + discriminator: lambda_recon * L1(G(x), y) + lambda_gan * LSGAN + lambda_identity * L1(G(y), y) +
    lambda_kl * KL(mu_x, logvar_x)  (reference Networks.py:1190-1348; eps is drawn for G(x), then for G(y))."""

    def __init__(self, latent_dim=64):
        super().__init__()
        self.G = VariationalAutoencoder(latent_dim)
        self.D = Discriminator()
        self.latent_dim = latent_dim
        self.debug_mode = False
        self.debug_info = {}
        self.optimizer_G = None             # (left unset by the reference's __init__)
        self.optimizer_D = None
        self.grad_reducer = None

    def forward(self, x, y):
        x, y = ops.to_nhwc(x), ops.to_nhwc(y)
        Gx, mu, logvar = self.G(x)
        Gy, mu_y, logvar_y = self.G(y)
        return Gx, mu, logvar, Gy, mu_y, logvar_y, self.D(Gx), self.D(y)

    def configure_loss(self, **kwargs):
        self.translation_loss = TranslationLoss()
        self.gan_loss_gen = GANLossGenerator()
        self.gan_loss_disc = GANLossDiscriminator()
        self.identity_loss = TranslationLoss()
        self.kl_loss = KLDivergenceLoss()
        self.lambda_gan = kwargs.get("lambda_gan", 1.0)
        self.lambda_identity = kwargs.get("lambda_identity", 5.0)
        self.lambda_kl = kwargs.get("lambda_kl", 1e-5)
        self.lambda_recon = kwargs.get("lambda_recon", 1.0)

    def enable_debug_mode(self, enabled=True):
        self.debug_mode = enabled

    def _losses(self, batch):
        x, y = ops.to_nhwc(batch["x"]), ops.to_nhwc(batch["y"])
        Gx, mu, logvar, Gy, _, _, DGx, Dy = self(x, y)
        t = {"loss_trans": self.translation_loss(Gx, y), "loss_identity": self.identity_loss(Gy, y),
             "loss_kl": self.kl_loss(mu, logvar)}
        self._gan_terms(t, DGx, Dy)
        t["G_loss"] = ops.weighted_sum([t["loss_trans"], t["gan_g"], t["loss_identity"], t["loss_kl"]],
                                       [self.lambda_recon, self.lambda_gan, self.lambda_identity, self.lambda_kl])
        # the reference detaches the discriminator OUTPUT of the fake branch (`gan_loss_disc(Dy, DGx.detach())`, :1277),
        # not its input: the fake term is a constant in D_loss and only (1 - D(y))^2 reaches D's parameters.  Kept as
        # written — D_loss reports both terms, the backward pass sees the real one.
        t["D_loss_backward"] = t["D_loss_real"]
        return t, Gx

    def training_step(self, batch):
        if self.optimizer_G is None or self.optimizer_D is None:
            raise ValueError("Optimizers have not been configured yet.")
        t, _ = self._losses(batch)
        self._alternating_step(t)
        h = _metrics_to_host(t, self.grad_reducer)
        m = {"G_loss": h["G_loss"], "D_loss": h["D_loss"], "loss_gan_disc_real": h["D_loss_real"],
             "loss_gan_disc_fake": h["D_loss_fake"], "loss_trans": h["loss_trans"], "loss_gan_real": h["gan_g_real"],
             "loss_gan_fake": h["gan_g_fake"], "loss_identity": h["loss_identity"], "loss_kl": h["loss_kl"]}
        if self.debug_mode:
            m["debug_info"] = self.debug_info
        return m

    def validation_step(self, batch):
        with torch.no_grad():
            t, Gx = self._losses(batch)
            h = _metrics_to_host(t)
            return {"total_loss": h["G_loss"] + h["D_loss"], "G_loss": h["G_loss"], "D_loss": h["D_loss"],
                    "loss_trans": h["loss_trans"], "loss_gan_real": h["gan_g_real"], "loss_gan_fake": h["gan_g_fake"],
                    "loss_identity": h["loss_identity"], "loss_kl": h["loss_kl"], "Gx": Gx}

