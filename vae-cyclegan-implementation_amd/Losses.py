"""Atomic losses with the reference's class surface (Losses.py:14-121 of the reference),
each reduction a HIP kernel (wavefront shuffle -> block partial -> ordered final sum).

Only the six atomic classes exist here: the reference's composite loss classes
(Losses.py:123-379) are dead code there and are not part of the training step.
"""
import torch.nn as nn

from . import ops


class TranslationLoss(nn.Module):
    """mean |generated - target|  (reference Losses.py:14-24)."""

    def forward(self, generated, target):
        return ops.l1_loss(generated, target)


class CycleConsistencyLoss(nn.Module):
    """mean|F(G(x)) - x| + mean|G(F(y)) - y|  (reference Losses.py:27-39)."""

    def forward(self, x, y, FGx, GFy):
        return ops.weighted_sum([ops.l1_loss(FGx, x), ops.l1_loss(GFy, y)], [1.0, 1.0])


class IdentityLoss(nn.Module):
    """mean|F(x) - x| + mean|G(y) - y|  (reference Losses.py:42-65)."""

    def forward(self, x, y, Fx, Gy):
        return ops.weighted_sum([ops.l1_loss(Fx, x), ops.l1_loss(Gy, y)], [1.0, 1.0])


class GANLossGenerator(nn.Module):
    """LSGAN generator terms: real -> 0, fake -> 1; returns (total, real, fake)  (reference Losses.py:67-83)."""

    def forward(self, D_real, D_fake):
        real_loss, _ = ops.mse_const(D_real, 0.0)
        fake_loss, _ = ops.mse_const(D_fake, 1.0)
        return ops.weighted_sum([real_loss, fake_loss], [1.0, 1.0]), real_loss, fake_loss


class GANLossDiscriminator(nn.Module):
    """LSGAN discriminator terms: real -> 1, fake -> 0; returns (total, real, fake)  (reference Losses.py:86-102)."""

    def forward(self, D_real, D_fake):
        real_loss, _ = ops.mse_const(D_real, 1.0)
        fake_loss, _ = ops.mse_const(D_fake, 0.0)
        return ops.weighted_sum([real_loss, fake_loss], [1.0, 1.0]), real_loss, fake_loss


class KLDivergenceLoss(nn.Module):
    """-0.5 * mean(1 + clamp(logvar) - mu^2 - exp(clamp(logvar)))  (reference Losses.py:105-121)."""

    def forward(self, mu, logvar):
        return ops.kl_loss(mu, logvar)
