"""Oracle for the GAN half of the autoencoder step -- TEST INFRASTRUCTURE (see oracle/__init__.py).

`PatchDiscriminator` and `PatchAdversarialLoss` come from the third-party `generative` package (train_autoencoder.py:26-27, 41, 600;
`monai-generative`, unpinned in pyproject.toml:20-33); its source is NOT under /root/reference and cannot be installed here, so both are
restated from upstream MONAI-GenerativeModels (generative/networks/nets/patchgan_discriminator.py, generative/losses/adversarial_loss.py)
as plain torch modules: PARITY UNPINNED -- nothing under /root/reference holds a vector for them.  What IS the reference's own: the call
sites (T-AE:371-397 discriminator step, :416-423 generator term) and the planner's arguments (configuration.py:966-967:
spatial_dims, in_channels, out_channels=1, num_channels=64, num_layers_d=3; everything else upstream's defaults).

Upstream structure restated here: every layer is a monai `Convolution` = Sequential(conv, adn) with ADN ordering "NDA" (norm, dropout,
activation); state_dict names `<layer>.conv.*`, `<layer>.adn.N.*`:
    initial_conv : Conv(k4, s2, p1, bias) -> LeakyReLU(0.2)
    0 .. L-1     : Conv(k4, s2 (s1 for the last), p1, no bias) -> BatchNorm -> LeakyReLU(0.2); channels double per layer
    final_conv   : Conv(k = last_conv_kernel_size or 4, s1, p (k-1)//2, bias), conv only
    initialise_weights: conv weights N(0, 0.02), BatchNorm weights N(1, 0.02) / biases 0
    forward(x) -> [output of every child]
PatchAdversarialLoss("least_squares"): MSELoss(LeakyReLU(0.05)(logits), full_like(1 real / 0 fake)) -- upstream puts that activation in
front of the least-squares criterion unless no_activation_leastsq=True.
"""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class _Conv(nn.Sequential):
    def __init__(self, sd, cin, cout, k, s, p, bias, norm, act):
        super().__init__()
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[sd]
        self.add_module("conv", conv(cin, cout, k, stride=s, padding=p, bias=bias))
        if norm or act:
            adn = nn.Sequential()
            if norm:
                adn.add_module("N", {2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[sd](cout))
            if act:
                adn.add_module("A", nn.LeakyReLU(0.2))
            self.add_module("adn", adn)


class PatchDiscriminator(nn.Sequential):
    def __init__(self, spatial_dims, num_channels, in_channels, out_channels=1, num_layers_d=3, kernel_size=4, padding=1, last_conv_kernel_size=None):
        super().__init__()
        self.num_layers_d = num_layers_d
        lk = kernel_size if last_conv_kernel_size is None else last_conv_kernel_size
        self.add_module("initial_conv", _Conv(spatial_dims, in_channels, num_channels, kernel_size, 2, padding, True, False, True))
        cin, cout = num_channels, num_channels * 2
        for l in range(num_layers_d):
            self.add_module(str(l), _Conv(spatial_dims, cin, cout, kernel_size, 1 if l == num_layers_d - 1 else 2, padding, False, True, True))
            cin, cout = cout, cout * 2
        self.add_module("final_conv", _Conv(spatial_dims, cin, out_channels, lk, 1, (lk - 1) // 2, True, False, False))
        self.apply(self._init)

    @staticmethod
    def _init(m):
        name = m.__class__.__name__
        if name.find("Conv") != -1 and hasattr(m, "weight"):
            nn.init.normal_(m.weight.data, 0.0, 0.02)
        elif name.find("BatchNorm") != -1:
            nn.init.normal_(m.weight.data, 1.0, 0.02)
            nn.init.constant_(m.bias.data, 0)

    def forward(self, x):
        out = [x]
        for child in self.children():
            out.append(child(out[-1]))
        return out[1:]


class PatchAdversarialLoss:
    def __init__(self, criterion="least_squares", no_activation_leastsq=False):
        assert criterion == "least_squares"
        self.slope = None if no_activation_leastsq else 0.05

    def __call__(self, logits, target_is_real, for_discriminator=False):
        a = logits.float() if self.slope is None else F.leaky_relu(logits.float(), self.slope)
        return F.mse_loss(a, torch.full_like(a, 1.0 if target_is_real else 0.0))


def generator_loss(ae, disc, adv, images, eps, kl_weight, adv_weight):
    """T-AE:406-423 without the perceptual term: L1 + kl_weight KL + adv_weight LS(D(recon)[-1], real)."""
    from .step import ae_loss
    base, recon, _, _ = ae_loss(ae, images, eps, kl_weight)
    gen = adv(disc(recon.contiguous().float())[-1], target_is_real=True, for_discriminator=False) * adv_weight
    return base + gen, recon, gen


def discriminator_loss(disc, adv, images, recon, adv_weight):
    """T-AE:376-383: adv_weight * 0.5 * (LS(D(recon.detach()), fake) + LS(D(images), real)); two separate forward calls (BatchNorm statistics
    per call)."""
    lf = adv(disc(recon.contiguous().detach())[-1], target_is_real=False, for_discriminator=True)
    lr = adv(disc(images.contiguous().detach())[-1], target_is_real=True, for_discriminator=True)
    return (lf + lr) * 0.5 * adv_weight
