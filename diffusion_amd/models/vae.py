"""Frozen SD-2 ``AutoencoderKL`` on stock PyTorch-ROCm (north_star keeps VAE-encode on PyTorch-ROCm).

Stands where diffusion/models/models.py:81,84 load ``diffusers.AutoencoderKL``; the reference calls
``vae.encode(x)['latent_dist'].sample()`` (stable_diffusion.py:167,171) and ``vae.decode(z).sample`` (:380).
Architecture restated from the published SD VAE config (block_out_channels 128/256/512/512, 2 layers per block,
latent_channels 4, GroupNorm 32, single-head mid attention); parameter names follow diffusers so a local
checkpoint loads with ``load_state_dict``.  Random-init when no local weights are given (no network here)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Res(nn.Module):

    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class _Attn(nn.Module):

    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.to_q(y)[:, None], self.to_k(y)[:, None], self.to_v(y)[:, None])[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)


class _Resample(nn.Module):
    """diffusers Downsample2D / Upsample2D keep their convolution under ``.conv`` (state_dict key ``....0.conv.weight``)."""

    def __init__(self, c, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=padding)

    def forward(self, x):
        return self.conv(x)


class _Mid(nn.Module):

    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(c, c), _Res(c, c)])
        self.attentions = nn.ModuleList([_Attn(c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Down(nn.Module):

    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin, cout), _Res(cout, cout)])
        self.downsamplers = nn.ModuleList([_Resample(cout, 2, 0)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](F.pad(x, (0, 1, 0, 1)))
        return x


class _Up(nn.Module):

    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout) for i in range(3)])
        self.upsamplers = nn.ModuleList([_Resample(cout, 1, 1)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](F.interpolate(x, scale_factor=2.0, mode='nearest'))
        return x


class _Encoder(nn.Module):

    def __init__(self, ch=(128, 256, 512, 512), zc=4):
        super().__init__()
        self.conv_in = nn.Conv2d(3, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList(
            [_Down(ch[max(i - 1, 0)], ch[i], i < len(ch) - 1) for i in range(len(ch))])
        self.mid_block = _Mid(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * zc, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class _Decoder(nn.Module):

    def __init__(self, ch=(128, 256, 512, 512), zc=4):
        super().__init__()
        rev = tuple(reversed(ch))
        self.conv_in = nn.Conv2d(zc, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        self.up_blocks = nn.ModuleList(
            [_Up(rev[max(i - 1, 0)], rev[i], i < len(ch) - 1) for i in range(len(ch))])
        self.conv_norm_out = nn.GroupNorm(32, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussian:

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, device=self.mean.device, dtype=self.mean.dtype,
                                                  generator=generator)

    def mode(self):
        return self.mean


class _Out(dict):

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class AutoencoderKL(nn.Module):

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4):
        super().__init__()
        self.encoder = _Encoder(block_out_channels, latent_channels)
        self.decoder = _Decoder(block_out_channels, latent_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def encode(self, x):
        return _Out(latent_dist=DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return _Out(sample=self.decoder(self.post_quant_conv(z)))
