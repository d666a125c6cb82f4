"""CPU oracle of the KL-VAE (latent path) — TEST INFRASTRUCTURE ONLY (imported by tests/ and
tests/golden/gen_golden.py; never by the product).

A functional, state-dict driven restatement in torch-CPU fp32 of
  Encoder.forward / Decoder.forward      ldm/modules/diffusionmodules/model.py:519-543 / :618-655
  ResnetBlock / AttnBlock / Downsample / Upsample / Normalize     :121-149 / :185-209 / :78-86 / :57-61 / :41-42
  AutoencoderKL.encode / decode          ldm/models/autoencoder.py:138-147
  DiagonalGaussianDistribution           ldm/modules/distributions/distributions.py:24-37
Parity status: PINNED — tests/test_oracle_golden.py checks it against tests/golden/vae.npz, which gen_golden.py::gen_vae
produced by running the reference's own Encoder / Decoder / DiagonalGaussianDistribution classes on CPU (AutoencoderKL
itself needs Lightning + diffusers and does not import; its two 1x1 convolutions are applied with F.conv2d there).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class VaeConfig:
    def __init__(self, ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution, z_channels,
                 double_z=True, embed_dim=None, **_):
        self.ch, self.out_ch, self.ch_mult = ch, out_ch, list(ch_mult)
        self.num_res_blocks, self.attn_resolutions = num_res_blocks, list(attn_resolutions)
        self.in_channels, self.resolution, self.z_channels = in_channels, resolution, z_channels
        self.double_z, self.embed_dim = double_z, embed_dim if embed_dim is not None else z_channels


def _norm(sd, p, x):                       # Normalize: GroupNorm(32, eps=1e-6, affine)  model.py:41-42
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x):                             # nonlinearity  :36-38
    return x * torch.sigmoid(x)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):                # ResnetBlock.forward with temb=None  :121-149
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)), padding=1)
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)), padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):                  # AttnBlock.forward  :185-209
    h_ = _norm(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h_), _conv(sd, p + ".k", h_), _conv(sd, p + ".v", h_)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(sd, p + ".proj_out", h_)


def downsample(sd, p, x):                  # Downsample.forward (with_conv)  :78-83
    return _conv(sd, p + ".conv", F.pad(x, (0, 1, 0, 1), mode="constant", value=0), stride=2, padding=0)


def upsample(sd, p, x):                    # Upsample.forward (with_conv)  :57-61
    return _conv(sd, p + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"), padding=1)


def encoder(cfg: VaeConfig, sd, x, prefix="encoder."):       # Encoder.forward  :519-543
    e = prefix
    h = _conv(sd, e + "conv_in", x, padding=1)
    res = cfg.resolution
    L = len(cfg.ch_mult)
    for l in range(L):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"{e}down.{l}.block.{j}", h)
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f"{e}down.{l}.attn.{j}", h)
        if l != L - 1:
            h = downsample(sd, f"{e}down.{l}.downsample", h)
            res //= 2
    h = resnet_block(sd, e + "mid.block_1", h)
    h = attn_block(sd, e + "mid.attn_1", h)
    h = resnet_block(sd, e + "mid.block_2", h)
    return _conv(sd, e + "conv_out", _swish(_norm(sd, e + "norm_out", h)), padding=1)


def decoder(cfg: VaeConfig, sd, z, prefix="decoder."):       # Decoder.forward  :618-655
    d = prefix
    L = len(cfg.ch_mult)
    h = _conv(sd, d + "conv_in", z, padding=1)
    h = resnet_block(sd, d + "mid.block_1", h)
    h = attn_block(sd, d + "mid.attn_1", h)
    h = resnet_block(sd, d + "mid.block_2", h)
    res = cfg.resolution // 2 ** (L - 1)
    for l in reversed(range(L)):
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"{d}up.{l}.block.{j}", h)
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f"{d}up.{l}.attn.{j}", h)
        if l != 0:
            h = upsample(sd, f"{d}up.{l}.upsample", h)
            res *= 2
    return _conv(sd, d + "conv_out", _swish(_norm(sd, d + "norm_out", h)), padding=1)


def encode(cfg: VaeConfig, sd, x):         # AutoencoderKL.encode -> the moments  autoencoder.py:138-142
    return _conv(sd, "quant_conv", encoder(cfg, sd, x))


def decode(cfg: VaeConfig, sd, z):         # AutoencoderKL.decode  :144-147
    return decoder(cfg, sd, _conv(sd, "post_quant_conv", z))


def gaussian_sample(moments, noise):       # DiagonalGaussianDistribution.sample  distributions.py:24-37
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise
