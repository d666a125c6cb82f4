"""Encoder / Decoder of the KL-VAE — drop-ins for ldm/modules/diffusionmodules/model.py:452-544 and :546-655.

Same constructor keywords (the ``ddconfig`` of configs/autoencoder_kl_64x64x3.yaml:14-24), same ``state_dict`` names
("conv_in.weight", "down.0.block.0.norm1.weight", "mid.attn_1.q.weight", "up.2.upsample.conv.weight", ...), same
``forward``.  The modules only hold the parameters; the arithmetic — GroupNorm(32, eps 1e-6) + swish, 3x3 / 1x1
convolutions, the (0,1,0,1)-padded stride-2 Downsample, nearest x2 Upsample folded into the next convolution, single-head
attention over all channels — runs in libdsdiff.so (DSD_BLOCK_VAE_ENCODER / DSD_BLOCK_VAE_DECODER, include/dsdiff.h).
"""
from __future__ import annotations

import torch

from .... import _lib
from ....blocks import _Block


def _iargs(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution, z_channels, double_z, embed_dim,
           with_quant):
    ch_mult, attn_resolutions = list(ch_mult), list(attn_resolutions)
    return [ch, out_ch, in_channels, resolution, z_channels, int(bool(double_z)), embed_dim, num_res_blocks, int(with_quant),
            len(ch_mult)] + ch_mult + [len(attn_resolutions)] + attn_resolutions


def _check(dropout, resamp_with_conv, use_linear_attn, attn_type):
    if dropout != 0.0 or not resamp_with_conv or use_linear_attn or attn_type not in ("vanilla", "vanilla-xformers"):
        raise NotImplementedError("the native VAE is dropout=0, resamp_with_conv=True, vanilla attention "
                                  "(every shipped autoencoder yaml)")


class Encoder(_Block):
    _zero_sites = False
    _strip = "encoder."

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", embed_dim=None, with_quant=False, device_index=0, **ignore_kwargs):
        super().__init__()
        _check(dropout, resamp_with_conv, use_linear_attn, attn_type)
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self._out_c = (2 * embed_dim) if with_quant else (2 * z_channels if double_z else z_channels)
        self._create(_lib.BLOCK_VAE_ENCODER, _iargs(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution,
                                                    z_channels, double_z, embed_dim or z_channels, with_quant), device_index)

    def _out_shape(self, x):
        B, _, H, W = x.shape
        f = 2 ** (self.num_resolutions - 1)
        return (B, self._out_c, H // f, W // f)

    def forward(self, x):
        return self._call(x)


class Decoder(_Block):
    _zero_sites = False
    _strip = "decoder."

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", embed_dim=None, with_quant=False, device_index=0, **ignorekwargs):
        super().__init__()
        _check(dropout, resamp_with_conv, use_linear_attn, attn_type)
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are unused by the reference's autoencoder")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self._out_ch = resolution, in_channels, out_ch
        self.z_shape = (1, z_channels, resolution // 2 ** (self.num_resolutions - 1), resolution // 2 ** (self.num_resolutions - 1))
        self._create(_lib.BLOCK_VAE_DECODER, _iargs(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution,
                                                    z_channels, True, embed_dim or z_channels, with_quant), device_index)

    def _out_shape(self, z):
        B, _, H, W = z.shape
        f = 2 ** (self.num_resolutions - 1)
        return (B, self._out_ch, H * f, W * f)

    def forward(self, z):
        self.last_z_shape = z.shape
        return self._call(z)
