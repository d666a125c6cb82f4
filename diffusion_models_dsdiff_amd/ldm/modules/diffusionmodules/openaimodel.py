"""UNetModel — drop-in for ldm/modules/diffusionmodules/openaimodel.py:571-958, the plain single-stream denoiser that the
latent path runs on the VAE latents (configs/v2-1-stable-unclip-h-inference.yaml:33-50: no spatial transformer, no class
embedding); ``use_spatial_transformer=True`` (cross-attention on ``context``, transformer_depth 1) is built as well.  Same constructor keywords, ``state_dict`` names and ``forward(x, timesteps, context=None, y=None)``; the
arithmetic runs in libdsdiff.so (DSD_BLOCK_UNET, include/dsdiff.h) on the same kernels as the four-stream DSUnetModel.
The reference's building blocks are re-exported under their original names.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from .... import _lib
from ...._lib import check, dptr, lib, stream_ptr
from ....blocks import _Block, ResBlock, AttentionBlock, Upsample, Downsample  # noqa: F401  (reference import paths)


class UNetModel(_Block):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 use_bf16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, disable_self_attentions=None, num_attention_blocks=None,
                 disable_middle_self_attn=False, use_linear_in_transformer=False, adm_in_channels=None, device_index=0):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None, 'use_spatial_transformer=True needs context_dim (the width of the cross-attention conditioning)'
            if isinstance(context_dim, (list, tuple)):      # (a ListConfig in the reference: one entry per transformer block)
                context_dim = list(context_dim)
                assert len(set(context_dim)) == 1, "one context width for all transformer blocks"
                context_dim = context_dim[0]
            if transformer_depth != 1:
                # the reference builds SpatialTransformer(..., depth, context_dim=<int>) and indexes context_dim[d] / context[d]:
                # with one conditioning tensor (openaimodel.py:946-952) only depth 1 runs there either
                raise NotImplementedError("UNetModel(use_spatial_transformer=True) takes transformer_depth=1 (one context tensor)")
        if context_dim is not None:
            assert use_spatial_transformer, 'context_dim is only meaningful with use_spatial_transformer=True'
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        if num_classes is not None or n_embed is not None:
            raise NotImplementedError("class-conditional / codebook heads are not on the sampling hot path")
        if dims != 2 or not conv_resample or dropout != 0 or use_fp16 or use_bf16:
            raise NotImplementedError("hot path is dims=2, conv_resample=True, dropout=0, fp32 (SURVEY.md 8)")
        if disable_self_attentions is not None or num_attention_blocks is not None or disable_middle_self_attn:
            raise NotImplementedError("disable_self_attentions / num_attention_blocks / disable_middle_self_attn are unused by every shipped yaml")
        channel_mult = list(channel_mult)
        nrb = len(channel_mult) * [num_res_blocks] if isinstance(num_res_blocks, int) else list(num_res_blocks)
        if len(nrb) != len(channel_mult):
            raise ValueError("provide num_res_blocks either as an int (globally constant) or "
                             "as a list/tuple (per-level) with the same length as channel_mult")
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = nrb, list(attention_resolutions), channel_mult
        self.num_classes = None
        self.dtype = torch.float32
        ar = self.attention_resolutions
        self.use_spatial_transformer, self.context_dim = bool(use_spatial_transformer), context_dim
        st = [1, int(transformer_depth), int(context_dim), int(bool(use_linear_in_transformer))] if use_spatial_transformer else []
        self._create(_lib.BLOCK_UNET, [in_channels, model_channels, out_channels, num_heads, num_head_channels, num_heads_upsample,
                                       int(bool(use_scale_shift_norm)), int(bool(resblock_updown)), int(bool(use_new_attention_order)),
                                       int(bool(legacy)), len(channel_mult)] + channel_mult + nrb + [len(ar)] + ar + st, device_index)
        half = model_channels // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half).contiguous()
        check(lib().dsd_set_timestep_freqs(self._h, C.c_void_p(freqs.data_ptr()), half))

    def _out_shape(self, x):
        return (x.shape[0], self.out_channels, x.shape[2], x.shape[3])

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """openaimodel.py:926-958."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if not x.is_cuda:
            raise _lib.DsdError("UNetModel runs on the MI355X only (no CPU fallback)")
        self.sync_params()
        x = x.float().contiguous()
        B, Cc, H, W = x.shape
        t = timesteps.to(x.device).float().contiguous()
        assert t.shape == (B,)
        out = torch.empty(self._out_shape(x), device=x.device, dtype=torch.float32)
        if self.use_spatial_transformer:
            if context is None:
                raise ValueError("UNetModel(use_spatial_transformer=True): context [B, tokens, context_dim] is required")
            ctx = context.to(x.device).float().contiguous()
            assert ctx.dim() == 3 and ctx.shape[0] == B and ctx.shape[2] == self.context_dim, tuple(ctx.shape)
            check(lib().dsd_block_forward(self._h, dptr(x), B, Cc, H, W, dptr(t), 1, dptr(ctx), ctx.shape[1], dptr(out), stream_ptr()))
        else:
            assert context is None, "context is only consumed by the spatial-transformer variant"
            check(lib().dsd_block_forward(self._h, dptr(x), B, Cc, H, W, dptr(t), 1, None, 0, dptr(out), stream_ptr()))
        return out

    def convert_to_fp16(self):
        raise NotImplementedError("the hot path is fp32 end-to-end (SURVEY.md 9, quirk 8)")

    def convert_to_fp32(self):
        return None
