"""Single-block modules behind the C ABI (dsd_block_*): native counterparts of the reference's building blocks
with the reference's constructor signatures and parameter names.

ResBlock / AttentionBlock / Upsample / Downsample  <- ldm/modules/diffusionmodules/openaimodel.py:93-164,167-284,426-473
FeatureDisentangle                                 <- UNet_DS_Diff/model.py:152-168
SE_Attention                                       <- Disc_diff/guided_diffusion/unet.py:82-109
CrossAttention / FeedForward / BasicTransformerBlock / SpatialTransformer <- ldm/modules/attention.py
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, dptr, lib, stream_ptr
from .UNet_DS_Diff.model import NativeModule


class _Block(NativeModule):
    token = False

    def _create(self, kind, iargs, device_index=0):
        _lib.require_gpu(device_index)
        arr = (C.c_int32 * len(iargs))(*[int(v) for v in iargs])
        check(lib().dsd_block_create(kind, arr, len(iargs), device_index, C.byref(self._h)))
        import os
        if os.environ.get("DSD_PRECISION"):
            self.set_precision(os.environ["DSD_PRECISION"])
        self._build_params()

    def _out_shape(self, x):
        return x.shape

    @torch.no_grad()
    def _call(self, x, aux=None, aux2=None):
        if not x.is_cuda:
            raise _lib.DsdError("native blocks run on the MI355X only (no CPU fallback)")
        self.sync_params()
        x = x.float().contiguous()
        if self.token:
            B, N, Cc = x.shape
            dims = (B, Cc, N, 1)
        else:
            B, Cc, H, W = x.shape
            dims = (B, Cc, H, W)
        out = torch.empty(self._out_shape(x), device=x.device, dtype=torch.float32)
        a1 = aux.float().contiguous() if aux is not None else None
        a2 = aux2.float().contiguous() if aux2 is not None else None
        l1 = 0 if a1 is None else a1.shape[1]
        l2 = 0 if a2 is None else a2.shape[1]
        check(lib().dsd_block_forward(self._h, dptr(x), *dims, dptr(a1), l1, dptr(a2), l2, dptr(out), stream_ptr()))
        return out


class ResBlock(_Block):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert dims == 2 and not use_conv and dropout == 0
        self.channels, self.out_channels, self.up, self.down = channels, out_channels or channels, up, down
        self._create(_lib.BLOCK_RES, [channels, self.out_channels, emb_channels, int(use_scale_shift_norm), int(up), int(down)])

    def _out_shape(self, x):
        B, _, H, W = x.shape
        s = 2 if self.up else 1
        d = 2 if self.down else 1
        return (B, self.out_channels, H * s // d, W * s // d)

    def forward(self, x, emb):
        return self._call(x, emb)


class AttentionBlock(_Block):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        if num_head_channels == -1:
            heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            heads = channels // num_head_channels
        self.num_heads = heads
        self._create(_lib.BLOCK_ATTN, [channels, heads, int(use_new_attention_order)])

    def forward(self, x):
        return self._call(x)


class Upsample(_Block):
    def __init__(self, channels, use_conv=True, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2 and (out_channels in (None, channels))
        self._create(_lib.BLOCK_UPSAMPLE, [channels])

    def _out_shape(self, x):
        B, Cc, H, W = x.shape
        return (B, Cc, H * 2, W * 2)

    def forward(self, x):
        return self._call(x)


class Downsample(_Block):
    def __init__(self, channels, use_conv=True, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2 and (out_channels in (None, channels))
        self._create(_lib.BLOCK_DOWNSAMPLE, [channels])

    def _out_shape(self, x):
        B, Cc, H, W = x.shape
        return (B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1)

    def forward(self, x):
        return self._call(x)


class FeatureDisentangle(_Block):
    def __init__(self, in_channels, half_conv_ch):
        super().__init__()
        self.half = half_conv_ch
        self._create(_lib.BLOCK_DISENTANGLE, [in_channels, half_conv_ch])

    def _out_shape(self, x):
        B, _, H, W = x.shape
        return (B, self.half, H, W)

    def forward(self, x):
        return self._call(x)


class SE_Attention(_Block):
    def __init__(self, channel=512, reduction=16):
        super().__init__()
        self._create(_lib.BLOCK_SE, [channel, reduction])

    def forward(self, x):
        return self._call(x)


class CrossAttention(_Block):
    token = True

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        self._create(_lib.BLOCK_CROSSATTN, [query_dim, context_dim or query_dim, heads, dim_head])

    def forward(self, x, context=None, mask=None):
        assert mask is None
        return self._call(x, context)


class FeedForward(_Block):
    token = True

    def __init__(self, dim, dim_out=None, mult=4, glu=True, dropout=0.):
        super().__init__()
        assert glu and dim_out in (None, dim), "only the GEGLU variant is used (BasicTransformerBlock gated_ff=True)"
        self._create(_lib.BLOCK_FF_GEGLU, [dim, mult])

    def forward(self, x):
        return self._call(x)


class BasicTransformerBlock(_Block):
    token = True

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        assert gated_ff and not disable_self_attn
        self._create(_lib.BLOCK_BASIC_TRANSFORMER, [dim, n_heads, d_head, context_dim or dim])

    def forward(self, x, context=None):
        return self._call(x, context)


class SpatialTransformer(_Block):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True):
        super().__init__()
        if context_dim is not None and not isinstance(context_dim, list):
            context_dim = [context_dim]
        cd = context_dim[0] if context_dim else n_heads * d_head
        assert not context_dim or all(c == cd for c in context_dim)
        self._create(_lib.BLOCK_SPATIAL_TRANSFORMER, [in_channels, n_heads, d_head, depth, cd, int(use_linear)])

    def forward(self, x, context=None):
        if not isinstance(context, list):
            context = [context]
        return self._call(x, context[0], context[1] if len(context) > 1 else None)
