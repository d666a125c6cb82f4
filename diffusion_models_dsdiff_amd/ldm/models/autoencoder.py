"""AutoencoderKL — drop-in for the sampling-side API of ldm/models/autoencoder.py:26-158 (encode / decode / forward).

Construction mirrors the reference: ``AutoencoderKL(ddconfig, lossconfig, embed_dim, ...)``; unless ``train_from_hgf`` the
reference forces ``in_channels = out_ch = 1`` (autoencoder.py:46-48) and so does this class.  ``state_dict`` names are the
reference's ("encoder.*", "decoder.*", "quant_conv.*", "post_quant_conv.*"), so its checkpoints load unchanged (the loss /
discriminator / EMA entries of a training checkpoint are ignored with ``strict=False`` exactly as init_from_ckpt does).
Training (losses, optimisers, Lightning hooks, data plumbing) is out of scope: SURVEY.md section 8.

Two native handles do the work: encoder + quant_conv and post_quant_conv + decoder (include/dsdiff.h,
DSD_BLOCK_VAE_ENCODER / DSD_BLOCK_VAE_DECODER).  The nn.Parameters registered here ARE the handles' parameters (shared
objects), so load_state_dict / .to() / in-place edits reach the library through the usual sync.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution
from ...UNet_DS_Diff.model import _Holder


class _EncoderQ(Encoder):
    _strip = ""      # keep "encoder." / "quant_conv." as in the AutoencoderKL state_dict


class _DecoderQ(Decoder):
    _strip = ""


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=None, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, ema_decay=None, learn_logvar=False, train_from_hgf=False,
                 only_finetune_decoder=False, training_opt=None, device_index=0):
        super().__init__()
        ddconfig = dict(ddconfig)
        if not train_from_hgf:                      # autoencoder.py:46-48
            ddconfig["in_channels"] = 1
            ddconfig["out_ch"] = 1
        assert ddconfig["double_z"]
        assert embed_dim is not None
        self.embed_dim = embed_dim
        self.image_key = image_key
        self.learn_logvar = learn_logvar
        enc = _EncoderQ(**ddconfig, embed_dim=embed_dim, with_quant=True, device_index=device_index)
        dec = _DecoderQ(**ddconfig, embed_dim=embed_dim, with_quant=True, device_index=device_index)
        object.__setattr__(self, "_enc", enc)       # not registered as sub-modules: their parameters appear once, below
        object.__setattr__(self, "_dec", dec)
        for half in (enc, dec):
            for nm, p in half.named_parameters():
                mod = self
                parts = nm.split(".")
                for part in parts[:-1]:
                    if part not in mod._modules:
                        mod.add_module(part, _Holder())
                    mod = mod._modules[part]
                mod.register_parameter(parts[-1], p)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        """autoencoder.py:94-104; weights_only=True: a checkpoint is data, nothing in it is executed."""
        sd = torch.load(path, map_location="cpu", weights_only=True)["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    def set_precision(self, precision: str):
        self._enc.set_precision(precision)
        self._dec.set_precision(precision)
        return self

    def encode(self, x):
        """:138-142"""
        return DiagonalGaussianDistribution(self._enc(x))

    def decode(self, z):
        """:144-147"""
        return self._dec(z)

    def forward(self, input, sample_posterior=True, noise=None, seed=None):
        """:149-158.  ``noise`` / ``seed`` (extensions) are handed to posterior.sample()."""
        posterior = self.encode(input)
        z = posterior.sample(noise=noise, seed=seed) if sample_posterior else posterior.mode()
        return self.decode(z), posterior
