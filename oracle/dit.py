"""CPU oracle of the DiT backbone — TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference's UNet_DS_Diff/DiT_models.py imports timm (PatchEmbed, Attention, Mlp), which is absent
from the build image, so the reference itself could not be run to produce fixtures and holds no tests or golden vectors
for this path.  This file restates DiT.forward (DiT_models.py:224-243) in torch-CPU fp32, with the three timm modules
written out from their published definition (timm 0.9, models/vision_transformer.py / layers/{patch_embed,mlp}.py):
  PatchEmbed   Conv2d(in, D, kernel = stride = p, bias) -> flatten(2).transpose(1, 2)            (norm_layer=None)
  Attention    qkv = Linear(D, 3D, bias); reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4); q * hd^-0.5;
               softmax(q k^T) v; transpose/reshape; proj = Linear(D, D)                            (no qk-norm, no dropout)
  Mlp          fc1 -> act -> fc2                                                                  (drop = 0)
State-dict names are those the reference's module tree produces.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim=256, max_period=10000):        # TimestepEmbedder.timestep_embedding :41-61
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def modulate(x, shift, scale):                                # :19-20
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


def attention(sd, p, x, heads):                               # timm Attention.forward
    B, N, Cc = x.shape
    hd = Cc // heads
    qkv = _lin(sd, p + ".qkv", x).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    return _lin(sd, p + ".proj", (attn @ v).transpose(1, 2).reshape(B, N, Cc))


def dit_forward(sd, x, t, y=None, *, patch_size, num_heads, out_channels):
    D = sd["pos_embed"].shape[-1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    x = F.conv2d(x, sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], stride=patch_size)   # PatchEmbed
    x = x.flatten(2).transpose(1, 2) + sd["pos_embed"]                                             # :232
    c = _lin(sd, "t_embedder.mlp.2", F.silu(_lin(sd, "t_embedder.mlp.0", timestep_embedding(t))))  # :63-66
    if y is not None:
        c = c + sd["y_embedder.embedding_table.weight"][y]                                         # :236-237 (eval: no dropout)
    for i in range(depth):                                                                         # DiTBlock.forward :118-122
        b = f"blocks.{i}"
        sm, cm, gm, sp, cp, gp = _lin(sd, b + ".adaLN_modulation.1", F.silu(c)).chunk(6, dim=1)
        x = x + gm.unsqueeze(1) * attention(sd, b + ".attn", modulate(F.layer_norm(x, (D,), eps=1e-6), sm, cm), num_heads)
        h = _lin(sd, b + ".mlp.fc1", modulate(F.layer_norm(x, (D,), eps=1e-6), sp, cp))
        x = x + gp.unsqueeze(1) * _lin(sd, b + ".mlp.fc2", F.gelu(h, approximate="tanh"))
    shift, scale = _lin(sd, "final_layer.adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)            # FinalLayer :138-142
    x = _lin(sd, "final_layer.linear", modulate(F.layer_norm(x, (D,), eps=1e-6), shift, scale))
    hh = int(x.shape[1] ** 0.5)                                                                    # unpatchify :209-222
    x = x.reshape(x.shape[0], hh, hh, patch_size, patch_size, out_channels)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], out_channels, hh * patch_size, hh * patch_size)
