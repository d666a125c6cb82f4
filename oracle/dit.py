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


def _r(x, half):
    """Round to the 16-bit type once (nearest even) and come back to fp32: what a half-precision operand holds."""
    return x if half is None else x.to(half).float()


def attention_half(sd, p, x16, heads, half):
    """timm Attention.forward with the roundings of the native half-precision mode (DSD_PREC_F16 / DSD_PREC_BF16): 16-bit
    operands, fp32 accumulation, q scaled by hd^-1/2 log2(e) before ITS rounding, base-2 softmax with fp32 statistics and a
    16-bit P.  (Under torch.autocast the reference rounds at the same places but keeps natural logarithms; the native kernel
    also exponentiates against a lazily updated maximum, which rounds P at a different power of two — a test tolerance, not
    an identity.)"""
    B, N, Cc = x16.shape
    hd = Cc // heads
    w, b = _r(sd[p + ".qkv.weight"], half), sd[p + ".qkv.bias"]
    qkv = F.linear(x16, w, b)
    qkv[..., :Cc] *= hd ** -0.5 * 1.4426950408889634
    qkv = _r(qkv, half).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    s = q @ k.transpose(-2, -1)
    pexp = torch.exp2(s - s.amax(dim=-1, keepdim=True))
    o = _r((_r(pexp, half) @ v) / pexp.sum(dim=-1, keepdim=True), half)
    return _r(F.linear(o.transpose(1, 2).reshape(B, N, Cc), _r(sd[p + ".proj.weight"], half), sd[p + ".proj.bias"]), half)


def attention(sd, p, x, heads):                               # timm Attention.forward
    B, N, Cc = x.shape
    hd = Cc // heads
    qkv = _lin(sd, p + ".qkv", x).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    return _lin(sd, p + ".proj", (attn @ v).transpose(1, 2).reshape(B, N, Cc))


def dit_forward(sd, x, t, y=None, *, patch_size, num_heads, out_channels, half=None):
    """half: None (fp32, the oracle proper) | torch.float16 | torch.bfloat16 — the latter two restate the NATIVE
    half-precision mode rounding for rounding (operands of the four Linears of a block and of the two attention products
    rounded once, everything else fp32); parity of that mode with the reference's autocast is unpinned like the rest."""
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
        if half is not None:
            n1 = _r(modulate(F.layer_norm(x, (D,), eps=1e-6), sm, cm), half)
            x = x + gm.unsqueeze(1) * attention_half(sd, b + ".attn", n1, num_heads, half)
            n2 = _r(modulate(F.layer_norm(x, (D,), eps=1e-6), sp, cp), half)
            h = _r(F.linear(n2, _r(sd[b + ".mlp.fc1.weight"], half), sd[b + ".mlp.fc1.bias"]), half)
            g = _r(F.gelu(h, approximate="tanh"), half)
            x = x + gp.unsqueeze(1) * _r(F.linear(g, _r(sd[b + ".mlp.fc2.weight"], half), sd[b + ".mlp.fc2.bias"]), half)
            continue
        x = x + gm.unsqueeze(1) * attention(sd, b + ".attn", modulate(F.layer_norm(x, (D,), eps=1e-6), sm, cm), num_heads)
        h = _lin(sd, b + ".mlp.fc1", modulate(F.layer_norm(x, (D,), eps=1e-6), sp, cp))
        x = x + gp.unsqueeze(1) * _lin(sd, b + ".mlp.fc2", F.gelu(h, approximate="tanh"))
    shift, scale = _lin(sd, "final_layer.adaLN_modulation.1", F.silu(c)).chunk(2, dim=1)            # FinalLayer :138-142
    x = _lin(sd, "final_layer.linear", modulate(F.layer_norm(x, (D,), eps=1e-6), shift, scale))
    hh = int(x.shape[1] ** 0.5)                                                                    # unpatchify :209-222
    x = x.reshape(x.shape[0], hh, hh, patch_size, patch_size, out_channels)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], out_channels, hh * patch_size, hh * patch_size)
