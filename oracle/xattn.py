"""Oracle (test infrastructure): CPU restatement of the cross-attention variant's blocks.

Follows ldm/modules/attention.py: GEGLU/FeedForward :47-76, Normalize :88-89 (GroupNorm eps 1e-6),
CrossAttention :145-194, BasicTransformerBlock :302-331, SpatialTransformer :366-428.
The model-level wiring in UNet_DS_Diff/the_best_model_backup_crossatten.py is broken in the
reference, so these are pinned at block level only (tests/golden/xattn.npz).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def cross_attention(sd, p, x, context=None, heads=8):
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, n, inner = q.shape
    d = inner // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * (d ** -0.5)
    sim = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", sim, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def feed_forward_geglu(sd, p, x):
    h = F.linear(x, sd[p + ".net.0.proj.weight"], sd[p + ".net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), sd[p + ".net.2.weight"], sd[p + ".net.2.bias"])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def basic_transformer_block(sd, p, x, context=None, heads=8, disable_self_attn=False):
    x = cross_attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x),
                        context if disable_self_attn else None, heads) + x
    x = cross_attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), context, heads) + x
    x = feed_forward_geglu(sd, p + ".ff", layer_norm(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd, p, x, contexts, heads, depth, use_linear=False):
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    if not use_linear:
        x = F.conv2d(x, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
    if use_linear:
        x = F.linear(x, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    for i in range(depth):
        x = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", x, contexts[i], heads)
    if use_linear:
        x = F.linear(x, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = F.conv2d(x, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return x + x_in
