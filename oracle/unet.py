"""Oracle (test infrastructure): CPU restatement of the DSUnetModel forward.

Functional, state-dict driven, plain torch-CPU fp32 ops.  Follows
  UNet_DS_Diff/model.py:171-756            (DSUnetModel ctor + forward)
  UNet_DS_Diff/model.py:152-168            (FeatureDisentangle)
  ldm/modules/diffusionmodules/openaimodel.py:93-121,138-164,167-284,426-473,496-555
  ldm/modules/diffusionmodules/util.py:161-181,209-226
  Disc_diff/guided_diffusion/unet.py:82-109 (SE_Attention)
of the reference.  Parameter names are the reference's state_dict names.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """ctor kwargs of DSUnetModel (UNet_DS_Diff/model.py:172-202) that shape the graph."""
    in_channels: int = 1
    model_channels: int = 320
    out_channels: int = 1
    num_res_blocks: object = 2
    attention_resolutions: Sequence[int] = (32, 16, 8)
    channel_mult: Sequence[int] = (1, 1, 2, 2, 3, 3)
    num_heads: int = -1
    num_head_channels: int = -1
    num_heads_upsample: int = -1
    use_scale_shift_norm: bool = False
    resblock_updown: bool = False
    use_new_attention_order: bool = False
    legacy: bool = True
    conv_resample: bool = True
    # UNetModel(use_spatial_transformer=True) only (ldm/modules/diffusionmodules/openaimodel.py:623-631,761-765)
    use_spatial_transformer: bool = False
    transformer_depth: int = 1
    context_dim: object = None
    use_linear_in_transformer: bool = False
    # ignored by the graph: image_size, dropout(=0), use_checkpoint, adm_in_channels, ...

    @staticmethod
    def from_params(params: dict) -> "UNetConfig":
        keys = UNetConfig.__dataclass_fields__.keys()
        return UNetConfig(**{k: v for k, v in params.items() if k in keys})


# --------------------------------------------------------------------------- spec
def _res(cin, cout, up=False, down=False):
    return {"kind": "res", "cin": cin, "cout": cout, "up": up, "down": down}


def build_spec(cfg: UNetConfig) -> dict:
    """Module tree, mirroring the ctor loops of UNet_DS_Diff/model.py:282-515."""
    mc = cfg.model_channels
    nrb = cfg.num_res_blocks
    if isinstance(nrb, int):
        nrb = len(cfg.channel_mult) * [nrb]
    num_heads = cfg.num_heads
    num_heads_upsample = cfg.num_heads_upsample
    if num_heads_upsample == -1:
        num_heads_upsample = num_heads
    nhc = cfg.num_head_channels

    def attn(ch, heads_arg):
        # model.py:307-315,323-328 + AttentionBlock.__init__ openaimodel.py:443-449
        nonlocal num_heads
        if nhc == -1:
            dim_head = ch // num_heads
        else:
            num_heads = ch // nhc
            dim_head = nhc
        if cfg.legacy:
            dim_head = ch // num_heads if cfg.use_spatial_transformer else nhc   # openaimodel.py:745-747
        if cfg.use_spatial_transformer:      # SpatialTransformer(ch, num_heads, dim_head, ...) — num_heads in the decoder too
            return {"kind": "st", "ch": ch, "heads": num_heads, "dim_head": dim_head}
        h_arg = num_heads if heads_arg is None else heads_arg
        heads = h_arg if dim_head == -1 else ch // dim_head
        return {"kind": "attn", "ch": ch, "heads": heads}

    input_blocks = [[{"kind": "conv", "cin": cfg.in_channels, "cout": mc}]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(nrb[level]):
            layers = [_res(ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(attn(ch, None))
            input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            if cfg.resblock_updown:
                input_blocks.append([_res(ch, ch, down=True)])
            else:
                input_blocks.append([{"kind": "down", "ch": ch}])
            chans.append(ch)
            ds *= 2
    middle = [_res(ch, ch), attn(ch, None), _res(ch, ch)]
    output_blocks = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(nrb[level] + 1):
            ich = chans.pop()
            layers = [_res(ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(attn(ch, num_heads_upsample))
            if level and i == nrb[level]:
                if cfg.resblock_updown:
                    layers.append(_res(ch, ch, up=True))
                else:
                    layers.append({"kind": "up", "ch": ch})
                ds //= 2
            output_blocks.append(layers)
    conv_ch = int(cfg.channel_mult[0] * mc) * cfg.channel_mult[-1]
    return {"input_blocks": input_blocks, "middle": middle, "output_blocks": output_blocks,
            "conv_ch": conv_ch, "half": int(conv_ch / 2), "final_ch": ch,
            "time_embed_dim": mc * 4}


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """name -> (shape, kind); kind in conv_w/conv_b/lin_w/lin_b/gn_w/gn_b. Names = reference state_dict."""
    spec = build_spec(cfg)
    ted = spec["time_embed_dim"]
    out: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def lin(name, cin, cout, bias=True):
        out[name + ".weight"] = ((cout, cin), "lin_w")
        if bias:
            out[name + ".bias"] = ((cout,), "lin_b")

    def conv(name, cin, cout, k):
        out[name + ".weight"] = ((cout, cin, k, k), "conv_w")
        out[name + ".bias"] = ((cout,), "conv_b")

    def conv1d(name, cin, cout):
        out[name + ".weight"] = ((cout, cin, 1), "conv_w")
        out[name + ".bias"] = ((cout,), "conv_b")

    def gn(name, c):
        out[name + ".weight"] = ((c,), "gn_w")
        out[name + ".bias"] = ((c,), "gn_b")

    def layer(prefix, L):
        if L["kind"] == "conv":
            conv(prefix, L["cin"], L["cout"], 3)
        elif L["kind"] == "res":
            gn(prefix + ".in_layers.0", L["cin"])
            conv(prefix + ".in_layers.2", L["cin"], L["cout"], 3)
            lin(prefix + ".emb_layers.1", ted, 2 * L["cout"] if cfg.use_scale_shift_norm else L["cout"])
            gn(prefix + ".out_layers.0", L["cout"])
            conv(prefix + ".out_layers.3", L["cout"], L["cout"], 3)
            if L["cin"] != L["cout"]:
                conv(prefix + ".skip_connection", L["cin"], L["cout"], 1)
        elif L["kind"] == "attn":
            gn(prefix + ".norm", L["ch"])
            conv1d(prefix + ".qkv", L["ch"], 3 * L["ch"])
            conv1d(prefix + ".proj_out", L["ch"], L["ch"])
        elif L["kind"] == "down":
            conv(prefix + ".op", L["ch"], L["ch"], 3)
        elif L["kind"] == "up":
            conv(prefix + ".conv", L["ch"], L["ch"], 3)

    lin("time_embed.0", cfg.model_channels, ted)
    lin("time_embed.2", ted, ted)
    for bi, layers in enumerate(spec["input_blocks"]):
        for li, L in enumerate(layers):
            layer(f"input_blocks.{bi}.{li}", L)
    for li, L in enumerate(spec["middle"]):
        layer(f"middle_block.{li}", L)
    for bi, layers in enumerate(spec["output_blocks"]):
        for li, L in enumerate(layers):
            layer(f"output_blocks.{bi}.{li}", L)
    gn("out.0", spec["final_ch"])
    conv("out.2", cfg.model_channels, cfg.out_channels, 3)
    for s in ("_a", "_al", "_l"):
        for bi, layers in enumerate(spec["input_blocks"]):
            for li, L in enumerate(layers):
                layer(f"input_blocks{s}.{bi}.{li}", L)
    cc, half = spec["conv_ch"], spec["half"]
    for nm in ("conv_style", "conv_content", "conv_anatomy", "conv_lesion"):
        gn(nm + ".conv_1.0", cc)
        conv(nm + ".conv_1.2", cc, cc, 3)
        gn(nm + ".conv_2.0", cc)
        conv(nm + ".conv_2.2", cc, half, 1)
    for nm in ("style_proj", "share_content_proj", "anatomy_proj", "lesion_proj"):
        lin(nm + ".0.se.0", half, half // 8, bias=False)
        lin(nm + ".0.se.2", half // 8, half, bias=False)
        conv(nm + ".1", half, half, 3)
    conv("all_proj.1", half * 6, cc, 1)
    return out


def make_state_dict(cfg: UNetConfig, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic random-init weights for the whole model (see oracle/synth.py::synth_params)."""
    from .synth import synth_params
    return synth_params([(n, s) for n, (s, _) in param_shapes(cfg).items()], seed)


# --------------------------------------------------------------------------- ops
def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """ldm/modules/diffusionmodules/util.py:161-181 (repeat_only=False)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)   # (as util.py:174-176: the host's table, moved)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x, w, b, eps=1e-5):
    """GroupNorm32(32, C) util.py:209-226 (fp32 compute)."""
    return F.group_norm(x.float(), 32, w, b, eps)


def qkv_attention(qkv: torch.Tensor, n_heads: int, new_order: bool) -> torch.Tensor:
    """QKVAttention (openaimodel.py:537-555) / QKVAttentionLegacy (:505-521)."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = (q * scale).reshape(bs * n_heads, ch, length)
        k = (k * scale).reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
        q, k = q * scale, k * scale
    w = torch.einsum("bct,bcs->bts", q, k)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


class _Net:
    def __init__(self, cfg: UNetConfig, sd: Dict[str, torch.Tensor]):
        self.cfg, self.sd, self.spec = cfg, sd, build_spec(cfg)

    def p(self, name):
        return self.sd[name]

    def conv(self, name, x, stride=1, pad=1):
        return F.conv2d(x, self.p(name + ".weight"), self.p(name + ".bias"), stride=stride, padding=pad)

    def gn(self, name, x):
        return group_norm32(x, self.p(name + ".weight"), self.p(name + ".bias"))

    def res(self, prefix, L, x, emb):
        """ResBlock._forward openaimodel.py:264-284."""
        h = F.silu(self.gn(prefix + ".in_layers.0", x))
        if L["up"]:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif L["down"]:
            h = F.avg_pool2d(h, 2, 2)
            x = F.avg_pool2d(x, 2, 2)
        h = self.conv(prefix + ".in_layers.2", h)
        e = F.linear(F.silu(emb), self.p(prefix + ".emb_layers.1.weight"), self.p(prefix + ".emb_layers.1.bias"))
        e = e[..., None, None]
        if self.cfg.use_scale_shift_norm:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = self.gn(prefix + ".out_layers.0", h) * (1 + scale) + shift
            h = self.conv(prefix + ".out_layers.3", F.silu(h))
        else:
            h = h + e
            h = self.conv(prefix + ".out_layers.3", F.silu(self.gn(prefix + ".out_layers.0", h)))
        if L["cin"] != L["cout"]:
            x = self.conv(prefix + ".skip_connection", x, pad=0)
        return x + h

    def attn(self, prefix, L, x):
        """AttentionBlock._forward openaimodel.py:467-473."""
        b, c, hh, ww = x.shape
        xf = x.reshape(b, c, -1)
        n = self.gn(prefix + ".norm", xf)
        qkv = F.conv1d(n, self.p(prefix + ".qkv.weight"), self.p(prefix + ".qkv.bias"))
        a = qkv_attention(qkv, L["heads"], self.cfg.use_new_attention_order)
        a = F.conv1d(a, self.p(prefix + ".proj_out.weight"), self.p(prefix + ".proj_out.bias"))
        return (xf + a).reshape(b, c, hh, ww)

    def block(self, prefix, layers, x, emb, context=None):
        """TimestepEmbedSequential.forward openaimodel.py:80-90."""
        for li, L in enumerate(layers):
            nm = f"{prefix}.{li}"
            k = L["kind"]
            if k == "conv":
                x = self.conv(nm, x)
            elif k == "res":
                x = self.res(nm, L, x, emb)
            elif k == "attn":
                x = self.attn(nm, L, x)
            elif k == "st":
                from .xattn import spatial_transformer
                x = spatial_transformer(self.sd, nm, x, [context] * self.cfg.transformer_depth, L["heads"], self.cfg.transformer_depth,
                                        self.cfg.use_linear_in_transformer)
            elif k == "down":
                x = self.conv(nm + ".op", x, stride=2)            # Downsample openaimodel.py:162-164
            elif k == "up":
                x = F.interpolate(x, scale_factor=2, mode="nearest")  # Upsample :111-121
                x = self.conv(nm + ".conv", x)
        return x

    def disentangle(self, nm, x):
        """FeatureDisentangle.forward model.py:165-168."""
        o = self.conv(nm + ".conv_1.2", F.silu(self.gn(nm + ".conv_1.0", x))) + x
        return self.conv(nm + ".conv_2.2", F.silu(self.gn(nm + ".conv_2.0", o)), pad=0)

    def se_proj(self, nm, x):
        """SE_Attention (Disc_diff/guided_diffusion/unet.py:105-109) + Conv3x3 (model.py:566-591)."""
        b, c = x.shape[:2]
        y = x.mean(dim=(2, 3))
        y = torch.sigmoid(F.linear(F.relu(F.linear(y, self.p(nm + ".0.se.0.weight"))), self.p(nm + ".0.se.2.weight")))
        return self.conv(nm + ".1", x * y.view(b, c, 1, 1))

    def forward(self, x, timesteps):
        """DSUnetModel.forward model.py:629-756."""
        cfg = self.cfg
        t_emb = timestep_embedding(timesteps, cfg.model_channels)
        emb = F.linear(t_emb, self.p("time_embed.0.weight"), self.p("time_embed.0.bias"))
        emb = F.linear(F.silu(emb), self.p("time_embed.2.weight"), self.p("time_embed.2.bias"))
        if x.shape[1] == 2:
            inp_n, inp_a = x[:, 0:1], x[:, 1:2]
            inp_al = torch.zeros_like(inp_n)
            inp_l = torch.zeros_like(inp_n)
        else:
            inp_a, inp_al, inp_l, inp_n = x[:, 1:2], x[:, 2:3], x[:, 3:4], x[:, 0:1]
        streams = {"": inp_n.float(), "_a": inp_a.float(), "_al": inp_al.float(), "_l": inp_l.float()}
        hs = {k: [] for k in streams}
        hfin = {}
        for s, h in streams.items():
            for bi, layers in enumerate(self.spec["input_blocks"]):
                h = self.block(f"input_blocks{s}.{bi}", layers, h, emb)
                hs[s].append(h)
            hfin[s] = h
        h_n = self.block("middle_block", self.spec["middle"], hfin[""], emb)
        h_a, h_al, h_l = hfin["_a"], hfin["_al"], hfin["_l"]
        d = self.disentangle
        h_n_style, h_n_content = d("conv_style", h_n), d("conv_content", h_n)
        style = [d("conv_style", h_a), d("conv_style", h_al), d("conv_style", h_l)]
        content = [d("conv_content", h_a), d("conv_content", h_al), d("conv_content", h_l)]
        anatomy = [d("conv_anatomy", h_a), d("conv_anatomy", h_al)]
        lesion = [d("conv_lesion", h_al), d("conv_lesion", h_l)]
        h_style = self.se_proj("style_proj", torch.mean(torch.stack(style), dim=0))
        h_share = self.se_proj("share_content_proj", torch.mean(torch.stack(content), dim=0))
        h_anat = self.se_proj("anatomy_proj", torch.mean(torch.stack(anatomy), dim=0))
        h_les = self.se_proj("lesion_proj", torch.mean(torch.stack(lesion), dim=0))
        h = torch.cat([h_n, h_share, h_style, h_anat, h_les], dim=1)
        h = self.conv("all_proj.1", F.silu(h), pad=0)
        for bi, layers in enumerate(self.spec["output_blocks"]):
            skip = (hs[""].pop() + hs["_a"].pop() + hs["_al"].pop() + hs["_l"].pop()) / 4
            h = torch.cat([h, skip], dim=1)
            h = self.block(f"output_blocks.{bi}", layers, h, emb)
        out = self.conv("out.2", F.silu(self.gn("out.0", h.float())))
        feats = {"style": style, "content": content, "anatomy": anatomy, "lesion": lesion,
                 "n_style_content": [h_style, h_n_style, h_share, h_n_content]}
        return out, feats


@torch.no_grad()
def unet_forward(cfg: UNetConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, timesteps: torch.Tensor):
    """Returns (out [B,out_ch,H,W], feature dict) like DSUnetModel.forward (model.py:751-756)."""
    return _Net(cfg, sd).forward(x, timesteps)


@torch.no_grad()
def plain_unet_forward(cfg: UNetConfig, sd: Dict[str, torch.Tensor], x: torch.Tensor, timesteps: torch.Tensor, context=None):
    """UNetModel.forward (ldm/modules/diffusionmodules/openaimodel.py:926-958): the single-stream denoiser of the latent
    path (no class embedding); with cfg.use_spatial_transformer every attention slot is a SpatialTransformer on `context`."""
    net = _Net(cfg, sd)
    t_emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(t_emb, net.p("time_embed.0.weight"), net.p("time_embed.0.bias"))
    emb = F.linear(F.silu(emb), net.p("time_embed.2.weight"), net.p("time_embed.2.bias"))
    h, hs = x.float(), []
    for bi, layers in enumerate(net.spec["input_blocks"]):
        h = net.block(f"input_blocks.{bi}", layers, h, emb, context)
        hs.append(h)
    h = net.block("middle_block", net.spec["middle"], h, emb, context)
    for bi, layers in enumerate(net.spec["output_blocks"]):
        h = torch.cat([h, hs.pop()], dim=1)
        h = net.block(f"output_blocks.{bi}", layers, h, emb, context)
    return net.conv("out.2", F.silu(net.gn("out.0", h)))


def block_forward(cfg: UNetConfig, sd, kind: str, prefix: str, L: dict, x, emb=None):
    """Single block entry for op-level tests."""
    net = _Net(cfg, sd)
    with torch.no_grad():
        if kind == "res":
            return net.res(prefix, L, x, emb)
        if kind == "attn":
            return net.attn(prefix, L, x)
        if kind == "disentangle":
            return net.disentangle(prefix, x)
        if kind == "se_proj":
            return net.se_proj(prefix, x)
    raise ValueError(kind)
