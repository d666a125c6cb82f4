"""DiT — drop-in for UNet_DS_Diff/DiT_models.py:145-372 (transformer backbone), executed by libdsdiff.so.

Same constructor keywords, the same factory functions (``DiT_XL_2`` ... ``DiT_S_8``, ``DiT_models``), the same
``state_dict`` names ("x_embedder.proj.weight", "blocks.3.attn.qkv.weight", "blocks.3.adaLN_modulation.1.bias",
"final_layer.linear.weight", "pos_embed", ...), the same ``forward(x, t, y=None, cond=None)`` and ``forward_with_cfg``.
The module holds the parameters; patch embedding, the adaLN-Zero blocks (LayerNorm + modulate, multi-head attention on the
flash kernel, tanh-GELU MLP, gated residuals), the final layer and unpatchify run behind the C ABI (DSD_BLOCK_DIT).

Parity status: UNPINNED by the reference — DiT_models.py needs ``timm`` (PatchEmbed / Attention / Mlp), which the build image
does not have, so no fixture could be produced from the reference itself; the native path is checked against
oracle/dit.py, a restatement that writes those three timm modules out from their published definition.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from .._lib import check, dptr, lib, stream_ptr
from ..blocks import _Block


class DiT(_Block):
    _zero_sites = False

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, device_index=0):
        super().__init__()
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels // 3 * 2 if learn_sigma else in_channels      # DiT_models.py:163 (sic)
        self.patch_size = patch_size
        self.num_heads = num_heads
        self.input_size = input_size
        self.num_classes = num_classes
        self._create(_lib.BLOCK_DIT, [input_size, patch_size, in_channels, hidden_size, depth, num_heads,
                                      int(hidden_size * mlp_ratio), num_classes, int(bool(learn_sigma)),
                                      int(class_dropout_prob > 0)], device_index)
        self.initialize_weights(hidden_size, depth)
        half = 128                                                                     # frequency_embedding_size 256
        freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32) / half).contiguous()
        check(lib().dsd_set_timestep_freqs(self._h, C.c_void_p(freqs.data_ptr()), half))

    @torch.no_grad()
    def initialize_weights(self, hidden_size, depth):
        """DiT.initialize_weights (:178-207): xavier-uniform Linears with zero bias, sin-cos pos_embed, N(0, 0.02) label table
        and timestep MLP, zero adaLN modulations and zero output layer."""
        sd = dict(self.named_parameters())
        for nm, p in sd.items():
            if nm.endswith(".weight") and p.dim() == 2 and "embedding_table" not in nm:
                torch.nn.init.xavier_uniform_(p)
            elif nm.endswith(".bias"):
                p.zero_()
        grid = int(sd["pos_embed"].shape[1] ** 0.5)
        sd["pos_embed"].copy_(torch.from_numpy(get_2d_sincos_pos_embed(hidden_size, grid)).float().unsqueeze(0))
        w = sd["x_embedder.proj.weight"]
        torch.nn.init.xavier_uniform_(w.view(w.shape[0], -1))
        if "y_embedder.embedding_table.weight" in sd:
            sd["y_embedder.embedding_table.weight"].normal_(std=0.02)
        sd["t_embedder.mlp.0.weight"].normal_(std=0.02)
        sd["t_embedder.mlp.2.weight"].normal_(std=0.02)
        for i in range(depth):
            sd[f"blocks.{i}.adaLN_modulation.1.weight"].zero_()
        sd["final_layer.adaLN_modulation.1.weight"].zero_()
        sd["final_layer.linear.weight"].zero_()

    def _out_shape(self, x):
        return (x.shape[0], self.out_channels, x.shape[2], x.shape[3])

    @torch.no_grad()
    def forward(self, x, t, y=None, cond=None):
        """:224-243.  x: (N, C, H, W), t: (N,), y: (N,) int labels or None, cond: extra channels concatenated to x."""
        if cond is not None:
            x = torch.cat([x, cond], dim=1)
        if not x.is_cuda:
            raise _lib.DsdError("DiT runs on the MI355X only (no CPU fallback)")
        self.sync_params()
        x = x.float().contiguous()
        N, Cc, H, W = x.shape
        tf = t.to(x.device).float().contiguous()
        assert tf.shape == (N,)
        yl = None if y is None else y.to(x.device).long().contiguous()
        out = torch.empty(self._out_shape(x), device=x.device, dtype=torch.float32)
        check(lib().dsd_block_forward(self._h, dptr(x), N, Cc, H, W, dptr(tf), 1,
                                      None if yl is None else C.c_void_p(yl.data_ptr()), 0 if yl is None else 1, dptr(out),
                                      stream_ptr()))
        return out

    def forward_with_cfg(self, x, t, y, cfg_scale):
        """:245-262 — classifier-free guidance: the batch holds the same samples twice (conditional half first, then the
        null-label half); only its first half is used as input, the guided noise estimate e_u + s (e_c - e_u) is formed on the
        first three output channels and written to both halves, the remaining channels pass through."""
        n = x.shape[0] // 2
        out = self.forward(torch.cat([x[:n], x[:n]]), t, y)
        e_c, e_u = out[:n, :3], out[n:, :3]
        guided = e_u + cfg_scale * (e_c - e_u)
        return torch.cat([guided.repeat(2, 1, 1, 1), out[:, 3:]], dim=1)


# ---- fixed sin-cos position table (:269-312): float64 numpy, the same operations in the same order as the reference so
# that a default-initialised model carries the same buffer
def _sincos_1d(dim, positions):
    """[len(positions), dim]: sin | cos of position x 10000^(-i / (dim / 2)), i = 0 .. dim / 2 - 1."""
    if dim % 2:
        raise ValueError("the embedding width must be even")
    freq = np.arange(dim // 2, dtype=np.float64)
    freq /= dim / 2.
    freq = 1. / 10000 ** freq
    angle = np.einsum('m,d->md', np.asarray(positions).reshape(-1), freq)
    return np.concatenate([np.sin(angle), np.cos(angle)], axis=1)


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    return _sincos_1d(embed_dim, pos)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    """grid[0] / grid[1]: the two coordinate planes; each gets half of the width."""
    if embed_dim % 2:
        raise ValueError("the embedding width must be even")
    return np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    """[grid_size^2 (+ extra_tokens zero rows in front), embed_dim]; the column coordinate fills the first half."""
    axis = np.arange(grid_size, dtype=np.float32)
    cols, rows = np.meshgrid(axis, axis)                       # meshgrid(w, h): plane 0 varies along the width
    table = get_2d_sincos_pos_embed_from_grid(embed_dim, np.stack([cols, rows]).reshape(2, 1, grid_size, grid_size))
    if cls_token and extra_tokens > 0:
        table = np.concatenate([np.zeros([extra_tokens, embed_dim]), table], axis=0)
    return table


# ---- configurations (:319-372)
def DiT_XL_2(**kwargs): return DiT(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kwargs)
def DiT_XL_4(**kwargs): return DiT(depth=28, hidden_size=1152, patch_size=4, num_heads=16, **kwargs)
def DiT_XL_8(**kwargs): return DiT(depth=28, hidden_size=1152, patch_size=8, num_heads=16, **kwargs)
def DiT_L_2(**kwargs): return DiT(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)
def DiT_L_4(**kwargs): return DiT(depth=24, hidden_size=1024, patch_size=4, num_heads=16, **kwargs)
def DiT_L_8(**kwargs): return DiT(depth=24, hidden_size=1024, patch_size=8, num_heads=16, **kwargs)
def DiT_B_2(**kwargs): return DiT(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kwargs)
def DiT_B_4(**kwargs): return DiT(depth=12, hidden_size=768, patch_size=4, num_heads=12, **kwargs)
def DiT_B_8(**kwargs): return DiT(depth=12, hidden_size=768, patch_size=8, num_heads=12, **kwargs)
def DiT_S_2(**kwargs): return DiT(depth=12, hidden_size=384, patch_size=2, num_heads=6, **kwargs)
def DiT_S_4(**kwargs): return DiT(depth=12, hidden_size=384, patch_size=4, num_heads=6, **kwargs)
def DiT_S_8(**kwargs): return DiT(depth=12, hidden_size=384, patch_size=8, num_heads=6, **kwargs)


DiT_models = {f"DiT-{size}/{patch}": globals()[f"DiT_{size}_{patch}"] for size in ("XL", "L", "B", "S") for patch in (2, 4, 8)}
