"""Kernel-level entry points (dsd_op_*) on torch CUDA tensors — used by tests and micro-benchmarks.

Activations cross this boundary as NHWC; helpers convert from the reference's NCHW.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import lib, check, dptr, stream_ptr


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16x6": 2, "f16x3": 3}   # the convolution modes (f16 / bf16 are DiT-handle modes)


STRUCTURES = {"auto": 0, "adirect": 16, "staged": 32, "adirect256": 64, "winograd": 128}


def conv2d(x_nhwc, w_oihw, bias, stride=1, upsample=False, emb=None, res=None, precision="f32", structure="auto"):
    N, H, W, Cin = x_nhwc.shape
    Cout, _, ks, _ = w_oihw.shape
    IH, IW = (H * 2, W * 2) if upsample else (H, W)
    pad = ks // 2
    OH, OW = (IH + 2 * pad - ks) // stride + 1, (IW + 2 * pad - ks) // stride + 1
    y = torch.empty((N, OH, OW, Cout), device=x_nhwc.device, dtype=torch.float32)
    check(lib().dsd_op_conv2d_prec(dptr(x_nhwc), N, H, W, Cin, dptr(w_oihw.contiguous()), dptr(bias), Cout, ks, stride,
                                   int(upsample), dptr(emb), dptr(res), PRECISIONS[precision] | STRUCTURES[structure],
                                   dptr(y), stream_ptr()))
    return y


def group_norm(x_nhwc, gamma, beta, eps=1e-5, silu=False):
    N, H, W, Cc = x_nhwc.shape
    y = torch.empty_like(x_nhwc)
    check(lib().dsd_op_group_norm(dptr(x_nhwc), N, H * W, Cc, dptr(gamma), dptr(beta), eps, int(silu), dptr(y),
                                  stream_ptr()))
    return y


def gn_silu_conv_out1(x_nhwc, gamma, beta, w_oihw, bias, eps=1e-5):
    """The U-Net's `out` layer in one pass: Conv3x3(SiLU(GroupNorm32(x))) -> [N, H, W] (one output channel)."""
    N, H, W, Cc = x_nhwc.shape
    y = torch.empty((N, H, W), device=x_nhwc.device, dtype=torch.float32)
    check(lib().dsd_op_gn_silu_conv_out1(dptr(x_nhwc), N, H, W, Cc, dptr(gamma), dptr(beta), eps, dptr(w_oihw.contiguous()),
                                         dptr(bias), dptr(y), stream_ptr()))
    return y


def qkv_attention(qkv_ntc, heads, new_order=True, split=False):
    """split: bf16x6 arithmetic (operands as 3 bf16 pieces, 6 products) instead of the fp32 matrix cores."""
    N, T, C3 = qkv_ntc.shape
    a = torch.empty((N, T, C3 // 3), device=qkv_ntc.device, dtype=torch.float32)
    check(lib().dsd_op_qkv_attention(dptr(qkv_ntc), N, T, C3 // 3, heads, int(new_order), int(split), dptr(a), stream_ptr()))
    return a


def gemm_half(x, w, bias=None, dtype="f16", epi="store", gate=None, T=1, y=None):
    """nn.Linear on 16-bit operands (one MFMA per product, fp32 accumulation): y = x w^T + bias.  epi "store" | "gelu" |
    "gated" (y fp32 in/out: y += gate[m // T] * round16(x w^T + bias))."""
    M, K = x.shape
    N = w.shape[0]
    e = {"store": 0, "gelu": 1, "gated": 2}[epi]
    if e == 2:
        assert y is not None and gate is not None
    else:
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    check(lib().dsd_op_gemm_half(dptr(x), dptr(w.contiguous()), dptr(bias), M, N, K, int(dtype == "bf16"), e, dptr(gate), int(T),
                                 dptr(y), stream_ptr()))
    return y


def attention_half(qkv_ntc, heads, dtype="f16", thr=-1.0):
    """timm Attention core on qkv[N,T,3C] (q | k | v): softmax(q k^T d^-1/2) v with 16-bit operands, fp32 statistics."""
    N, T, C3 = qkv_ntc.shape
    a = torch.empty((N, T, C3 // 3), device=qkv_ntc.device, dtype=torch.float32)
    check(lib().dsd_op_attention_half(dptr(qkv_ntc), N, T, C3 // 3, heads, int(dtype == "bf16"), float(thr), dptr(a), stream_ptr()))
    return a


def timestep_embedding(t, dim, freqs=None):
    """freqs: optional [dim//2] fp32 CUDA tensor, the caller's own exp(-ln(1e4) k / half) table (the shim path)."""
    is_float = t.dtype.is_floating_point
    t = t.float().contiguous() if is_float else t.long().contiguous()
    y = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(lib().dsd_op_timestep_embedding(C.c_void_p(t.data_ptr()), int(is_float), t.shape[0], dim, dptr(freqs), dptr(y),
                                          stream_ptr()))
    return y


def linear(x, w, bias=None, act_in=0):
    N, K = x.shape
    y = torch.empty((N, w.shape[0]), device=x.device, dtype=torch.float32)
    check(lib().dsd_op_linear(dptr(x), N, K, dptr(w.contiguous()), dptr(bias), w.shape[0], act_in, dptr(y), stream_ptr()))
    return y


def philox_normal(n, seed, step, device="cuda"):
    y = torch.empty((n,), device=device, dtype=torch.float32)
    check(lib().dsd_op_philox_normal(dptr(y), n, seed, step, stream_ptr()))
    return y
