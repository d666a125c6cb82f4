#!/usr/bin/env python3
"""Calibration only (never on a product path): torch.nn.functional.scaled_dot_product_attention (the ROCm build's flash /
memory-efficient kernels) on the DiT-B/8 @512 attention shape, next to dsd_bench_attention_half.   python tools/lib_attn_ref.py"""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
N, H, T, d = 16, 12, 4096, 64
fl = 4.0 * N * H * T * T * d
for dt, bf in ((torch.float16, 0), (torch.bfloat16, 1)):
    q, k, v = (torch.randn(N, H, T, d, device="cuda", dtype=dt) for _ in range(3))
    for name, ctx in (("default", None), ("flash", "FLASH_ATTENTION"), ("efficient", "EFFICIENT_ATTENTION")):
        try:
            from torch.nn.attention import sdpa_kernel, SDPBackend
            cm = sdpa_kernel(getattr(SDPBackend, ctx)) if ctx else None
            def run():
                if cm is None:
                    return F.scaled_dot_product_attention(q, k, v)
                with sdpa_kernel(getattr(SDPBackend, ctx)):
                    return F.scaled_dot_product_attention(q, k, v)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{str(dt)[6:]:9s} sdpa[{name}]: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)
        except Exception as e:  # a backend this build does not have
            print(f"{str(dt)[6:]:9s} sdpa[{name}]: unavailable ({type(e).__name__}: {str(e)[:80]})", flush=True)
    ms = C.c_float()
    _lib.check(L.dsd_bench_attention_half(N, T, H * d, H, bf, -1, 10, C.byref(ms)))
    print(f"{str(dt)[6:]:9s} attention16:   {ms.value * 1e3:8.1f} us {fl / ms.value / 1e9:7.1f} TF/s", flush=True)
