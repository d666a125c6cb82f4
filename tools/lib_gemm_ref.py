#!/usr/bin/env python3
"""Calibration only (never on a product path): what the vendor GEMM library (torch.matmul -> hipBLASLt / rocBLAS) reaches on
the DiT-B/8 @512 Linear shapes, next to dsd_bench_gemm_half on the same box.   python tools/lib_gemm_ref.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
for dt, bf in ((torch.float16, 0), (torch.bfloat16, 1)):
    for M, N, K in ((65536, 2304, 768), (65536, 768, 3072), (65536, 3072, 768), (65536, 768, 768)):
        x = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt)
        b = torch.randn(N, device="cuda", dtype=dt)
        for _ in range(3):
            y = torch.nn.functional.linear(x, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            y = torch.nn.functional.linear(x, w, b)
        e1.record()
        torch.cuda.synchronize()
        lib_ms = e0.elapsed_time(e1) / 10
        ms = C.c_float()
        _lib.check(L.dsd_bench_gemm_half(M, N, K, bf, 0, -1, 10, C.byref(ms)))
        fl = 2.0 * M * N * K
        print(f"{str(dt)[6:]:9s} M={M} N={N} K={K}: library {lib_ms * 1e3:7.1f} us {fl / lib_ms / 1e9:7.1f} TF/s | gemm16 {ms.value * 1e3:7.1f} us {fl / ms.value / 1e9:7.1f} TF/s", flush=True)
