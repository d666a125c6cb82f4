#!/usr/bin/env python3
"""Calibration only (never on a product path): the vendor convolution library (torch.nn.functional.conv2d -> MIOpen, fp32) on
the headline's dominant layer shapes, next to dsd_bench_conv2d in its fp32-grade modes.   python tools/lib_conv_ref.py"""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
torch.backends.cudnn.benchmark = True
for N, H, Cin, Cout in ((16, 256, 320, 320), (16, 128, 320, 320), (16, 64, 640, 640)):   # (MIOpen searches ~2 min per shape)
    fl = 2.0 * N * H * H * Cin * Cout * 9
    x = torch.randn(N, Cin, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    b = torch.randn(Cout, device="cuda")
    out = []
    for name, fmt in (("NCHW", torch.contiguous_format), ("NHWC", torch.channels_last)):
        xx, ww = x.contiguous(memory_format=fmt), w.contiguous(memory_format=fmt)
        try:
            for _ in range(3):
                y = F.conv2d(xx, ww, b, padding=1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                y = F.conv2d(xx, ww, b, padding=1)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out.append(f"library fp32 {name} {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF/s")
        except Exception as e:
            out.append(f"library fp32 {name} failed ({type(e).__name__})")
        del xx, ww
    for label, variant in (("bf16x6", 11), ("f32 MFMA", 0)):
        ms, flo = C.c_float(), C.c_double()
        _lib.check(L.dsd_bench_conv2d(N, H, H, Cin, Cout, 3, 1, variant, 5, C.byref(ms), C.byref(flo)))
        out.append(f"dsd {label} {ms.value:7.3f} ms {fl / ms.value / 1e9:6.1f} TF/s")
    print(f"{N}x{H}x{H} {Cin}->{Cout}: " + " | ".join(out), flush=True)
