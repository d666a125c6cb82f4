#!/usr/bin/env python3
"""Same-process A/B of the dominant convolution kernel's MFMA shape (32x32x16 vs 16x16x32, dsd_set_conv_mfma16) on the large
3x3 layers, interleaved rounds (cdna_hip_programming.md rule 24).  python tools/ab_mfma16.py [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
SHAPES = [(16, 256, 256, 320, 320, 3, 1), (16, 256, 256, 640, 320, 3, 1), (16, 128, 128, 320, 320, 3, 1), (16, 64, 64, 640, 640, 3, 1),
          (16, 128, 128, 640, 320, 3, 1), (16, 64, 64, 1280, 640, 3, 1)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for shp in SHAPES:
    res = {0: [], 1: []}
    for r in range(rounds):
        for on in (0, 1):
            L.dsd_set_conv_mfma16(on)
            ms, fl = C.c_float(), C.c_double()
            _lib.check(L.dsd_bench_conv2d(*shp, 41, 8, C.byref(ms), C.byref(fl)))
            res[on].append((ms.value, fl.value / ms.value / 1e9))
    L.dsd_set_conv_mfma16(0)
    f = lambda v: " ".join(f"{m:.3f}ms/{t:.1f}TF" for m, t in v)
    med = lambda v: sorted(m for m, _ in v)[len(v) // 2]
    print(shp, "| 32x32x16:", f(res[0]), "| 16x16x32:", f(res[1]), f"| ratio of medians (16/32) {med(res[1]) / med(res[0]):.4f}", flush=True)
