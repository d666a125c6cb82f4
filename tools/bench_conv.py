#!/usr/bin/env python3
"""Micro-benchmark of conv kernel variants on the hot shapes (GPU box).  python tools/bench_conv.py [variants...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
SHAPES = [  # N, H, W, Cin, Cout, ks, stride
    (16, 256, 256, 320, 320, 3, 1), (16, 256, 256, 640, 320, 3, 1), (16, 128, 128, 320, 320, 3, 1),
    (16, 64, 64, 640, 640, 3, 1), (16, 32, 32, 640, 640, 3, 1), (16, 16, 16, 960, 960, 3, 1), (16, 8, 8, 960, 960, 3, 1),
    (16, 256, 256, 640, 320, 1, 1), (16, 256, 256, 320, 320, 3, 2), (1, 256, 256, 320, 320, 3, 1),
]
if os.environ.get("DSD_SHAPES") == "wino":   # 3x3 stride-1 layers the F(2,3) kernel takes (variants 41 vs 50)
    SHAPES = [(16, 256, 256, 320, 320, 3, 1), (16, 256, 256, 640, 320, 3, 1), (16, 128, 128, 320, 320, 3, 1), (16, 64, 64, 640, 640, 3, 1),
              (16, 32, 32, 640, 640, 3, 1), (16, 16, 16, 960, 960, 3, 1), (16, 64, 64, 1280, 640, 3, 1), (1, 256, 256, 320, 320, 3, 1),
              (1, 128, 128, 320, 320, 3, 1), (1, 64, 64, 640, 640, 3, 1)]
if os.environ.get("DSD_SHAPES") == "wino2":
    SHAPES = [(16, 256, 256, 320, 320, 3, 1), (16, 64, 64, 640, 640, 3, 1), (16, 128, 128, 256, 256, 3, 1)]
if os.environ.get("DSD_SHAPES") == "small":
    SHAPES = [(16, 16, 16, 960, 960, 3, 1), (16, 8, 8, 960, 960, 3, 1), (16, 8, 8, 1920, 960, 3, 1), (16, 16, 16, 960, 2880, 1, 1),
              (16, 8, 8, 960, 480, 1, 1), (16, 8, 8, 2880, 960, 1, 1), (16, 16, 16, 1920, 960, 1, 1), (16, 32, 32, 640, 1920, 1, 1)]
if os.environ.get("DSD_SHAPES") == "b1":   # batch-1 layer shapes (few output tiles)
    SHAPES = [(1, 256, 256, 320, 320, 3, 1), (1, 128, 128, 320, 320, 3, 1), (1, 128, 128, 640, 320, 3, 1), (1, 64, 64, 640, 640, 3, 1),
              (1, 64, 64, 1280, 640, 3, 1), (1, 32, 32, 640, 640, 3, 1), (1, 32, 32, 1280, 640, 3, 1), (1, 16, 16, 960, 960, 3, 1),
              (1, 8, 8, 960, 960, 3, 1), (2, 64, 64, 640, 640, 3, 1), (4, 32, 32, 640, 640, 3, 1)]
variants = [int(v) for v in sys.argv[1:]] or [0, 1]
for shp in SHAPES:
    row = []
    for v in variants:
        ms, fl = C.c_float(), C.c_double()
        iters = 5 if shp[1] >= 128 else 20
        _lib.check(L.dsd_bench_conv2d(*shp, v, iters, C.byref(ms), C.byref(fl)))
        row.append(f"v{v}: {ms.value:8.3f} ms {fl.value / ms.value / 1e9:7.1f} TF/s")
    print(shp, " | ".join(row), flush=True)
