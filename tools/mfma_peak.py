#!/usr/bin/env python3
"""Bare-MFMA rates of this device (csrc/peak.hip).  python tools/mfma_peak.py [workgroups_per_cu] [ms_per_launch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
wpc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ms_t = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
NAMES = {0: "32x32x16 registers only, random", 1: "32x32x16 weights from LDS, random", 2: "32x32x16 registers only, zeros",
         3: "32x32x16 weights from LDS, zeros", 4: "16x16x32 registers only, random", 5: "16x16x32 weights from LDS, random",
         6: "16x16x32 registers only, zeros", 7: "16x16x32 weights from LDS, zeros"}
for rnd in range(2):
    for v in (0, 4, 1, 5, 2, 6, 3, 7):
        ms, tf = C.c_float(), C.c_double()
        _lib.check(L.dsd_bench_mfma_peak(v, wpc, ms_t, 5, C.byref(ms), C.byref(tf)))
        print(f"round {rnd} variant {v} ({NAMES[v]:35s}): {ms.value:8.3f} ms/launch  {tf.value:8.1f} TF/s issued "
              f"= {tf.value / 2500:.3f} of 2.5 PF = {tf.value / 2500 * 2.4:.2f} GHz-equivalent", flush=True)
