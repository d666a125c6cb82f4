#!/usr/bin/env python3
"""What-if table of the half-precision attention kernel at DiT-B/8 @512 (16 x 4096 tokens x 12 heads of 64): one cost removed at
a time, interleaved rounds.    python tools/attn_whatif.py [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NAMES = {-1: "product", 0: "diag build, nothing removed", 1: "no softmax arithmetic", 2: "no K/V staging in the loop", 3: "no softmax, no staging",
         4: "no P V product", 8: "no Q K^T product", 13: "softmax arithmetic + staging only",
         100: "LDS-DMA kernel", 101: "LDS-DMA, no softmax arithmetic", 102: "LDS-DMA, no staging in the loop", 103: "LDS-DMA, neither",
         200: "LDS-DMA, 8 waves (256 queries)", 202: "LDS-DMA, 8 waves, no staging in the loop"}
N, T, Cc, H = 16, 4096, 768, 12
fl = 4.0 * N * H * T * T * (Cc // H)
res = {w: [] for w in NAMES}
for r in range(rounds):
    for w in NAMES:
        ms = C.c_float()
        _lib.check(L.dsd_bench_attention_half(N, T, Cc, H, 0, w, 5, C.byref(ms)))
        res[w].append(ms.value)
for w, v in res.items():
    m = sorted(v)[len(v) // 2]
    print(f"  {NAMES[w]:36s} {m * 1e3:8.1f} us  {fl / m / 1e9:7.1f} TF/s (of the full work)", flush=True)
