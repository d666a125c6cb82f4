#!/usr/bin/env python3
"""What-if table of the half-precision GEMM kernel on the DiT-B/8 @512 shapes (one cost removed at a time, interleaved rounds).
    python tools/gemm_whatif.py [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NAMES = {-1: "product", 0: "diag build, nothing removed", 1: "no DMA staging", 2: "fragments read once", 3: "no DMA, fragments once",
         4: "no epilogue", 16: "output stays in L2", 32: "DMA issued, never waited for", 34: "DMA never waited for, fragments read once", 64: "32x32x16 MFMA build", 128: "one loader wave per SIMD", 129: "one loader per SIMD, pieces spread", 7: "MFMA + barrier only", 15: "MFMA only"}
for M, N, K in ((65536, 2304, 768), (65536, 768, 3072), (65536, 3072, 768)):
    fl = 2.0 * M * N * K
    res = {w: [] for w in NAMES}
    for r in range(rounds):
        for w in NAMES:
            ms = C.c_float()
            _lib.check(L.dsd_bench_gemm_half(M, N, K, 0, 0, w, 10, C.byref(ms)))
            res[w].append(ms.value)
    print(f"M={M} N={N} K={K}")
    for w, v in res.items():
        m = sorted(v)[len(v) // 2]
        print(f"  {NAMES[w]:30s} {m * 1e3:8.1f} us  {fl / m / 1e9:7.1f} TF/s", flush=True)
