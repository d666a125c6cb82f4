#!/usr/bin/env python3
"""Where a workgroup of the dominant convolution kernel spends its time (diagnostic build with clock stamps).
python tools/conv_stamps.py [N H W Cin Cout]      default 16 256 256 320 320"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_models_dsdiff_amd import _lib
L = _lib.lib()
_lib.require_gpu(0)
shape = [int(v) for v in sys.argv[1:6]] or [16, 256, 256, 320, 320]
whatifs = [int(v) for v in os.environ.get("DSD_WHATIF", "0").split(",")]
cap = 1 << 16
WHAT = {0: "plain A-direct kernel (the product kernel of the layers without tap reuse)", 2: "no activation loads", 4: "no weight loads / LDS writes",
        8: "no barrier", 16: "no weight fragment reads", 31: "bare MFMA stream of this kernel",
        32: "loads issued in bursts of eight (the schedule before r2; correct results)",
        512: "tap reuse: one filter row in LDS serves the three taps (correct results)",
        256: "activations pre-split in memory: 12 piece loads per tile, no split VALU (VERDICT r1 item 5a, conv side)"}
for wi in whatifs:
  buf = np.zeros((cap, 8), dtype=np.int64)
  n = C.c_int()
  _lib.check(L.dsd_bench_conv2d_stamps(*shape, 200, wi, buf.ctypes.data_as(C.POINTER(C.c_longlong)), cap, C.byref(n)))
  print(f"== what-if {wi}: {WHAT.get(wi, '?')}")
  if True:
    st = buf[: n.value]
    core, real = st[:, 0::2].astype(np.float64), st[:, 1::2].astype(np.float64) * 10.0   # ns (100 MHz counter)
    t0 = real[:, 0].min()
    pro, loop, epi = real[:, 1] - real[:, 0], real[:, 2] - real[:, 1], real[:, 3] - real[:, 2]
    clk = (core[:, 2] - core[:, 1]) / np.maximum(loop, 1.0)                                 # GHz inside the k-loop
    print(f"shape {shape}: {n.value} workgroups, kernel span {(real[:, 3].max() - t0) / 1e3:.1f} us")
    for name, v in (("prologue", pro), ("k-loop", loop), ("epilogue", epi), ("whole workgroup", real[:, 3] - real[:, 0])):
        print(f"  {name:16s} mean {v.mean() / 1e3:8.2f} us   median {np.median(v) / 1e3:8.2f}   p5 {np.percentile(v, 5) / 1e3:8.2f}   p95 {np.percentile(v, 95) / 1e3:8.2f}")
    print(f"  in-loop clock    median {np.median(clk):.3f} GHz (p5 {np.percentile(clk, 5):.3f}, p95 {np.percentile(clk, 95):.3f})")
    kt = shape[3] * 9 // 32
    cyc = (core[:, 2] - core[:, 1]) / kt
    print(f"  cycles per k-tile median {np.median(cyc):.0f} (MFMA-only: 120 x 32 = 3840)")
    # slots: 256 workgroups run at a time; the gap between a workgroup's end and the next start in start order
    order = np.argsort(real[:, 0])
    starts, ends = real[order, 0], np.sort(real[:, 3])
    rounds = n.value // 256
    if rounds > 1:
        gaps = starts[256:] - ends[: len(starts) - 256]
        print(f"  successor start - predecessor end (k-th start vs k-th end, by order): median {np.median(gaps) / 1e3:.2f} us, mean {gaps.mean() / 1e3:.2f}")
    busy = (real[:, 3] - real[:, 0]).sum() / 256 / (real[:, 3].max() - t0)
    print(f"  slot occupancy (sum of workgroup times / 256 / span) {busy:.3f}")
