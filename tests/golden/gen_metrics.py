#!/opt/conda/bin/python3.9
"""Golden values for the two image metrics the reference takes from scikit-image (inference/test_metrics.py:7-8: peak_signal_noise_ratio
and structural_similarity(win_size=9)), computed by REAL scikit-image 0.18.3 — present only under /opt/conda/bin/python3.9:

    /opt/conda/bin/python3.9 tests/golden/gen_metrics.py   ->  tests/golden/metrics.npz
"""
import os

import numpy as np
import skimage
from skimage.metrics import peak_signal_noise_ratio, structural_similarity

rng = np.random.default_rng(41)
out = {"skimage_version": np.array(skimage.__version__)}
z, y, x = np.meshgrid(np.linspace(-1, 1, 14), np.linspace(-1, 1, 40), np.linspace(-1, 1, 36), indexing="ij")
true3 = (np.exp(-3 * (x * x + y * y + z * z)) * 900 + 40 * np.sin(7 * x) * np.cos(5 * y)).astype(np.float32)
pred3 = (true3 + rng.normal(0, 25, true3.shape)).astype(np.float32)
true2, pred2 = true3[6].astype(np.float64), pred3[6].astype(np.float64)
for name, t, p in (("v3", true3, pred3), ("s2", true2, pred2)):
    dr = t.max() - t.min()
    out[name + "_true"], out[name + "_pred"] = t, p
    out[name + "_psnr"] = np.float64(peak_signal_noise_ratio(t, p, data_range=dr))
    out[name + "_ssim9"] = np.float64(structural_similarity(t, p, win_size=9, data_range=dr))
    out[name + "_ssim7"] = np.float64(structural_similarity(t, p, data_range=dr))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics.npz"), **out)
print({k: float(v) for k, v in out.items() if v.ndim == 0 and k != "skimage_version"})
