#!/opt/conda/bin/python3.9
"""Fixtures for tests/test_h5lite.py, written by REAL h5py (3.3.0 / libhdf5 1.10.6: the only interpreter of this image that
has it is /opt/conda/bin/python3.9) the way the reference writes its slice files:

    /opt/conda/bin/python3.9 tests/golden/gen_h5.py

  layer_3.h5         preprocess/to_h5.py:40-50 verbatim in effect: ``with h5py.File(p, 'w') as f: f[key] = array`` for
                     F_Data1 / F_Data2 / S_Data1 / S_Data2 (contiguous datasets, default library version bounds)
  brats_slice.h5     one key per modality (t1, t2, flair, t1ce, seg; inference_2d_BraTs.py's loader keys), created with
                     ``compression='gzip', shuffle=True`` (chunked + filter pipeline), int16 / uint8 / float64 / big-endian
  latest.h5          the same arrays as layer_3.h5 through ``libver='latest'`` (superblock v3, object headers v2, link messages)
  expected.npz       the arrays themselves (numpy), what ``h5py.File(p)[key][()]`` returns
"""
import os

import h5py
import numpy as np

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5")
os.makedirs(here, exist_ok=True)
rng = np.random.default_rng(20261004)
exp = {}

layer = {"F_Data1": rng.standard_normal((24, 20)).astype(np.float32),
         "F_Data2": (rng.standard_normal((24, 20)) * 300).astype(np.float32),
         "S_Data1": rng.integers(-2000, 4000, (24, 20)).astype(np.int16),
         "S_Data2": rng.standard_normal((24, 20))}
with h5py.File(os.path.join(here, "layer_3.h5"), "w") as f:
    for k, v in layer.items():
        f[k] = v
exp.update({"layer_3/" + k: v for k, v in layer.items()})

with h5py.File(os.path.join(here, "latest.h5"), "w", libver="latest") as f:
    for k, v in layer.items():
        f[k] = v
exp.update({"latest/" + k: v for k, v in layer.items()})

brats = {"t1": rng.integers(0, 4096, (30, 26)).astype(np.int16),
         "t2": rng.standard_normal((30, 26)).astype(np.float32),
         "flair": rng.standard_normal((30, 26)),
         "t1ce": rng.standard_normal((30, 26)).astype(">f4"),
         "seg": rng.integers(0, 4, (30, 26)).astype(np.uint8)}
with h5py.File(os.path.join(here, "brats_slice.h5"), "w") as f:
    f.create_dataset("t1", data=brats["t1"], compression="gzip", shuffle=True, chunks=(8, 8))
    f.create_dataset("t2", data=brats["t2"], compression="gzip", compression_opts=9, chunks=(7, 26))
    f.create_dataset("flair", data=brats["flair"], chunks=(16, 16), fletcher32=True)
    f.create_dataset("t1ce", data=brats["t1ce"])
    f.create_dataset("seg", data=brats["seg"], compression="gzip", shuffle=True)
    g = f.create_group("meta")
    g["spacing"] = np.array([1.0, 1.0, 2.5], dtype=np.float32)
exp.update({"brats_slice/" + k: v.astype(v.dtype.newbyteorder("=")) for k, v in brats.items()})
exp["brats_slice/meta/spacing"] = np.array([1.0, 1.0, 2.5], dtype=np.float32)

np.savez(os.path.join(here, "expected.npz"), **exp)
for n in sorted(os.listdir(here)):
    print(n, os.path.getsize(os.path.join(here, n)))
