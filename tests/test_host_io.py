"""Host I/O around the sampling path: slice -> volume bookkeeping, the NIfTI-1 container and the array
metrics.  PARITY UNPINNED by the reference (its metric / trainer modules need ants, lpips, SimpleITK, Lightning and do not
import; it holds no fixtures for them): the checks are the defining formulas, round trips and hand-computed cases."""
import gzip
import struct

import numpy as np
import pytest

from diffusion_models_dsdiff_amd import host_io as H


def test_parse_slice_path_and_volume_assembly():
    assert H.parse_slice_path("/data/images_ts/BraTS_017/t1ce_42.h5") == ("BraTS_017", 42)
    assert H.parse_slice_path("a/b/c_d_7.h5") == ("b", 7)
    va = H.VolumeAssembler()
    rng = np.random.default_rng(0)
    imgs = rng.standard_normal((5, 1, 4, 6)).astype(np.float32)
    paths = [f"/x/idA/s_{i}.h5" for i in (3, 0, 2)] + [f"/x/idB/s_{i}.h5" for i in (1, 0)]
    va.add_paths(paths[:2], imgs[:2])            # batches of ragged size, ids interleaved across batches
    va.add_paths(paths[2:], imgs[2:])
    assert sorted(va.ids()) == ["idA", "idB"]
    template = np.ones((5, 4, 6), dtype=np.float32)
    vol = va.volume("idA", template)
    assert vol.shape == (5, 4, 6)
    assert np.array_equal(vol[3], imgs[0, 0]) and np.array_equal(vol[0], imgs[1, 0]) and np.array_equal(vol[2], imgs[2, 0])
    assert not vol[1].any() and not vol[4].any()                          # never-predicted slices stay zero (zeros_like)
    assert va.volume("idB").shape == (2, 4, 6)


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.uint8])
def test_nifti_round_trip_and_copy_information(tmp_path, ext, dtype):
    rng = np.random.default_rng(1)
    vol = (rng.standard_normal((3, 5, 7)) * 50).astype(dtype)
    p = str(tmp_path / ("t" + ext))
    H.write_nifti(p, vol, spacing=(2.5, 0.9, 0.8))                          # (z, y, x) spacing
    back, hdr = H.read_nifti(p)
    assert back.dtype == np.dtype(dtype) and np.array_equal(back, vol)
    raw = (gzip.open(p, "rb") if ext.endswith(".gz") else open(p, "rb")).read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack("<8h", raw[40:56])[:4] == (3, 7, 5, 3)              # dim: x, y, z (x fastest in the file)
    assert np.allclose(struct.unpack("<8f", raw[76:108])[1:4], (0.8, 0.9, 2.5))
    # prediction written with the template's geometry ("CopyInformation"), own dtype
    pred = rng.standard_normal((3, 5, 7)).astype(np.float32)
    q = str(tmp_path / ("p" + ext))
    H.write_nifti(q, pred, template_header=hdr)
    back2, hdr2 = H.read_nifti(q)
    assert np.array_equal(back2, pred)
    assert hdr2[76:108] == hdr[76:108] and hdr2[252:344] == hdr[252:344]     # pixdim and qform / sform carried over


def test_nifti_rejects_garbage(tmp_path):
    p = tmp_path / "bad.nii"
    p.write_bytes(b"\0" * 400)
    with pytest.raises(ValueError):
        H.read_nifti(str(p))


def test_metrics_against_their_definitions():
    rng = np.random.default_rng(2)
    t = rng.uniform(0, 1, (6, 8, 9))
    p = t + 0.05 * rng.standard_normal(t.shape)
    m = np.zeros_like(t, dtype=np.uint8)
    m[1:5, 2:7, 3:8] = 1
    for mask in (None, m):
        sel = np.ones_like(t, bool) if mask is None else mask.astype(bool)
        tt, pp = t[sel], p[sel]
        assert np.isclose(H.nrmse(t, p, mask), np.sqrt(np.mean((tt - pp) ** 2)) / (tt.max() - tt.min()))
        s = lambda a: np.clip((a - a.mean()) / (a.std() / 400.) + 2048., 1e-10, 4095)
        ts, ps = s(tt), s(pp)
        assert np.isclose(H.mape(t, p, mask), np.mean(np.abs(ts - ps) / np.abs(ts)))
        assert np.isclose(H.smape(t, p, mask), np.mean(np.abs(ps - ts) / (np.abs(ts) + np.abs(ps))))
        assert np.isclose(H.logac(t, p, mask), np.mean(np.abs(np.log(ps / ts))))
        assert np.isclose(H.medsymac(t, p, mask), np.exp(np.median(np.abs(np.log(ps / ts)))) - 1)
    assert H.nrmse(t, t) == 0.0 and H.smape(t, t) == 0.0 and H.medsymac(t, t) == 0.0
    # PSNR: hand-computed on the mask's bounding box with the exclusive upper bound the reference's slicing has
    tc, pc = (t * m)[1:4, 2:6, 3:7], (p * m)[1:4, 2:6, 3:7]
    want = 10 * np.log10((tc.max() - tc.min()) ** 2 / np.mean((tc - pc) ** 2))
    assert np.isclose(H.psnr(t.copy(), p.copy(), m), want)
    z = np.full((4, 4, 4), 3.0)
    assert np.allclose(H.scale12bit(np.array([1.0, 2.0, 3.0])), [2048 - 400 * np.sqrt(1.5), 2048.0, 2048 + 400 * np.sqrt(1.5)])


def test_skimage_metrics_against_real_skimage():
    """peak_signal_noise_ratio / structural_similarity vs values scikit-image 0.18.3 itself computed (tests/golden/metrics.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz"))
    for name in ("v3", "s2"):
        t, p = g[name + "_true"], g[name + "_pred"]
        dr = t.max() - t.min()
        assert abs(H.peak_signal_noise_ratio(t, p, data_range=dr) - float(g[name + "_psnr"])) < 1e-9
        assert abs(H.structural_similarity(t, p, win_size=9, data_range=dr) - float(g[name + "_ssim9"])) < 1e-9
        assert abs(H.structural_similarity(t, p, data_range=dr) - float(g[name + "_ssim7"])) < 1e-9
    # the reference's wrappers on top: mask crop (exclusive upper bound) + 12-bit rescale for ssim, zeroing + crop for psnr
    t, p = g["v3_true"].copy(), g["v3_pred"].copy()
    mask = np.zeros(t.shape, dtype=np.uint8)
    mask[1:13, 4:37, 3:33] = 1
    sl = (slice(1, 12), slice(4, 36), slice(3, 32))
    tt, pp = H.scale12bit(t[sl]), H.scale12bit(p[sl])
    assert H.ssim(t, p, mask) == H.structural_similarity(tt, pp, win_size=9, data_range=tt.max() - tt.min())
    assert 0.5 < H.ssim(t, p, mask) < 1.0
    assert H.psnr(t, p, mask) == H.peak_signal_noise_ratio(t[sl], p[sl], data_range=t[sl].max() - t[sl].min())
