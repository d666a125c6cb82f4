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


def _ms_ssim_torch(preds, target, betas=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
    """Independent restatement of the published MS-SSIM in torch ops, laid out as a convolution pipeline (reflect padding,
    2-D Gaussian kernel, crop) instead of host_io's separable scipy filters."""
    import torch
    import torch.nn.functional as F
    x = torch.as_tensor(preds, dtype=torch.float64)[None, None]
    y = torch.as_tensor(target, dtype=torch.float64)[None, None]
    R = max(float(x.max() - x.min()), float(y.max() - y.min()))
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    d = torch.arange(-5, 6, dtype=torch.float64)
    g = torch.exp(-(d / 1.5) ** 2 / 2)
    g = g / g.sum()
    k = (g[:, None] * g[None, :])[None, None]
    vals = []
    for i in range(5):
        xp, yp = F.pad(x, (5, 5, 5, 5), mode="reflect"), F.pad(y, (5, 5, 5, 5), mode="reflect")
        stack = torch.cat([xp, yp, xp * xp, yp * yp, xp * yp])
        mx, my, xx, yy, xy = F.conv2d(stack, k).split(1)
        sx, sy, sxy = xx - mx * mx, yy - my * my, xy - mx * my
        up, lo = 2 * sxy + c2, sx + sy + c2
        full = ((2 * mx * my + c1) * up / ((mx * mx + my * my + c1) * lo))[..., 5:-5, 5:-5].mean()
        cs = (up / lo)[..., 5:-5, 5:-5].mean()
        vals.append(torch.relu(full if i == 4 else cs))
        x, y = F.avg_pool2d(x, 2), F.avg_pool2d(y, 2)
    return float(torch.prod(torch.stack(vals) ** torch.tensor(betas, dtype=torch.float64)))


def test_ms_ssim_definition_and_reference_wrapper():
    """MS-SSIM (what the reference's metric pass reports as "ssim", test_metrics.py:249-274 through torchmetrics): PARITY
    UNPINNED (torchmetrics absent, no value of it in the reference) — checked against an independent torch restatement of
    the published definition, its limiting cases, and the wrapper's crop / rescale / per-slice mean written out."""
    rng = np.random.default_rng(5)
    t = rng.uniform(0, 1, (200, 183))
    assert abs(H.multiscale_structural_similarity(t, t) - 1.0) < 1e-12
    prev = 1.0
    for noise in (0.02, 0.1, 0.4):
        p = t + noise * rng.standard_normal(t.shape)
        v = H.multiscale_structural_similarity(p, t)
        assert abs(v - _ms_ssim_torch(p, t)) < 1e-10 and 0.0 < v < prev
        assert abs(v - H.multiscale_structural_similarity(t, p)) < 1e-12        # symmetric
        prev = v
    with pytest.raises(ValueError):
        H.multiscale_structural_similarity(t[:150, :150], t[:150, :150])          # 5 scales of an 11-tap window need > 160 px
    vol_t = rng.uniform(0, 1, (4, 190, 200))
    vol_p = vol_t + 0.05 * rng.standard_normal(vol_t.shape)
    mask = np.zeros(vol_t.shape, np.uint8)
    mask[0:4, 5:190, 8:196] = 1
    tt, pp = H.scale12bit((vol_t * mask)[0:3, 5:189, 8:195]), H.scale12bit((vol_p * mask)[0:3, 5:189, 8:195])
    want = np.mean([H.multiscale_structural_similarity(pp[i], tt[i]) for i in range(3)])
    keep = vol_t.copy()
    assert abs(H.ssim_torch(vol_t, vol_p, mask) - want) < 1e-12 and np.array_equal(vol_t, keep)   # inputs are not modified


def test_metric_table_and_csv(tmp_path):
    """inference/get_metric_BraTs.py:54-126: one row per id (nrmse, smape, logac, medsymac, cc = 0, mi = NaN [ANTs, not built],
    ssim = MS-SSIM, lpips = 0, fid = 0, psnr), the float32 mean row FIRST, header ids + the ten names."""
    import csv
    rng = np.random.default_rng(6)
    gt_dir, pred_dir = tmp_path / "gt", tmp_path / "pred"
    pred_dir.mkdir()
    files = []
    for k, id_ in enumerate(("0007", "0123", "0042")):
        (gt_dir / id_).mkdir(parents=True)
        gt = rng.uniform(0, 1, (3, 200, 208)).astype(np.float32)
        H.write_nifti(str(gt_dir / id_ / "ce.nii.gz"), gt)
        H.write_nifti(str(pred_dir / f"{id_}_task_pred.nii.gz"), (gt + 0.03 * (k + 1) * rng.standard_normal(gt.shape)).astype(np.float32))
        files.append(str(pred_dir / f"{id_}_task_pred.nii.gz"))
    (pred_dir / "notes.txt").write_text("x")
    table = H.metric_table(files + [str(pred_dir / "notes.txt")], str(gt_dir))
    assert [r[0] for r in table] == [0, "0007", "0123", "0042"] and all(len(r) == 11 for r in table)
    rows = np.asarray([r[1:] for r in table[1:]], dtype=np.float32)
    assert np.allclose(table[0][1:], rows.mean(axis=0), equal_nan=True)
    col = {n: i + 1 for i, n in enumerate(H.METRIC_COLUMNS)}
    gt0, _ = H.read_nifti(str(gt_dir / "0007" / "ce.nii.gz"))
    p0, _ = H.read_nifti(files[0])
    assert table[1][col["psnr"]] == H.psnr(gt0.astype(np.float64), p0.astype(np.float64))
    assert table[1][col["ssim"]] == H.ssim_torch(gt0.astype(np.float64), p0.astype(np.float64)) and table[1][col["cc"]] == 0.0
    assert np.isnan(table[1][col["mi"]]) and table[1][col["nrmse"]] < table[2][col["nrmse"]] < table[3][col["nrmse"]]
    out = tmp_path / "m.csv"
    H.write_metric_csv(str(out), table)
    back = list(csv.reader(open(out)))
    assert back[0] == ["ids"] + H.METRIC_COLUMNS and len(back) == 5 and back[2][0] == "0007"


def test_host_io_robustness_cases(tmp_path):
    """stray .h5 names are skipped with a warning, a NaN scl_slope means "no scaling", a 4-D header with a unit
    time axis reads as a 3-D volume."""
    d = tmp_path / "in" / "idA"
    d.mkdir(parents=True)
    for name in ("t1_0.h5", "t1_3.h5", "stats.h5", "export_final.h5"):
        H.write_h5(str(d / name), {"F_Data1": np.zeros((4, 4), np.float32)})
    with pytest.warns(UserWarning, match="skipped"):
        found = H.find_slice_files(str(tmp_path / "in"))
    assert [p.split("/")[-1] for p in found] == ["t1_0.h5", "t1_3.h5"]
    vol = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    p = str(tmp_path / "v.nii")
    H.write_nifti(p, vol)
    raw = bytearray(open(p, "rb").read())
    struct.pack_into("<2f", raw, 112, float("nan"), 0.0)          # scl_slope = NaN
    struct.pack_into("<8h", raw, 40, 4, 4, 3, 2, 1, 1, 1, 1)      # dim[0] = 4, trailing unit time axis
    open(p, "wb").write(bytes(raw))
    back, _ = H.read_nifti(p)
    assert back.shape == (2, 3, 4) and back.dtype == np.int16 and np.array_equal(back, vol)
    struct.pack_into("<2f", raw, 112, 2.0, 1.0)
    open(p, "wb").write(bytes(raw))
    assert np.array_equal(H.read_nifti(p)[0], vol.astype(np.float32) * 2 + 1)
