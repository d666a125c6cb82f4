"""Host side of the inference entry points, either side of the sampling path — numpy only.

What the reference does around ``sample_fn`` and what is rebuilt here:
  * slice bookkeeping of ``predict_step`` / ``on_predict_batch_end`` / ``on_predict_end``
    (trainers/trainer_use_gaussian_diff.py:602-655): ``.../<id>/<name>_<slice>.h5`` paths -> (id, slice index), 2-D samples
    collected per id and written back into a zero-filled [D,H,W] volume shaped like the template  -> ``parse_slice_path``,
    ``VolumeAssembler``;
  * the volume file format: the reference reads / writes NIfTI through SimpleITK (``sitk.ReadImage`` /
    ``CopyInformation`` / ``WriteImage``, :639-648).  SimpleITK is not installable here, so ``read_nifti`` /
    ``write_nifti`` implement the single-file NIfTI-1 container (348-byte header + voxel block, optionally gzipped) from its
    public specification; "CopyInformation" = the prediction is written with the template's header (geometry) bytes;
  * the array-level metrics of inference/test_metrics.py:21-26,149-224 (scale12bit, NRMSE, MAPE, sMAPE, logac, medsymac) and
    PSNR (:378-400, skimage's definition) as used by inference/get_metric_BraTs.py.
  * the h5 slice files (preprocess/to_h5.py:40-50 writer, training_project/utils/my_transform.py:142-154 ``LoadH5`` reader): the
    HDF5 container is read and written by ``h5lite.py`` (numpy only, pinned by files real h5py wrote) -> ``read_h5``,
    ``write_h5``, ``LoadH5`` re-exported here.
  * the two scikit-image metrics behind ``psnr`` / ``ssim`` (test_metrics.py:7-8,227-246,378-400): ``peak_signal_noise_ratio``,
    ``structural_similarity`` restated from scikit-image 0.18 and PINNED by values real scikit-image computed
    (tests/golden/metrics.npz, generator tests/golden/gen_metrics.py).
  * the "ssim" the metric pass actually reports — torchmetrics' MS-SSIM per axial slice (test_metrics.py:249-274) — restated
    from its published definition (``multiscale_structural_similarity``, ``ssim_torch``; parity unpinned: torchmetrics is
    absent), and the per-id table + mean row of inference/get_metric_BraTs.py:54-126 (``metric_table``, written as CSV).
Not rebuilt: CW-SSIM (pyssim), FID / LPIPS (network weights), ANTs similarity / Mattes mutual information (the table's ``mi``
column holds NaN).

PARITY UNPINNED for the array metrics and the slice bookkeeping: inference/test_metrics.py and the trainers do not import
here (ants, lpips, SimpleITK, Lightning), and the reference holds no fixtures for these functions; tests/test_host_io.py checks
them against their defining formulas.  The HDF5 container and the scikit-image metrics ARE pinned (above).
"""
from __future__ import annotations

import gzip
import os
import struct
from collections import defaultdict
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

from .h5lite import H5File, LoadH5, read_h5, write_h5  # noqa: F401  (the h5 side of f-2)


# ------------------------------------------------------------------------------------------------ slice bookkeeping
def parse_slice_path(path: str) -> Tuple[str, int]:
    """trainer_use_gaussian_diff.py:608-609: id = parent directory name, slice index = the integer after the last '_'."""
    return path.split("/")[-2], int(os.path.basename(path).split(".")[0].split("_")[-1])


class VolumeAssembler:
    """on_predict_batch_end / on_predict_end (:625-655): collect 2-D samples per volume id, then stack them into
    ``np.zeros_like(template)`` at their slice index (slices that were never predicted stay zero)."""

    def __init__(self):
        self.pred: Dict[str, Dict[int, np.ndarray]] = defaultdict(dict)

    def add_batch(self, ids: Iterable[str], slice_idx: Iterable[int], images) -> None:
        images = np.asarray(images.detach().cpu().numpy() if hasattr(images, "detach") else images)
        for id_, si, img in zip(ids, slice_idx, images):
            self.pred[id_][int(si)] = np.asarray(img)

    def add_paths(self, paths: Iterable[str], images) -> None:
        parsed = [parse_slice_path(p) for p in paths]
        self.add_batch([p[0] for p in parsed], [p[1] for p in parsed], images)

    def ids(self):
        return list(self.pred.keys())

    def volume(self, id_: str, template: Optional[np.ndarray] = None, depth: Optional[int] = None) -> np.ndarray:
        slices = self.pred[id_]
        if template is not None:
            vol = np.zeros_like(template)
        else:
            first = next(iter(slices.values()))
            hw = first.shape[-2:]
            vol = np.zeros((depth if depth is not None else max(slices) + 1,) + tuple(hw), dtype=first.dtype)
        for si, img in slices.items():
            vol[si] = img.reshape(vol.shape[1:])      # a [1,H,W] sample broadcasts into pred_array[slice] the same way (:646)
        return vol


def find_slice_files(root: str) -> list:
    """Every ``<root>/<id>/<name>_<slice>.h5`` (the layout preprocess/to_h5.py:38-41 writes), ordered by (id, slice index)."""
    found = []
    for id_ in sorted(os.listdir(root)):
        d = os.path.join(root, id_)
        if not os.path.isdir(d):
            continue
        for fn in os.listdir(d):
            if fn.endswith(".h5"):
                path = os.path.join(d, fn)
                try:
                    found.append((id_, parse_slice_path(path)[1], path))
                except ValueError:      # stats.h5, a MATLAB export, ...: not a <name>_<slice>.h5 slice file
                    import warnings
                    warnings.warn(f"{path}: name does not end in _<slice index>; skipped")
    found.sort()
    return [f[2] for f in found]


def load_condition_slices(paths: Iterable[str], keys: Iterable[str]) -> np.ndarray:
    """[N, len(keys), H, W] float32: dataset ``key`` of every slice file as one condition channel (the trainer's
    ``torch.cat([batch[k] for k in condition_keys], 1)``, trainer_use_gaussian_diff.py:604-606).  No intensity scaling or
    resizing is applied: the reference does those in its MONAI transform chain before the model sees the slice."""
    keys = list(keys)
    out = []
    for p in paths:
        f = H5File(p)
        out.append(np.stack([np.asarray(f[k], dtype=np.float32) for k in keys]))
    return np.stack(out) if out else np.zeros((0, len(keys), 0, 0), dtype=np.float32)


# ------------------------------------------------------------------------------------------------ NIfTI-1 container
_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
                 768: np.uint32}
_NIFTI_CODES = {np.dtype(v).str[1:]: k for k, v in _NIFTI_DTYPES.items()}


def read_nifti(path: str):
    """-> (array [D,H,W] in SimpleITK's GetArrayFromImage order (z, y, x), header bytes).  Single-file NIfTI-1
    (.nii / .nii.gz), 3-D, scl_slope / scl_inter applied when set."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    hdr = raw[:348]
    end = "<" if struct.unpack("<i", hdr[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", hdr[:4])[0] != 348:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    dim = struct.unpack(end + "8h", hdr[40:56])
    datatype, bitpix = struct.unpack(end + "2h", hdr[70:74])
    vox_offset = int(struct.unpack(end + "f", hdr[108:112])[0])
    slope, inter = struct.unpack(end + "2f", hdr[112:120])
    if datatype not in _NIFTI_DTYPES:
        raise ValueError(f"{path}: NIfTI datatype {datatype} unsupported")
    nd = dim[0]
    shape = tuple(int(d) for d in dim[1:1 + nd])
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(end)
    n = int(np.prod(shape))
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=max(vox_offset, 352)).reshape(shape[::-1])   # file order: x fastest
    arr = arr.astype(dt.newbyteorder("="))
    while arr.ndim > 3 and arr.shape[0] == 1:        # dim[0] = 4 with a trailing unit time axis: still a 3-D volume
        arr = arr[0]
    # scl_slope = 0 (or a non-finite value: some writers leave NaN there) means "no scaling" per the NIfTI-1 specification
    scaled = np.isfinite(slope) and slope != 0.0 and np.isfinite(inter) and not (slope == 1.0 and inter == 0.0)
    if scaled:
        arr = arr.astype(np.float32) * np.float32(slope) + np.float32(inter)
    return arr, bytes(hdr)


def write_nifti(path: str, array: np.ndarray, template_header: Optional[bytes] = None, spacing=(1.0, 1.0, 1.0)) -> None:
    """Write [D,H,W] (z, y, x) as single-file NIfTI-1.  With ``template_header`` the geometry (pixdim, qform / sform) is the
    template's — the role of ``pred_nii.CopyInformation(template_nii)`` (:648); datatype and dims follow ``array``."""
    array = np.ascontiguousarray(array)
    code = _NIFTI_CODES.get(array.dtype.str[1:])
    if code is None:
        raise ValueError(f"dtype {array.dtype} has no NIfTI-1 code")
    if template_header is not None:
        hdr = bytearray(template_header[:348])
        end = "<" if struct.unpack("<i", hdr[:4])[0] == 348 else ">"
    else:
        hdr, end = bytearray(348), "<"
        struct.pack_into(end + "i", hdr, 0, 348)
        struct.pack_into(end + "8f", hdr, 76, 1.0, float(spacing[2]), float(spacing[1]), float(spacing[0]), 1.0, 1.0, 1.0, 1.0)
        struct.pack_into(end + "h", hdr, 252, 0)           # qform_code
        struct.pack_into(end + "h", hdr, 254, 1)           # sform_code: scanner-anat, identity scaled by the spacing
        struct.pack_into(end + "4f", hdr, 280, float(spacing[2]), 0.0, 0.0, 0.0)
        struct.pack_into(end + "4f", hdr, 296, 0.0, float(spacing[1]), 0.0, 0.0)
        struct.pack_into(end + "4f", hdr, 312, 0.0, 0.0, float(spacing[0]), 0.0)
        hdr[123] = 10                                       # xyzt_units: mm + s
    dims = [array.ndim] + list(array.shape[::-1]) + [1] * (7 - array.ndim)
    struct.pack_into(end + "8h", hdr, 40, *dims)
    struct.pack_into(end + "2h", hdr, 70, code, array.dtype.itemsize * 8)
    struct.pack_into(end + "f", hdr, 108, 352.0)
    struct.pack_into(end + "2f", hdr, 112, 1.0, 0.0)         # the data are stored unscaled
    hdr[344:348] = b"n+1\0"
    data = array.astype(array.dtype.newbyteorder(end), copy=False).tobytes()
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(bytes(hdr) + b"\0\0\0\0" + data)


# ------------------------------------------------------------------------------------------------ metrics
def _mask(true_array, mask):
    return mask.astype(bool) if mask is not None else np.ones_like(true_array, dtype=bool)


def scale12bit(img):
    """test_metrics.py:21-26"""
    return np.clip(((img - np.mean(img)) / (np.std(img) / 400.)) + 2048., 1e-10, 4095)


def nrmse(true_array, pred_array, mask=None):
    """:149-160 — RMSE / (max - min) of the reference inside the mask."""
    m = _mask(true_array, mask)
    t, p = true_array[m], pred_array[m]
    return float(np.sqrt(np.mean((t - p) ** 2)) / (np.max(t) - np.min(t)))


def mape(true_array, pred_array, mask=None):
    """:163-176 (on the 12-bit rescaled images)"""
    m = _mask(true_array, mask)
    t, p = scale12bit(true_array[m]), scale12bit(pred_array[m])
    return float(np.mean(np.fabs(t - p) / np.fabs(t)))


def smape(true_array, pred_array, mask=None):
    """:179-192"""
    m = _mask(true_array, mask)
    t, p = scale12bit(true_array[m]), scale12bit(pred_array[m])
    return float(np.mean(np.fabs(p - t) / (np.fabs(t) + np.fabs(p))))


def logac(true_array, pred_array, mask=None):
    """:195-207"""
    m = _mask(true_array, mask)
    t, p = scale12bit(true_array[m]), scale12bit(pred_array[m])
    return float(np.mean(np.fabs(np.log(p / t))))


def medsymac(true_array, pred_array, mask=None):
    """:211-223 — median symmetric accuracy (Morley 2016)"""
    m = _mask(true_array, mask)
    t, p = scale12bit(true_array[m]), scale12bit(pred_array[m])
    return float(np.exp(np.median(np.fabs(np.log(p / t)))) - 1)


def peak_signal_noise_ratio(image_true, image_test, data_range):
    """scikit-image's metric (skimage.metrics.peak_signal_noise_ratio, 0.18): both images as float64, 10 log10(R^2 / MSE)."""
    t, p = np.asarray(image_true, dtype=np.float64), np.asarray(image_test, dtype=np.float64)
    return float(10.0 * np.log10(float(data_range) ** 2 / np.mean((t - p) ** 2)))


def structural_similarity(im1, im2, win_size=7, data_range=None, K1=0.01, K2=0.03):
    """scikit-image's mean SSIM (skimage.metrics.structural_similarity, 0.18 defaults: uniform window over every axis,
    sample covariance, float64): local means / (co)variances by a ``win_size``^ndim box filter (borders reflected), the SSIM
    map  (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)),  C = (K R)^2, averaged over the interior (the
    (win_size - 1) / 2 border is cropped)."""
    from scipy.ndimage import uniform_filter
    x, y = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    if x.shape != y.shape:
        raise ValueError("Input images must have the same dimensions.")
    if win_size % 2 != 1 or min(x.shape) < win_size:
        raise ValueError("win_size must be odd and not exceed any image side")
    if data_range is None:
        raise ValueError("data_range must be given for floating-point images")
    npix = win_size ** x.ndim
    cov_norm = npix / (npix - 1.0)
    ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
    uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    R = float(data_range)
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[tuple(slice(pad, n - pad) for n in S.shape)].mean(dtype=np.float64))


def ssim(true_array, pred_array, mask=None):
    """:227-246 — crop to the mask's bounding box, 12-bit rescale, scikit-image's SSIM with a 9^3 window and the cropped
    reference's range.  (The reference's version only defines the cropped images when a mask is given; without one the whole
    volume is used here.)"""
    t, p = np.asarray(true_array), np.asarray(pred_array)
    if mask is not None:
        nz = np.nonzero(mask.astype(bool))
        sl = tuple(slice(int(a.min()), int(a.max())) for a in nz)
        t, p = t[sl], p[sl]
    t, p = scale12bit(t), scale12bit(p)
    return structural_similarity(t, p, win_size=9, data_range=t.max() - t.min())


def _gauss1d(size, sigma):
    d = np.arange((1 - size) / 2, (1 + size) / 2, 1.0, dtype=np.float64)
    g = np.exp(-((d / sigma) ** 2) / 2)
    return g / g.sum()


def multiscale_structural_similarity(preds, target, data_range=None, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03,
                                     betas=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
    """MS-SSIM of ONE 2-D image pair (Wang, Simoncelli, Bovik 2003), the form torchmetrics'
    ``multiscale_structural_similarity_index_measure`` evaluates with its defaults — what the reference's metric pass
    reports as "ssim" (inference/test_metrics.py:249-274): per scale a Gaussian window (11 taps, sigma 1.5, separable),
    local means / biased (co)variances, luminance-contrast-structure map  (2 mx my + c1)(2 sxy + c2) / ((mx^2 + my^2 +
    c1)(sx + sy + c2))  and contrast-structure map  (2 sxy + c2) / (sx + sy + c2), each averaged over the window-valid
    interior and clamped at 0 ("relu" normalisation); 2x2 average pooling between the five scales; result
    prod_{i<5} cs_i^beta_i * ssim_5^beta_5.  data_range None -> max(range(preds), range(target)); c = (k R)^2.
    float64 throughout.  PARITY UNPINNED: torchmetrics is not installable here and the reference holds no value of it."""
    from scipy.ndimage import correlate1d
    x, y = np.asarray(preds, dtype=np.float64), np.asarray(target, dtype=np.float64)
    if x.shape != y.shape or x.ndim != 2:
        raise ValueError("expected two 2-D images of the same shape")
    if min(x.shape) // 2 ** (len(betas) - 1) <= kernel_size - 1:
        raise ValueError(f"image {x.shape} too small for {len(betas)} scales of an {kernel_size}-tap window")
    R = float(max(x.max() - x.min(), y.max() - y.min())) if data_range is None else float(data_range)
    c1, c2 = (k1 * R) ** 2, (k2 * R) ** 2
    g = _gauss1d(kernel_size, sigma)
    pad = (kernel_size - 1) // 2
    blur = lambda a: correlate1d(correlate1d(a, g, axis=0, mode="mirror"), g, axis=1, mode="mirror")
    vals = []
    for i in range(len(betas)):
        mx, my = blur(x), blur(y)
        sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
        upper, lower = 2 * sxy + c2, sxx + syy + c2
        full = ((2 * mx * my + c1) * upper) / ((mx * mx + my * my + c1) * lower)
        inner = (slice(pad, -pad), slice(pad, -pad))
        sim, cs = max(float(full[inner].mean()), 0.0), max(float((upper / lower)[inner].mean()), 0.0)
        vals.append(sim if i == len(betas) - 1 else cs)
        if i < len(betas) - 1:                                  # F.avg_pool2d(.., (2, 2)): trailing odd row / column dropped
            h2, w2 = (x.shape[0] // 2) * 2, (x.shape[1] // 2) * 2
            x = x[:h2, :w2].reshape(h2 // 2, 2, w2 // 2, 2).mean(axis=(1, 3))
            y = y[:h2, :w2].reshape(h2 // 2, 2, w2 // 2, 2).mean(axis=(1, 3))
    return float(np.prod(np.asarray(vals) ** np.asarray(betas, dtype=np.float64)))


def ssim_torch(true_array, pred_array, mask=None):
    """inference/test_metrics.py:249-274 — the "ssim" column of the reference's metric table: zero outside the mask, crop
    to the mask's bounding box (exclusive upper bound, as the reference slices), 12-bit rescale of each volume, MS-SSIM of
    every axial slice pair, averaged.  (The reference zeroes its INPUT arrays in place; copies are used here.)"""
    m = _mask(true_array, mask)
    t, p = np.where(m, true_array, 0), np.where(m, pred_array, 0)
    nz = np.nonzero(m)
    sl = tuple(slice(int(a.min()), int(a.max())) for a in nz)
    t, p = scale12bit(t[sl]), scale12bit(p[sl])
    return float(np.mean([multiscale_structural_similarity(p[i], t[i]) for i in range(t.shape[0])]))


METRIC_COLUMNS = ["nrmse", "smape", "logac", "medsymac", "cc", "mi", "ssim", "lpips", "fid", "psnr"]


def metric_row(gt_img, pred_img, mask_img=None):
    """One row of inference/get_metric_BraTs.py:78-104 (without the id): the columns the reference fills with a constant 0
    (cc, lpips, fid: their calls are commented out there) stay 0; ``mi`` is ANTs' Mattes mutual information in the
    reference (test_metrics.py:77-90) — ANTs is not available and is not restated: the column holds NaN."""
    gt_img, pred_img = np.asarray(gt_img, dtype=np.float64), np.asarray(pred_img, dtype=np.float64)
    return [nrmse(gt_img, pred_img, mask_img), smape(gt_img, pred_img, mask_img), logac(gt_img, pred_img, mask_img),
            medsymac(gt_img, pred_img, mask_img), 0.0, float("nan"), ssim_torch(gt_img, pred_img, mask_img), 0.0, 0.0,
            psnr(gt_img, pred_img, mask_img)]


def metric_table(pred_files, gt_dir, gt_name="ce.nii.gz", mask_name=None):
    """get_metric_BraTs.py:54-115: for every prediction ``<id>_..._pred.nii.gz`` (id = text before the first '_') or
    ``{id: path}`` entry, the ground truth ``<gt_dir>/<id>/<gt_name>`` (and, with mask_name, ``<gt_dir>/<id>/<mask_name>``
    > 0 as mask); returns the table with the MEAN row first (float32 mean over the ids, as the reference computes it) and
    the id rows after it: [["ids"] + METRIC_COLUMNS header not included]."""
    if not isinstance(pred_files, dict):
        pred_files = {os.path.basename(f).split(".")[0].split("_")[0]: f for f in pred_files if f.endswith(".nii.gz")}
    rows = []
    for id_, path in pred_files.items():
        gt, _ = read_nifti(os.path.join(gt_dir, id_, gt_name))
        pred, _ = read_nifti(path)
        mask = None
        if mask_name:
            mask = read_nifti(os.path.join(gt_dir, id_, mask_name))[0] > 0
        rows.append([str(id_)] + metric_row(gt, pred, mask))
    if not rows:
        return []
    mean = np.mean(np.asarray([r[1:] for r in rows], dtype=np.float32), axis=0)
    return [[0] + [float(v) for v in mean]] + rows


def write_metric_csv(path, table):
    """The sheet get_metric_BraTs.py:119-125 writes (header ids + the ten columns, mean row first), as CSV."""
    import csv
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["ids"] + METRIC_COLUMNS)
        for r in table:
            w.writerow(r)


def psnr(true_array, pred_array, mask=None):
    """:378-400 — crop to the mask's bounding box (exclusive upper bound, as the reference slices), zero outside the mask,
    then 10 log10(data_range^2 / MSE) with data_range = max - min of the cropped reference (skimage's definition)."""
    m = _mask(true_array, mask)
    t, p = np.where(m, true_array, 0), np.where(m, pred_array, 0)
    nz = np.nonzero(m)
    sl = tuple(slice(int(a.min()), int(a.max())) for a in nz)
    t, p = t[sl], p[sl]
    return peak_signal_noise_ratio(t, p, data_range=t.max() - t.min())
