"""HDF5 slice files (SURVEY f-2): the numpy reader against files REAL h5py wrote the way the reference does
(tests/golden/h5/, generator tests/golden/gen_h5.py), and real h5py against what the numpy writer wrote."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from diffusion_models_dsdiff_amd import h5lite, host_io

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
PY39 = "/opt/conda/bin/python3.9"   # the one interpreter of the image with h5py (absent elsewhere: those tests skip)


@pytest.fixture(scope="module")
def expected():
    return dict(np.load(os.path.join(H5, "expected.npz")))


@pytest.mark.parametrize("name", ["layer_3", "latest", "brats_slice"])
def test_reader_matches_h5py_files(expected, name):
    """layer_3: to_h5.py's `f[key] = array` (superblock v0, symbol table, contiguous); latest: libver='latest' (superblock v3,
    object header v2, link messages); brats_slice: chunked + gzip + shuffle + fletcher32, big-endian, nested group."""
    f = h5lite.H5File(os.path.join(H5, name + ".h5"))
    want = {k.split("/", 1)[1]: v for k, v in expected.items() if k.startswith(name + "/")}
    top = sorted({k.split("/")[0] for k in want})
    assert f.keys() == top
    for k, v in want.items():
        got = f[k]
        assert got.dtype == v.dtype and got.shape == v.shape, k
        assert np.array_equal(got, v), k                      # bit-exact: byte work
        assert got.dtype.isnative


def test_load_h5_transform(expected):
    """LoadH5(path_key, keys) as the reference's data pipeline calls it (my_transform.py:142-154)."""
    p = os.path.join(H5, "layer_3.h5")
    d = host_io.LoadH5("path", ["F_Data1", "S_Data2"])({"path": p, "other": 1})
    assert set(d) == {"path", "other", "F_Data1", "S_Data2"} and d["path"] == p
    assert np.array_equal(d["F_Data1"], expected["layer_3/F_Data1"])
    assert np.array_equal(d["S_Data2"], expected["layer_3/S_Data2"])
    with pytest.raises(KeyError):
        host_io.LoadH5("path", ["nope"])({"path": p})


def test_writer_reader_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    arrays = {"F_Data1": rng.standard_normal((17, 33)).astype(np.float32), "F_Data2": rng.standard_normal((17, 33)),
              "S_Data1": rng.integers(-5, 5, (17, 33)).astype(np.int16), "S_Data2": rng.integers(0, 255, (4, 5, 6)).astype(np.uint8),
              "half": rng.standard_normal(7).astype(np.float16), "scalar_like": np.array([3], dtype=np.int64),
              "empty": np.zeros((0, 4), dtype=np.float32), "be": rng.standard_normal((3, 2)).astype(">f8"), "k9": np.arange(9, dtype=np.uint32),
              "k10": np.arange(10, dtype=np.int8)}   # 10 names: more than the default symbol node holds
    p = str(tmp_path / "w.h5")
    host_io.write_h5(p, arrays)
    got = host_io.read_h5(p)
    assert sorted(got) == sorted(arrays)
    for k, v in arrays.items():
        assert got[k].shape == v.shape and np.array_equal(got[k], v), k
    with pytest.raises(ValueError):
        host_io.write_h5(p, {})
    with pytest.raises(ValueError):
        host_io.write_h5(p, {"a/b": np.zeros(1)})
    with pytest.raises(TypeError):
        host_io.write_h5(p, {"c": np.zeros(2, dtype=np.complex64)})


def test_not_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file" * 10)
    with pytest.raises(h5lite.H5FormatError):
        h5lite.H5File(str(p))


def test_matlab_v73_user_block():
    """A file libhdf5 wrote with a 512-byte user block (scipy's MATLAB 7.3 sample, if this image has it)."""
    p = "/usr/local/lib/python3.10/dist-packages/scipy/io/matlab/tests/data/testhdf5_7.4_GLNX86.mat"
    if not os.path.exists(p):
        pytest.skip("sample file not in this image")
    f = h5lite.H5File(p)
    assert f.base == 512 and len(f.keys()) > 0
    read = 0
    for k in f.keys():
        try:
            a = f[k]
            read += 1
            assert a.size > 0
        except (NotImplementedError, h5lite.H5FormatError):
            pass   # MATLAB strings / cells / structs use types outside the subset
    assert read > 0


@pytest.mark.skipif(not os.path.exists(PY39), reason="no interpreter with h5py in this image")
def test_h5py_reads_what_write_h5_wrote(tmp_path):
    """The other direction: real h5py / libhdf5 opens the writer's file and returns the same arrays."""
    rng = np.random.default_rng(6)
    arrays = {"F_Data1": rng.standard_normal((24, 20)).astype(np.float32), "F_Data2": rng.standard_normal((24, 20)),
              "S_Data1": rng.integers(-2000, 4000, (24, 20)).astype(np.int16), "S_Data2": rng.integers(0, 255, (24, 20)).astype(np.uint8)}
    p = str(tmp_path / "layer_0.h5")
    host_io.write_h5(p, arrays)
    np.savez(str(tmp_path / "want.npz"), **arrays)
    code = ("import h5py, numpy as np, sys\n"
            "want = np.load(sys.argv[2])\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "assert sorted(f.keys()) == sorted(want.files), list(f.keys())\n"
            "for k in want.files:\n"
            "    a = f[k][()]\n"
            "    assert a.dtype == want[k].dtype and a.shape == want[k].shape and np.array_equal(a, want[k]), k\n"
            "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([PY39, "-c", code, p, str(tmp_path / "want.npz")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


def test_slice_directory_to_condition_batch(tmp_path):
    """to_h5.py's directory layout -> ordered slice list -> [N,C,H,W] condition batch -> volumes at their slice index."""
    rng = np.random.default_rng(7)
    vols = {"case_b": rng.standard_normal((3, 2, 8, 8)).astype(np.float32), "case_a": rng.standard_normal((12, 2, 8, 8)).astype(np.float32)}
    for id_, v in vols.items():
        os.makedirs(tmp_path / id_)
        for z in range(v.shape[0]):
            host_io.write_h5(str(tmp_path / id_ / f"layer_{z}.h5"), {"F_Data1": v[z, 0], "F_Data2": v[z, 1]})
    paths = host_io.find_slice_files(str(tmp_path))
    assert [host_io.parse_slice_path(p) for p in paths] == [("case_a", z) for z in range(12)] + [("case_b", z) for z in range(3)]
    cond = host_io.load_condition_slices(paths, ["F_Data2", "F_Data1"])
    assert cond.shape == (15, 2, 8, 8) and cond.dtype == np.float32
    assert np.array_equal(cond[:12, 0], vols["case_a"][:, 1]) and np.array_equal(cond[12:, 1], vols["case_b"][:, 0])
    asm = host_io.VolumeAssembler()
    asm.add_paths(paths, cond[:, :1])
    assert np.array_equal(asm.volume("case_a"), vols["case_a"][:, 1])   # slice 10 lands behind slice 9, not behind slice 1
