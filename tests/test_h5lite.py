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


@pytest.mark.skipif(not os.path.exists(PY39), reason="no interpreter with h5py in this image")
def test_reader_against_h5py_variants(tmp_path):
    """Files written on the spot by real h5py with the options a slice file can meet: scalar and 1-D..4-D datasets, many
    datasets in one group (several symbol nodes / a deeper B-tree), chunk shapes that do not divide the dataset, gzip levels,
    shuffle, fletcher32, every integer / float width, both byte orders, old and new library bounds (v1 B-tree and
    fixed-array chunk indices, paged and not), nested groups."""
    code = r'''
import sys, h5py, numpy as np
out = sys.argv[1]
rng = np.random.default_rng(123)
exp = {}
def arr(shape, dt):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return rng.standard_normal(shape).astype(dt)
    info = np.iinfo(dt.newbyteorder("="))
    return rng.integers(max(info.min, -1000), min(info.max, 1000), shape).astype(dt)
for tag, kw in (("old", {}), ("v110", {"libver": ("v110", "v110")}), ("new", {"libver": "latest"})):
    with h5py.File(f"{out}/{tag}.h5", "w", **kw) as f:
        items = {}
        for i, dt in enumerate(["<f4", "<f8", "<f2", ">f4", ">f8", "<i1", "<i2", "<i4", "<i8", "<u1", "<u2", "<u4", "<u8", ">i2", ">u4"]):
            items[f"d{i:02d}"] = arr((5, 7), dt)
        items["scalar"] = np.float32(3.25)
        items["vec"] = arr((11,), "<f4")
        items["vol"] = arr((3, 9, 10), "<i2")
        items["four"] = arr((2, 3, 4, 5), "<f8")
        if tag == "old":                      # >8 links in a new-style group = dense storage (outside the subset)
            for k, v in items.items():
                f[k] = v
                exp[f"{tag}/{k}"] = np.asarray(v)
            for j in range(40):
                f[f"many_{j:02d}"] = np.full((2, 2), j, dtype=np.int32)
                exp[f"{tag}/many_{j:02d}"] = np.full((2, 2), j, dtype=np.int32)
        else:
            for k in ("d00", "d03", "d06", "scalar", "vol"):
                f[k] = items[k]
                exp[f"{tag}/{k}"] = np.asarray(items[k])
        g = f.create_group("grp")
        a = arr((37, 29), "<f4")
        g.create_dataset("ragged_chunks", data=a, chunks=(16, 10), compression="gzip", compression_opts=1)
        exp[f"{tag}/grp/ragged_chunks"] = a
        b = arr((64, 48), "<i2")
        g.create_dataset("shuffled", data=b, chunks=(64, 48), compression="gzip", shuffle=True, fletcher32=True)
        exp[f"{tag}/grp/shuffled"] = b
        c = arr((9, 9), ">f8")
        g.create_dataset("chunk_plain", data=c, chunks=(4, 4))
        exp[f"{tag}/grp/chunk_plain"] = c
        e = arr((70, 66), "<i4")
        g.create_dataset("many_chunks", data=e, chunks=(2, 2))              # 1155 chunks: a paged fixed array under v1.10+
        exp[f"{tag}/grp/many_chunks"] = e
        h = arr((70, 66), "<f4")
        g.create_dataset("many_gz", data=h, chunks=(2, 3), compression="gzip")
        exp[f"{tag}/grp/many_gz"] = h
        d = arr((6,), "<u1")
        g.create_dataset("unwritten", shape=(4, 4), dtype="<f4")            # never written: zeros
        exp[f"{tag}/grp/unwritten"] = np.zeros((4, 4), np.float32)
        g.create_group("sub")["deep"] = d
        exp[f"{tag}/grp/sub/deep"] = d
np.savez(f"{out}/exp.npz", **{k: v.astype(v.dtype.newbyteorder("=")) for k, v in exp.items()})
print("ok")
'''
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([PY39, "-c", code, str(tmp_path)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
    exp = np.load(str(tmp_path / "exp.npz"))
    files = {}
    n = 0
    for key in exp.files:
        tag, name = key.split("/", 1)
        f = files.setdefault(tag, h5lite.H5File(str(tmp_path / f"{tag}.h5")))
        got = f[name]
        want = exp[key]
        assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want), key
        n += 1
    assert n >= 90
