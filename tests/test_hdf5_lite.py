"""drt_amd.hdf5_lite (the capture reader used where h5py is missing) against HDF5 files written by the HDF5 C
library and by h5py (tests/golden/hdf5/README.md)."""
import os

import numpy as np
import pytest

from drt_amd import captured_data as cd, hdf5_lite as h5

D = os.path.join(os.path.dirname(__file__), "golden", "hdf5")


@pytest.mark.parametrize("name,dtype", [("smpl_f64le", "<f8"), ("smpl_i32be", ">i4")])
def test_contiguous_arrays_of_both_byte_orders(name, dtype):
    with h5.File(os.path.join(D, name + ".h5")) as f:
        assert f.keys() == ["TestArray"] and "TestArray" in f and "nope" not in f
        d = f["TestArray"]
        assert d.shape == (6, 5) and d.dtype == np.dtype(dtype) and len(d) == 6
        expect = np.add.outer(np.arange(6), np.arange(5)).astype(dtype)
        assert np.array_equal(d[...], expect) and np.array_equal(d[:], expect) and np.array_equal(np.asarray(d), expect)
        assert np.array_equal(d[3], expect[3]) and np.array_equal(d[-1], expect[-1]) and np.array_equal(d[1:3, ::2], expect[1:3, ::2])
        with pytest.raises(IndexError):
            d[6]
        with pytest.raises(KeyError):
            f["missing"]


def test_float_types_in_one_file():
    f = h5.File(os.path.join(D, "float.h5"))
    vals = np.add.outer(np.arange(5), np.arange(6))
    for k in ("float16", "float32", "float64"):
        assert f[k].shape == (5, 6) and f[k].dtype == np.dtype(k) and np.array_equal(f[k][...], vals.astype(k))
    with pytest.raises(h5.Hdf5Unsupported):
        f["longdouble"]                                   # 16-byte floats are refused, not misread


def test_chunked_extendible_array_with_unwritten_chunks():
    d = h5.File(os.path.join(D, "smpl_SDSextendible.h5"))["ExtendibleArray"]
    expect = np.zeros((10, 5), dtype=">i4")
    expect[:3, :3] = 1; expect[3:, 0] = 2; expect[:2, 3:] = 3
    assert d.shape == (10, 5) and d.dtype == np.dtype(">i4") and np.array_equal(d[...], expect)
    assert np.array_equal(d[4], expect[4])
    # row / slab indexing decodes only the overlapping chunks (one view of a capture at a time) -- same values
    for i in range(-10, 10):
        assert np.array_equal(d[i], expect[i])
    for sl in (slice(0, 3), slice(2, 9), slice(7, None), slice(5, 5), slice(-4, -1)):
        assert np.array_equal(d[sl], expect[sl])
    calls = []
    orig = d._unfilter
    d._unfilter = lambda raw, mask: (calls.append(1), orig(raw, mask))[1]
    d[0]
    one_row = len(calls)
    d[...]
    assert 0 < one_row < len(calls) - one_row          # a row touches fewer chunks than the whole dataset
    with pytest.raises(IndexError):
        d[10]


def test_deflate_compressed_chunks_and_nested_groups():
    f = h5.File(os.path.join(D, "attr-u16.h5"))
    d = f["wfm_group0/vectors/vector0/data"]
    assert d.shape == (256, 8) and d.dtype == np.uint8 and d._filters and d._filters[0][0] == 1
    bits = ((np.arange(256)[:, None] >> np.arange(7, -1, -1)[None, :]) & 1).astype(np.uint8)
    assert np.array_equal(d[...], bits)
    assert np.array_equal(f["wfm_group0"]["vectors"]["vector0"]["data"][200], bits[200])
    assert all(np.array_equal(d[i], bits[i]) for i in range(0, 256, 17)) and np.array_equal(d[100:140], bits[100:140])
    assert "wfm_group0" in f and set(f["wfm_group0"].keys()) >= {"axes", "traces", "vectors"}


def test_unsupported_content_is_refused_loudly():
    with pytest.raises(h5.Hdf5Unsupported, match="compound"):
        h5.File(os.path.join(D, "itemsize.h5"))["Test"]    # written by h5py: the file structure parses, the type is refused
    f = h5.File(os.path.join(D, "slink.h5"))
    refused = 0
    for k in f.keys():
        try:
            f[k]
        except h5.Hdf5Unsupported:
            refused += 1
    assert refused >= 1                                       # the soft links
    with pytest.raises(ValueError):
        h5.File(__file__)


def test_capture_loader_falls_back_to_hdf5_lite(monkeypatch):
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):
        if name == "h5py":
            raise ImportError("no h5py here")
        return real_import(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_h5py)
    cap = cd._open_capture(os.path.join(D, "smpl_f64le.h5"))
    assert isinstance(cap, h5.File) and np.asarray(cap["TestArray"][2]).tolist() == [2, 3, 4, 5, 6]


def _capture_like(n=5, resy=12, resx=16, seed=0):
    rng = np.random.default_rng(seed)
    P = resy * resx
    return {"cam_proj": rng.standard_normal((n, 4, 4)), "cam_k": rng.standard_normal((3, 3)),
            "screen_position": rng.standard_normal((n, P, 3)) * (rng.random((n, P, 1)) > 0.7),
            "mask": ((rng.random((n, resy, resx)) > 0.5) * 255).astype(np.uint8),
            "ray_origin": rng.standard_normal((n, P, 3)).astype(np.float32), "ray_dir": rng.standard_normal((n, P, 3))}


def test_writer_is_read_back_by_the_reader_and_by_libhdf5(tmp_path):
    """hdf5_lite.write_simple (what tools/make_capture.py uses without h5py): our reader returns the arrays, and so does
    the HDF5 C library itself where the image has one."""
    arrays = _capture_like()
    arrays["idx"] = np.arange(-5, 5, dtype=np.int64)
    arrays["flag"] = np.array([True, False, True])
    path = str(tmp_path / "cap.h5")
    h5.write_simple(path, arrays)
    with h5.File(path) as f:
        assert sorted(f.keys()) == sorted(arrays)
        for k, v in arrays.items():
            want = v.astype(np.uint8) if v.dtype == np.bool_ else v
            assert f[k].shape == want.shape and f[k].dtype == want.dtype and np.array_equal(f[k][...], want)
            if want.ndim > 1:
                assert np.array_equal(f[k][1], want[1])
    with pytest.raises(ValueError):
        h5.write_simple(path, {str(k): np.zeros(1) for k in range(9)})
    with pytest.raises(h5.Hdf5Unsupported):
        h5.write_simple(path, {"c": np.zeros(2, dtype=np.complex128)})
    import h5lib
    if h5lib.lib() is None:
        pytest.skip("no libhdf5 in this image: the writer is checked by the reader only")
    h5.write_simple(path, arrays)
    for k, v in arrays.items():
        want = v.astype(np.uint8) if v.dtype == np.bool_ else v
        assert np.array_equal(h5lib.read(path, k, want.dtype), want)


@pytest.mark.parametrize("deflate,shuffle", [(0, False), (4, False), (6, True)])
def test_capture_layout_written_by_libhdf5(tmp_path, deflate, shuffle):
    """A file with the capture's layout ([n,P,3] / [n,resy,resx] datasets, chunked one view per chunk as a writer
    of multi-GB captures would, optionally deflate + shuffle) written by the HDF5 C library, read view by view."""
    import h5lib
    if h5lib.lib() is None:
        pytest.skip("no libhdf5 in this image")
    arrays = _capture_like(n=6, seed=3)
    P = arrays["screen_position"].shape[1]
    chunks = {"screen_position": (1, P, 3), "ray_dir": (2, P // 2, 3), "mask": (1, 12, 16), "ray_origin": (4, P, 1)}
    path = str(tmp_path / "lib.h5")
    h5lib.write(path, arrays, chunks=chunks, deflate=deflate, shuffle=shuffle)
    with h5.File(path) as f:
        for k, v in arrays.items():
            d = f[k]
            assert d.shape == v.shape and d.dtype == v.dtype
            assert (d._kind == "chunked") == (k in chunks)
            for i in range(len(v)):
                assert np.array_equal(d[i], v[i]), (k, i)
            assert np.array_equal(d[...], v) and np.array_equal(d[1:4], v[1:4])
