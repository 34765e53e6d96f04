"""ctypes binding of the HDF5 C library where the image has one (test infrastructure only: it validates
drt_amd.hdf5_lite's reader and writer against the genuine implementation; the product never loads it)."""
import ctypes
import ctypes.util
import glob

import numpy as np

hid = ctypes.c_int64
_L = None


def lib():
    global _L
    if _L is None:
        cands = [ctypes.util.find_library("hdf5")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
        for c in cands:
            if not c:
                continue
            try:
                L = ctypes.CDLL(c)
                L.H5open()
            except (OSError, AttributeError):
                continue
            sig = {"H5Fopen": (hid, [ctypes.c_char_p, ctypes.c_uint, hid]), "H5Fcreate": (hid, [ctypes.c_char_p, ctypes.c_uint, hid, hid]),
                   "H5Fclose": (ctypes.c_int, [hid]), "H5Dopen2": (hid, [hid, ctypes.c_char_p, hid]), "H5Dclose": (ctypes.c_int, [hid]),
                   "H5Dget_space": (hid, [hid]), "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid]),
                   "H5Sget_simple_extent_dims": (ctypes.c_int, [hid, ctypes.c_void_p, ctypes.c_void_p]),
                   "H5Dread": (ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
                   "H5Dwrite": (ctypes.c_int, [hid, hid, hid, hid, hid, ctypes.c_void_p]),
                   "H5Screate_simple": (hid, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
                   "H5Pcreate": (hid, [hid]), "H5Pset_chunk": (ctypes.c_int, [hid, ctypes.c_int, ctypes.c_void_p]),
                   "H5Pset_deflate": (ctypes.c_int, [hid, ctypes.c_uint]), "H5Pset_shuffle": (ctypes.c_int, [hid]),
                   "H5Dcreate2": (hid, [hid, ctypes.c_char_p, hid, hid, hid, hid, hid]),
                   "H5Dget_type": (hid, [hid]), "H5Tget_size": (ctypes.c_size_t, [hid]), "H5Tget_class": (ctypes.c_int, [hid])}
            for name, (res, args) in sig.items():
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
            _L = L
            break
    return _L


_NATIVE = {np.dtype("f8"): "H5T_NATIVE_DOUBLE_g", np.dtype("f4"): "H5T_NATIVE_FLOAT_g", np.dtype("u1"): "H5T_NATIVE_UCHAR_g",
           np.dtype("i8"): "H5T_NATIVE_LLONG_g", np.dtype("i4"): "H5T_NATIVE_INT_g", np.dtype("u2"): "H5T_NATIVE_USHORT_g"}


def _native(dt):
    return hid.in_dll(lib(), _NATIVE[np.dtype(dt)]).value


def read(path, name, dtype):
    L = lib()
    f = L.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, "libhdf5 cannot open the file"
    d = L.H5Dopen2(f, name.encode(), 0)
    assert d >= 0, f"libhdf5 cannot open dataset {name}"
    s = L.H5Dget_space(d)
    nd = L.H5Sget_simple_extent_ndims(s)
    dims = (ctypes.c_uint64 * nd)()
    L.H5Sget_simple_extent_dims(s, dims, None)
    t = L.H5Dget_type(d)
    out = np.empty(tuple(dims), dtype)
    assert L.H5Tget_size(t) == np.dtype(dtype).itemsize
    assert L.H5Dread(d, _native(dtype), 0, 0, 0, out.ctypes.data) >= 0
    L.H5Dclose(d)
    L.H5Fclose(f)
    return out


def write(path, arrays, chunks=None, deflate=0, shuffle=False):
    """{name: array} -> HDF5 file written by the library; ``chunks[name]`` = chunk shape (with optional filters)."""
    L = lib()
    f = L.H5Fcreate(path.encode(), 2, 0, 0)          # H5F_ACC_TRUNC
    assert f >= 0
    dcpl_cls = hid.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        dims = (ctypes.c_uint64 * a.ndim)(*a.shape)
        s = L.H5Screate_simple(a.ndim, dims, None)
        pl = 0
        if chunks and name in chunks:
            pl = L.H5Pcreate(dcpl_cls)
            c = (ctypes.c_uint64 * a.ndim)(*chunks[name])
            assert L.H5Pset_chunk(pl, a.ndim, c) >= 0
            if shuffle:
                assert L.H5Pset_shuffle(pl) >= 0
            if deflate:
                assert L.H5Pset_deflate(pl, deflate) >= 0
        d = L.H5Dcreate2(f, name.encode(), _native(a.dtype), s, 0, pl, 0)
        assert d >= 0
        assert L.H5Dwrite(d, _native(a.dtype), 0, 0, 0, a.ctypes.data) >= 0
        L.H5Dclose(d)
    L.H5Fclose(f)
    return path
