"""HDF5 prediction files without h5py: a small ctypes binding of the HDF5 C library itself.

The reference writes `predictions.h5` through h5py (gluefactory/utils/export_predictions.py:33,81-90: one group per
`"<seq>/<idx>.ppm"` name, one dataset per key) and reads it back in `CacheLoader` (models/cache_loader.py:91-171).
h5py is not installed in this image, but the HDF5 shared library is (`/opt/conda/lib/libhdf5.so*`, 1.10.x); this module
binds the handful of C entry points needed to produce and read exactly that layout -- the files are ordinary HDF5
files (h5py / h5dump read them).  `export_predictions` prefers h5py when it is importable, then this binding, then
its `.npz` container.

Plumbing only (file IO on the host); nothing here is on the GPU path.
"""
import ctypes
import ctypes.util
import glob
import os
from ctypes import POINTER, c_char_p, c_int, c_int64, c_size_t, c_uint, c_ulonglong, c_void_p

import numpy as np

hid_t = c_int64
herr_t = c_int
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5I_GROUP, H5I_DATASET = 2, 5
H5_INDEX_NAME, H5_ITER_INC = 0, 0

_lib = None
_native = {}


class Hdf5Unavailable(RuntimeError):
    pass


def _find():
    cands = []
    if os.environ.get("GFC_HDF5_LIB"):
        cands.append(os.environ["GFC_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        cands += sorted(p for p in glob.glob(pat) if "_hl" not in p and "_cpp" not in p and "fortran" not in p)
    return cands


def available() -> bool:
    try:
        lib()
        return True
    except Hdf5Unavailable:
        return False


def lib():
    """dlopen libhdf5 once, declare the prototypes used below, initialise the library (H5open)."""
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for path in _find():
        try:
            h = ctypes.CDLL(path)
            break
        except OSError as e:  # keep looking
            err = e
    else:
        raise Hdf5Unavailable(f"no loadable HDF5 C library found ({err})")
    proto = {
        "H5open": (herr_t, []),
        "H5Eset_auto2": (herr_t, [hid_t, c_void_p, c_void_p]),
        "H5Fcreate": (hid_t, [c_char_p, c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [c_char_p, c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_create_intermediate_group": (herr_t, [hid_t, c_uint]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Screate": (hid_t, [c_int]),
        "H5Screate_simple": (hid_t, [c_int, POINTER(c_ulonglong), POINTER(c_ulonglong)]),
        "H5Sget_simple_extent_ndims": (c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (c_int, [hid_t, POINTER(c_ulonglong), POINTER(c_ulonglong)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_fields": (herr_t, [hid_t, c_size_t, c_size_t, c_size_t, c_size_t, c_size_t]),
        "H5Tset_size": (herr_t, [hid_t, c_size_t]),
        "H5Tset_ebias": (herr_t, [hid_t, c_size_t]),
        "H5Tget_class": (c_int, [hid_t]),
        "H5Tget_size": (c_size_t, [hid_t]),
        "H5Tget_sign": (c_int, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Oopen": (hid_t, [hid_t, c_char_p, hid_t]),
        "H5Oclose": (herr_t, [hid_t]),
        "H5Iget_type": (c_int, [hid_t]),
    }
    try:
        for name, (res, args) in proto.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        # HDF5 1.10 exports the un-versioned H5Ovisit; 1.12 and later only H5Ovisit1 / 2 / 3.  H5Ovisit1 has the 1.10
        # signature (the callback's info struct differs between versions, it is not read here).
        visit = getattr(h, "H5Ovisit1", None) or getattr(h, "H5Ovisit")
        visit.restype, visit.argtypes = herr_t, [hid_t, c_int, c_int, c_void_p, c_void_p]
        h.gfc_visit = visit
    except AttributeError as e:  # a library without one of the entry points: treated like no library at all
        raise Hdf5Unavailable(f"{path}: {e}") from e
    if h.H5open() < 0:
        raise Hdf5Unavailable("H5open failed")
    h.H5Eset_auto2(0, None, None)  # no error-stack printing: failures are reported through return codes below
    for key, sym in (("f4", "H5T_NATIVE_FLOAT_g"), ("f8", "H5T_NATIVE_DOUBLE_g"), ("i4", "H5T_NATIVE_INT32_g"),
                     ("i8", "H5T_NATIVE_INT64_g"), ("u1", "H5T_NATIVE_UINT8_g"), ("i1", "H5T_NATIVE_INT8_g"),
                     ("u4", "H5T_NATIVE_UINT32_g"), ("u8", "H5T_NATIVE_UINT64_g"), ("i2", "H5T_NATIVE_INT16_g"),
                     ("u2", "H5T_NATIVE_UINT16_g"), ("f4le", "H5T_IEEE_F32LE_g"), ("lcpl", "H5P_CLS_LINK_CREATE_ID_g")):
        try:
            _native[key] = hid_t.in_dll(h, sym).value
        except ValueError as e:  # symbol not exported by this build of the library
            raise Hdf5Unavailable(f"{path}: {e}") from e
    # IEEE binary16, the way h5py builds it: a 2-byte copy of F32LE with the half-precision bit fields
    f2 = h.H5Tcopy(_native["f4le"])
    ok = h.H5Tset_fields(f2, 15, 10, 5, 0, 10) >= 0 and h.H5Tset_size(f2, 2) >= 0 and h.H5Tset_ebias(f2, 15) >= 0
    if f2 < 0 or not ok:
        raise Hdf5Unavailable("cannot build the float16 datatype")
    _native["f2"] = f2
    _lib = h
    return h


def _chk(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def _type_of(a: np.ndarray):
    key = a.dtype.kind + str(a.dtype.itemsize)
    if a.dtype == np.bool_:
        return a.astype(np.uint8), _native["u1"]
    if key not in _native:
        raise TypeError(f"HDF5 export: unsupported dtype {a.dtype}")
    return a, _native[key]


def write_records(path, records: dict):
    """{name: {key: ndarray}} -> HDF5 file: group `name` (intermediate groups created, as h5py's create_group does for
    "seq/idx.ppm"), one contiguous dataset per key in its native dtype."""
    h = lib()
    f = _chk(h.H5Fcreate(str(path).encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
    lcpl = _chk(h.H5Pcreate(_native["lcpl"]), "H5Pcreate")
    try:
        _chk(h.H5Pset_create_intermediate_group(lcpl, 1), "H5Pset_create_intermediate_group")
        for name, rec in records.items():
            g = _chk(h.H5Gcreate2(f, str(name).encode(), lcpl, H5P_DEFAULT, H5P_DEFAULT), f"create group {name}")
            try:
                for key, value in rec.items():
                    a = np.asarray(value)
                    if not a.flags.c_contiguous:  # (np.ascontiguousarray would turn a 0-d value into 1-d)
                        a = np.ascontiguousarray(a)
                    a, tid = _type_of(a)
                    if a.ndim == 0:
                        sp = _chk(h.H5Screate(0), "H5Screate")
                    else:
                        dims = (c_ulonglong * a.ndim)(*a.shape)
                        sp = _chk(h.H5Screate_simple(a.ndim, dims, None), "H5Screate_simple")
                    d = _chk(h.H5Dcreate2(g, str(key).encode(), tid, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT),
                             f"create dataset {name}/{key}")
                    try:
                        if a.size:
                            _chk(h.H5Dwrite(d, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(c_void_p)),
                                 f"write {name}/{key}")
                    finally:
                        h.H5Dclose(d)
                        h.H5Sclose(sp)
            finally:
                h.H5Gclose(g)
    finally:
        h.H5Pclose(lcpl)
        _chk(h.H5Fclose(f), "close")


_VISIT = ctypes.CFUNCTYPE(herr_t, hid_t, c_char_p, c_void_p, c_void_p)


def read_records(path) -> dict:
    """HDF5 file -> {group path: {dataset name: ndarray}} for every dataset in the file."""
    h = lib()
    f = _chk(h.H5Fopen(str(path).encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    names = []

    def visit(obj, name, info, data):  # noqa: ARG001
        names.append(name.decode())
        return 0

    cb = _VISIT(visit)
    out = {}
    try:
        _chk(h.gfc_visit(f, H5_INDEX_NAME, H5_ITER_INC, ctypes.cast(cb, c_void_p), None), "H5Ovisit")
        for full in names:
            if full == "." or "/" not in full:
                continue
            o = h.H5Oopen(f, full.encode(), H5P_DEFAULT)
            if o < 0:
                continue
            kind = h.H5Iget_type(o)
            h.H5Oclose(o)
            if kind != H5I_DATASET:
                continue
            d = _chk(h.H5Dopen2(f, full.encode(), H5P_DEFAULT), f"open dataset {full}")
            sp, tp = h.H5Dget_space(d), h.H5Dget_type(d)
            try:
                nd = h.H5Sget_simple_extent_ndims(sp)
                dims = (c_ulonglong * max(nd, 1))()
                if nd > 0:
                    h.H5Sget_simple_extent_dims(sp, dims, None)
                shape = tuple(int(dims[i]) for i in range(nd))
                cls, size = h.H5Tget_class(tp), int(h.H5Tget_size(tp))
                if cls == H5T_FLOAT:
                    key = "f" + str(size)
                elif cls == H5T_INTEGER:
                    key = ("i" if h.H5Tget_sign(tp) else "u") + str(size)
                else:
                    raise TypeError(f"{full}: unsupported HDF5 type class {cls}")
                a = np.empty(shape, dtype=np.dtype(key))
                if a.size:
                    _chk(h.H5Dread(d, _native[key], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(c_void_p)),
                         f"read {full}")
            finally:
                h.H5Tclose(tp)
                h.H5Sclose(sp)
                h.H5Dclose(d)
            grp, key = full.rsplit("/", 1)
            out.setdefault(grp, {})[key] = a
    finally:
        h.H5Fclose(f)
    return out
