"""A small ctypes reader over the host's HDF5 C library — test infrastructure for the netCDF-4 container of the `writetonc`
sink (microclimf_amd/csrc/mcf_nc4file.hpp).  It goes through HDF5's own read path (H5Dread with type conversion to the
native int / double, the deflate filter of the library, the dimension-scale calls of its high-level library), so what the
tests see is what any HDF5-based reader — libnetcdf is one — gets from the file.  No netCDF library exists in the image."""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_ulonglong


def _open(names):
    for n in names:
        if not n:
            continue
        try:
            return C.CDLL(n, mode=C.RTLD_GLOBAL)
        except OSError:
            pass
    return None


def load():
    """(libhdf5, libhdf5_hl) or None when the host has none"""
    h = _open([os.environ.get("MCF_HDF5_LIB"), "libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so",
               "libhdf5_serial.so.103", "/opt/conda/lib/libhdf5.so", ctypes.util.find_library("hdf5")])
    hl = _open([os.environ.get("MCF_HDF5_HL_LIB"), "libhdf5_hl.so", "libhdf5_hl.so.100", "libhdf5_hl.so.200", "libhdf5_serial_hl.so",
                "libhdf5_serial_hl.so.100", "/opt/conda/lib/libhdf5_hl.so", ctypes.util.find_library("hdf5_hl")])
    if h is None or hl is None:
        return None
    h.H5open()
    for name, res, args in [
        ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]), ("H5Fclose", C.c_int, [hid_t]),
        ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", C.c_int, [hid_t]),
        ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]), ("H5Dget_create_plist", hid_t, [hid_t]),
        ("H5Dget_storage_size", hsize_t, [hid_t]),
        ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
        ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        ("H5Sclose", C.c_int, [hid_t]), ("H5Tclose", C.c_int, [hid_t]), ("H5Pclose", C.c_int, [hid_t]),
        ("H5Tequal", C.c_int, [hid_t, hid_t]), ("H5Tget_size", C.c_size_t, [hid_t]), ("H5Tget_class", C.c_int, [hid_t]),
        ("H5Pget_chunk", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]), ("H5Pget_nfilters", C.c_int, [hid_t]),
        ("H5Pget_filter2", C.c_int, [hid_t, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.c_uint),
                                     C.c_size_t, C.c_char_p, C.POINTER(C.c_uint)]),
        ("H5Pget_link_creation_order", C.c_int, [hid_t, C.POINTER(C.c_uint)]),
        ("H5Fget_create_plist", hid_t, [hid_t]),
        ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Aexists", C.c_int, [hid_t, C.c_char_p]),
        ("H5Aget_type", hid_t, [hid_t]), ("H5Aget_space", hid_t, [hid_t]),
        ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]), ("H5Aclose", C.c_int, [hid_t]),
        ("H5Lget_name_by_idx", C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        ("H5Gget_num_objs", C.c_int, [hid_t, C.POINTER(hsize_t)]),
    ]:
        f = getattr(h, name)
        f.restype, f.argtypes = res, args
    for name, res, args in [("H5DSis_scale", C.c_int, [hid_t]), ("H5DSis_attached", C.c_int, [hid_t, hid_t, C.c_uint]),
                            ("H5DSget_num_scales", C.c_int, [hid_t, C.c_uint]),
                            ("H5DSget_scale_name", C.c_ssize_t, [hid_t, C.c_char_p, C.c_size_t])]:
        f = getattr(hl, name)
        f.restype, f.argtypes = res, args
    return h, hl


def _gid(h, name):
    return hid_t.in_dll(h, name).value


class File:
    def __init__(self, path):
        libs = load()
        if libs is None:
            raise RuntimeError("no HDF5 library on this host")
        self.h, self.hl = libs
        self.f = self.h.H5Fopen(str(path).encode(), 0, 0)      # H5F_ACC_RDONLY
        if self.f < 0:
            raise OSError(f"H5Fopen failed for {path}")
        self._open = {}

    def close(self):
        for d in self._open.values():
            self.h.H5Dclose(d)
        self._open = {}
        if self.f >= 0:
            self.h.H5Fclose(self.f)
            self.f = -1

    def names_in_creation_order(self):
        n = hsize_t()
        self.h.H5Gget_num_objs(self.f, C.byref(n))
        out = []
        for i in range(n.value):
            b = C.create_string_buffer(256)
            # H5_INDEX_CRT_ORDER = 1, H5_ITER_INC = 0
            if self.h.H5Lget_name_by_idx(self.f, b".", 1, 0, i, b, 256, 0) < 0:
                raise OSError("the root group's links are not indexed by creation order")
            out.append(b.value.decode())
        return out

    def dset(self, name):
        if name not in self._open:
            d = self.h.H5Dopen2(self.f, name.encode(), 0)
            if d < 0:
                raise KeyError(name)
            self._open[name] = d
        return self._open[name]

    def shape(self, name):
        sp = self.h.H5Dget_space(self.dset(name))
        nd = self.h.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            self.h.H5Sget_simple_extent_dims(sp, dims, None)
        self.h.H5Sclose(sp)
        return tuple(dims[i] for i in range(nd))

    def is_type(self, name, type_global):
        t = self.h.H5Dget_type(self.dset(name))
        r = self.h.H5Tequal(t, _gid(self.h, type_global))
        self.h.H5Tclose(t)
        return r > 0

    def read(self, name):
        """the dataset through the library's conversion: int32 datasets as native int32, everything else as float64"""
        shp = self.shape(name)
        t = self.h.H5Dget_type(self.dset(name))
        is_int = self.h.H5Tget_class(t) == 0           # H5T_INTEGER
        self.h.H5Tclose(t)
        a = np.empty(shp, dtype=np.int32 if is_int else np.float64)
        mem = _gid(self.h, "H5T_NATIVE_INT_g" if is_int else "H5T_NATIVE_DOUBLE_g")
        if self.h.H5Dread(self.dset(name), mem, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError(f"H5Dread failed for {name}")
        return a

    def chunk_and_filters(self, name):
        p = self.h.H5Dget_create_plist(self.dset(name))
        nd = len(self.shape(name))
        ch = (hsize_t * max(nd, 1))()
        got = self.h.H5Pget_chunk(p, nd, ch)
        filters = []
        for i in range(max(self.h.H5Pget_nfilters(p), 0)):
            flags, ncd, cfg = C.c_uint(), C.c_size_t(4), C.c_uint()
            cd = (C.c_uint * 4)()
            fid = self.h.H5Pget_filter2(p, i, C.byref(flags), C.byref(ncd), cd, 0, None, C.byref(cfg))
            filters.append((fid, tuple(cd[j] for j in range(ncd.value))))
        self.h.H5Pclose(p)
        return (tuple(ch[i] for i in range(nd)) if got >= 0 else None), filters

    def storage_size(self, name):
        return int(self.h.H5Dget_storage_size(self.dset(name)))

    def attr(self, name, att):
        """text attributes as bytes, integer attributes as int (read as native int); name = None: the root group"""
        obj = self.f if name is None else self.dset(name)
        if self.h.H5Aexists(obj, att.encode()) <= 0:
            raise KeyError(f"{name}@{att}")
        a = self.h.H5Aopen(obj, att.encode(), 0)
        t = self.h.H5Aget_type(a)
        try:
            if self.h.H5Tget_class(t) == 3:              # H5T_STRING
                n = self.h.H5Tget_size(t)
                b = C.create_string_buffer(n + 1)
                if self.h.H5Aread(a, t, b) < 0:
                    raise OSError("H5Aread")
                return b.raw[:n].rstrip(b"\0")
            v = C.c_int()
            if self.h.H5Aread(a, _gid(self.h, "H5T_NATIVE_INT_g"), C.byref(v)) < 0:
                raise OSError("H5Aread")
            return v.value
        finally:
            self.h.H5Tclose(t)
            self.h.H5Aclose(a)

    def is_scale(self, name):
        return self.hl.H5DSis_scale(self.dset(name)) > 0

    def scale_name(self, name):
        b = C.create_string_buffer(256)
        self.hl.H5DSget_scale_name(self.dset(name), b, 256)
        return b.value.decode()

    def attached(self, var, scale, dim):
        return self.hl.H5DSis_attached(self.dset(var), self.dset(scale), dim) > 0

    def num_scales(self, var, dim):
        return self.hl.H5DSget_num_scales(self.dset(var), dim)
