"""Test-only bridge to the REAL HDF5 library (libhdf5.so from the image's conda tree, via ctypes) --
the independent implementation that pins icsg3d_amd/hdf5_min.py: it writes Keras-layout files the way
h5py does (tests/golden/make_h5_golden.py, round-trip tests) and reads back what our writer produced.
Never imported by the product package."""
import ctypes as C
import glob
import os

import numpy as np

_CANDIDATES = ["/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"]


def find_lib():
    for p in _CANDIDATES + sorted(glob.glob("/opt/conda/lib/libhdf5.so.*")):
        if os.path.exists(p):
            return p
    return None


class H5:
    def __init__(self):
        path = find_lib()
        if path is None:
            raise OSError("libhdf5 not found")
        self.l = l = C.CDLL(path)
        hid = C.c_int64
        for name, res, args in [
            ("H5open", C.c_int, []),
            ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
            ("H5Fclose", C.c_int, [hid]),
            ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Gclose", C.c_int, [hid]),
            ("H5Oopen", hid, [hid, C.c_char_p, hid]), ("H5Oclose", C.c_int, [hid]),
            ("H5Screate", hid, [C.c_int]), ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]),
            ("H5Sclose", C.c_int, [hid]), ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(C.c_uint64), C.c_void_p]),
            ("H5Tcopy", hid, [hid]), ("H5Tset_size", C.c_int, [hid, C.c_size_t]), ("H5Tset_strpad", C.c_int, [hid, C.c_int]),
            ("H5Tget_size", C.c_size_t, [hid]), ("H5Tget_class", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]),
            ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", C.c_int, [hid, hid, C.c_void_p]),
            ("H5Aopen", hid, [hid, C.c_char_p, hid]), ("H5Aread", C.c_int, [hid, hid, C.c_void_p]),
            ("H5Aget_type", hid, [hid]), ("H5Aget_space", hid, [hid]), ("H5Aclose", C.c_int, [hid]),
            ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
            ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
            ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
            ("H5Dget_space", hid, [hid]), ("H5Dget_type", hid, [hid]), ("H5Dclose", C.c_int, [hid]),
            ("H5Pcreate", hid, [hid]), ("H5Pset_create_intermediate_group", C.c_int, [hid, C.c_uint]),
            ("H5Pset_chunk", C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]), ("H5Pclose", C.c_int, [hid]),
            ("H5Eset_auto2", C.c_int, [hid, C.c_void_p, C.c_void_p]),
        ]:
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        l.H5open()
        l.H5Eset_auto2(0, None, None)      # no error stack printing; callers check return values
        g = lambda n: C.c_int64.in_dll(l, n).value
        self.F32, self.C_S1 = g("H5T_IEEE_F32LE_g"), g("H5T_C_S1_g")
        self.NATIVE_FLOAT = g("H5T_NATIVE_FLOAT_g")
        self.LCPL, self.DCPL = g("H5P_CLS_LINK_CREATE_ID_g"), g("H5P_CLS_DATASET_CREATE_ID_g")

    def _ok(self, v, what):
        if v < 0:
            raise RuntimeError("libhdf5: %s failed" % what)
        return v

    # ---- writing, the way h5py does it for Keras
    def _str_attr(self, loc, name, value):
        l = self.l
        if isinstance(value, (bytes, bytearray)):           # scalar fixed-length string (f.attrs[k] = b"...")
            t = l.H5Tcopy(self.C_S1); l.H5Tset_size(t, max(len(value), 1)); l.H5Tset_strpad(t, 1)
            s = l.H5Screate(0)
            buf = C.create_string_buffer(bytes(value), max(len(value), 1))
        else:                                               # 1-D array of fixed-length strings (numpy 'S')
            arr = np.asarray(value, dtype="S")
            if arr.dtype.itemsize == 0:
                arr = arr.astype("S1")
            t = l.H5Tcopy(self.C_S1); l.H5Tset_size(t, arr.dtype.itemsize); l.H5Tset_strpad(t, 1)
            dims = (C.c_uint64 * 1)(arr.shape[0])
            s = l.H5Screate_simple(1, dims, None)
            buf = C.create_string_buffer(arr.tobytes(), max(arr.nbytes, 1))
        a = self._ok(l.H5Acreate2(loc, name.encode(), t, s, 0, 0), "H5Acreate2 " + name)
        self._ok(l.H5Awrite(a, t, buf), "H5Awrite")
        l.H5Aclose(a); l.H5Sclose(s); l.H5Tclose(t)

    def write_keras(self, path, layers, full_model=False, chunked=False, extra_root_attrs=None):
        """layers: [(layer_name, [(weight_name, float32 array), ...])] -- mirrors keras.engine.saving
        save_weights_to_hdf5_group (layer_names / weight_names attributes, datasets named by weight name with
        intermediate groups) as executed by h5py on libhdf5."""
        l = self.l
        f = self._ok(l.H5Fcreate(path.encode(), 2, 0, 0), "H5Fcreate")
        g = f
        if full_model:
            g = self._ok(l.H5Gcreate2(f, b"model_weights", 0, 0, 0), "H5Gcreate2")
            self._str_attr(f, "keras_version", b"2.3.1"); self._str_attr(f, "backend", b"tensorflow")
            self._str_attr(f, "model_config", b'{"class_name": "Model"}')
        for k, v in (extra_root_attrs or {}).items():
            self._str_attr(f, k, v)
        self._str_attr(g, "layer_names", [ln.encode() for ln, _ in layers])
        self._str_attr(g, "backend", b"tensorflow"); self._str_attr(g, "keras_version", b"2.3.1")
        lcpl = l.H5Pcreate(self.LCPL)
        l.H5Pset_create_intermediate_group(lcpl, 1)
        for ln, ws in layers:
            lg = self._ok(l.H5Gcreate2(g, ln.encode(), 0, 0, 0), "H5Gcreate2 " + ln)
            self._str_attr(lg, "weight_names", [wn.encode() for wn, _ in ws])
            for wn, arr in ws:
                arr = np.ascontiguousarray(arr, np.float32)
                if arr.ndim:
                    dims = (C.c_uint64 * arr.ndim)(*arr.shape)
                    s = l.H5Screate_simple(arr.ndim, dims, None)
                else:
                    s = l.H5Screate(0)
                dcpl = 0
                if chunked and arr.ndim:
                    dcpl = l.H5Pcreate(self.DCPL)
                    cd = (C.c_uint64 * arr.ndim)(*[max(1, (n + 1) // 2) for n in arr.shape])
                    l.H5Pset_chunk(dcpl, arr.ndim, cd)
                d = self._ok(l.H5Dcreate2(lg, wn.encode(), self.F32, s, lcpl, dcpl, 0), "H5Dcreate2 " + wn)
                self._ok(l.H5Dwrite(d, self.NATIVE_FLOAT, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite")
                l.H5Dclose(d); l.H5Sclose(s)
                if dcpl:
                    l.H5Pclose(dcpl)
            l.H5Gclose(lg)
        l.H5Pclose(lcpl)
        if full_model:
            l.H5Gclose(g)
        l.H5Fclose(f)

    # ---- reading (validates files written by icsg3d_amd/hdf5_min.py)
    def read_dataset(self, path, name):
        l = self.l
        f = self._ok(l.H5Fopen(path.encode(), 0, 0), "H5Fopen")
        d = self._ok(l.H5Dopen2(f, name.encode(), 0), "H5Dopen2 " + name)
        s = l.H5Dget_space(d)
        nd = l.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        l.H5Sget_simple_extent_dims(s, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        out = np.empty(shape, np.float32)
        self._ok(l.H5Dread(d, self.NATIVE_FLOAT, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
        l.H5Sclose(s); l.H5Dclose(d); l.H5Fclose(f)
        return out

    def read_str_attr(self, path, obj, name):
        """fixed-length string attribute (scalar or 1-D) of object `obj` -> list of bytes"""
        l = self.l
        f = self._ok(l.H5Fopen(path.encode(), 0, 0), "H5Fopen")
        o = self._ok(l.H5Oopen(f, obj.encode(), 0), "H5Oopen " + obj)
        a = self._ok(l.H5Aopen(o, name.encode(), 0), "H5Aopen " + name)
        t, s = l.H5Aget_type(a), l.H5Aget_space(a)
        size = l.H5Tget_size(t)
        nd = l.H5Sget_simple_extent_ndims(s)
        dims = (C.c_uint64 * max(nd, 1))()
        l.H5Sget_simple_extent_dims(s, dims, None)
        n = int(dims[0]) if nd else 1
        buf = C.create_string_buffer(size * n)
        self._ok(l.H5Aread(a, t, buf), "H5Aread")
        l.H5Tclose(t); l.H5Sclose(s); l.H5Aclose(a); l.H5Oclose(o); l.H5Fclose(f)
        raw = buf.raw
        return [raw[i * size:(i + 1) * size].rstrip(b"\0") for i in range(n)]
