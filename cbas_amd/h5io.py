"""``_cls.h5`` reader/writer with the exact on-disk layout of the reference
(backend/cbas.py:413-421,437-440): dataset ``cls`` (N, D) IEEE half, chunks (8192, D), maxshape
(None, D), and the string attributes ``encoder_model_identifier`` / ``schema_version`` that the
project loader checks (backend/startup_page.py:100-117).

Backend: ``h5py`` when it is importable (it is a CBAS dependency); otherwise the HDF5 C library
is driven directly through ctypes (``libhdf5`` 1.10+).  Either way a real HDF5 file is produced —
there is no private format.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Optional

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import h5py as _h5py
except Exception:  # noqa: BLE001
    _h5py = None

CHUNK_ROWS = 8192          # backend/cbas.py:420
hid_t, hsize_t, herr_t = C.c_int64, C.c_uint64, C.c_int
_H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
_H5T_VARIABLE = C.c_size_t(-1).value


class _HDF5:
    """Just enough of the HDF5 1.10 C API."""
    _inst = None

    @classmethod
    def get(cls) -> "_HDF5":
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        cands = [os.environ.get("CBAS_HDF5_LIB"), ctypes.util.find_library("hdf5"),
                 ctypes.util.find_library("hdf5_serial")]
        for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                    "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
            cands += sorted(glob.glob(pat))
        lib = None
        for c in cands:
            if not c:
                continue
            try:
                lib = C.CDLL(c)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError("neither h5py nor libhdf5 is available: cannot read/write _cls.h5 files "
                               "(set CBAS_HDF5_LIB to the HDF5 shared library)")
        self.lib = L = lib
        if L.H5open() < 0:
            raise RuntimeError("H5open failed")

        def sig(name, res, *args):
            f = getattr(L, name)
            f.restype, f.argtypes = res, list(args)
            return f
        P = C.POINTER
        self.Fcreate = sig("H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
        self.Fopen = sig("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
        self.Fflush = sig("H5Fflush", herr_t, hid_t, C.c_int)
        self.Fclose = sig("H5Fclose", herr_t, hid_t)
        self.Screate_simple = sig("H5Screate_simple", hid_t, C.c_int, P(hsize_t), P(hsize_t))
        self.Screate = sig("H5Screate", hid_t, C.c_int)
        self.Sclose = sig("H5Sclose", herr_t, hid_t)
        self.Sselect_hyperslab = sig("H5Sselect_hyperslab", herr_t, hid_t, C.c_int, P(hsize_t), P(hsize_t),
                                     P(hsize_t), P(hsize_t))
        self.Sget_simple_extent_ndims = sig("H5Sget_simple_extent_ndims", C.c_int, hid_t)
        self.Sget_simple_extent_dims = sig("H5Sget_simple_extent_dims", C.c_int, hid_t, P(hsize_t), P(hsize_t))
        self.Pcreate = sig("H5Pcreate", hid_t, hid_t)
        self.Pset_chunk = sig("H5Pset_chunk", herr_t, hid_t, C.c_int, P(hsize_t))
        self.Pset_obj_track_times = sig("H5Pset_obj_track_times", herr_t, hid_t, C.c_uint)
        self.Pclose = sig("H5Pclose", herr_t, hid_t)
        self.Dcreate2 = sig("H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
        self.Dopen2 = sig("H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
        self.Dset_extent = sig("H5Dset_extent", herr_t, hid_t, P(hsize_t))
        self.Dget_space = sig("H5Dget_space", hid_t, hid_t)
        self.Dget_type = sig("H5Dget_type", hid_t, hid_t)
        self.Dwrite = sig("H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
        self.Dread = sig("H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
        self.Dclose = sig("H5Dclose", herr_t, hid_t)
        self.Tcopy = sig("H5Tcopy", hid_t, hid_t)
        self.Tset_fields = sig("H5Tset_fields", herr_t, hid_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t)
        self.Tset_size = sig("H5Tset_size", herr_t, hid_t, C.c_size_t)
        self.Tget_size = sig("H5Tget_size", C.c_size_t, hid_t)
        self.Tget_class = sig("H5Tget_class", C.c_int, hid_t)
        self.Tset_ebias = sig("H5Tset_ebias", herr_t, hid_t, C.c_size_t)
        self.Tset_cset = sig("H5Tset_cset", herr_t, hid_t, C.c_int)
        self.Tis_variable_str = sig("H5Tis_variable_str", C.c_int, hid_t)
        self.Tclose = sig("H5Tclose", herr_t, hid_t)
        self.Acreate2 = sig("H5Acreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t)
        self.Awrite = sig("H5Awrite", herr_t, hid_t, hid_t, C.c_void_p)
        self.Aexists = sig("H5Aexists", C.c_int, hid_t, C.c_char_p)
        self.Aopen = sig("H5Aopen", hid_t, hid_t, C.c_char_p, hid_t)
        self.Aget_type = sig("H5Aget_type", hid_t, hid_t)
        self.Aread = sig("H5Aread", herr_t, hid_t, hid_t, C.c_void_p)
        self.Aclose = sig("H5Aclose", herr_t, hid_t)
        self.Lexists = sig("H5Lexists", C.c_int, hid_t, C.c_char_p, hid_t)
        self.free_memory = sig("H5free_memory", herr_t, C.c_void_p)
        self.Eset_auto2 = sig("H5Eset_auto2", herr_t, hid_t, C.c_void_p, C.c_void_p)
        self.Eset_auto2(0, None, None)       # errors are reported through return codes below

        def glob_id(name):
            return hid_t.in_dll(L, name).value
        self.T_IEEE_F32LE = glob_id("H5T_IEEE_F32LE_g")
        self.T_IEEE_F64LE = glob_id("H5T_IEEE_F64LE_g")
        self.T_C_S1 = glob_id("H5T_C_S1_g")
        self.P_DATASET_CREATE = glob_id("H5P_CLS_DATASET_CREATE_ID_g")
        # IEEE binary16 exactly as h5py defines its 'f2': sign 15, exponent 10..14 (bias 15), mantissa 0..9
        t = self.Tcopy(self.T_IEEE_F32LE)
        ok = self.Tset_fields(t, 15, 10, 5, 0, 10) >= 0 and self.Tset_size(t, 2) >= 0 and self.Tset_ebias(t, 15) >= 0
        if t < 0 or not ok:
            raise RuntimeError("could not build the HDF5 float16 datatype")
        self.T_F16 = t

    def vlen_utf8(self):
        t = self.Tcopy(self.T_C_S1)
        self.Tset_size(t, _H5T_VARIABLE)
        self.Tset_cset(t, 1)             # H5T_CSET_UTF8
        return t


def _dims(*v):
    return (hsize_t * len(v))(*v)


class ClsWriter:
    """``with ClsWriter(tmp_path, dim, attrs) as w: w.append(rows_f16); w.flush()``"""

    def __init__(self, path: str, dim: int, attrs: Optional[Dict[str, str]] = None, dtype: str = "f2"):
        """``dtype``: "f2" is what the reference writes (backend/cbas.py:420); "f4" / "f8" exist to make the foreign
        files infer_file must also read (any numeric `cls` dataset, :507-508)."""
        self.path, self.dim, self.rows = path, int(dim), 0
        attrs = attrs or {}
        if dtype not in ("f2", "f4", "f8"):
            raise ValueError(f"dtype {dtype!r}: f2, f4 or f8")
        self._np = np.dtype(dtype)
        if _h5py is not None:
            self._f = _h5py.File(path, "w")
            for k, v in attrs.items():
                self._f.attrs[k] = v
            self._d = self._f.create_dataset("cls", shape=(0, dim), maxshape=(None, dim), dtype=dtype,
                                             chunks=(CHUNK_ROWS, dim))
            self._h = None
            return
        H = self._h = _HDF5.get()
        self._t = {"f2": H.T_F16, "f4": H.T_IEEE_F32LE, "f8": H.T_IEEE_F64LE}[dtype]
        self._fid = H.Fcreate(path.encode(), 2, 0, 0)            # H5F_ACC_TRUNC
        if self._fid < 0:
            raise OSError(f"cannot create HDF5 file {path}")
        for k, v in attrs.items():
            t, sp = H.vlen_utf8(), H.Screate(0)                  # H5S_SCALAR
            a = H.Acreate2(self._fid, k.encode(), t, sp, 0, 0)
            buf = C.c_char_p(str(v).encode("utf-8"))
            rc = H.Awrite(a, t, C.byref(buf)) if a >= 0 else -1
            H.Aclose(a); H.Sclose(sp); H.Tclose(t)
            if rc < 0:
                raise OSError(f"cannot write attribute {k} to {path}")
        sp = H.Screate_simple(2, _dims(0, dim), _dims(_H5S_UNLIMITED, dim))
        pl = H.Pcreate(H.P_DATASET_CREATE)
        H.Pset_chunk(pl, 2, _dims(CHUNK_ROWS, dim))
        # h5py creates datasets with track_times=False (what the reference's files look like): no creation / modification
        # timestamps in the object header, so equal rows give byte-identical files
        H.Pset_obj_track_times(pl, 0)
        self._did = H.Dcreate2(self._fid, b"cls", self._t, sp, 0, pl, 0)
        H.Pclose(pl); H.Sclose(sp)
        if self._did < 0:
            H.Fclose(self._fid)
            raise OSError(f"cannot create dataset 'cls' in {path}")

    def append(self, rows: np.ndarray) -> None:
        rows = np.ascontiguousarray(rows)
        if rows.dtype != self._np:
            rows = rows.astype(self._np)              # IEEE round-to-nearest-even, as h5py's f4 -> f2 write
        n = rows.shape[0]
        if n == 0:
            return
        assert rows.ndim == 2 and rows.shape[1] == self.dim
        if self._h is None:
            self._d.resize(self.rows + n, axis=0)
            self._d[-n:] = rows
        else:
            H = self._h
            if H.Dset_extent(self._did, _dims(self.rows + n, self.dim)) < 0:
                raise OSError("H5Dset_extent failed")
            fs = H.Dget_space(self._did)
            H.Sselect_hyperslab(fs, 0, _dims(self.rows, 0), None, _dims(n, self.dim), None)
            ms = H.Screate_simple(2, _dims(n, self.dim), None)
            rc = H.Dwrite(self._did, self._t, ms, fs, 0, rows.ctypes.data)
            H.Sclose(ms); H.Sclose(fs)
            if rc < 0:
                raise OSError("H5Dwrite failed")
        self.rows += n

    def flush(self) -> None:
        if self._h is None:
            self._f.flush()
        else:
            self._h.Fflush(self._fid, 1)             # H5F_SCOPE_GLOBAL

    def close(self) -> None:
        if self._h is None:
            if self._f is not None:
                self._f.close()
                self._f = None
        elif self._fid is not None:
            self._h.Dclose(self._did)
            self._h.Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class ClsReader:
    """Read access to a ``_cls.h5``: ``.shape``, ``.attrs`` (dict of str), ``read(a, b)`` -> rows [a, b) as float16 when
    the dataset is IEEE half (what encode_file writes; a byte copy), else as float32 (the reference reads any numeric
    dtype and converts with ``.float()``: backend/cbas.py:507-508)."""

    def __init__(self, path: str):
        self.path = path
        if _h5py is not None:
            self._f = _h5py.File(path, "r")
            self._d = self._f["cls"]
            self.shape = tuple(self._d.shape)
            self.itemsize = self._d.dtype.itemsize
            self.is_half = self._d.dtype == np.float16
            self.attrs = {k: (v.decode() if isinstance(v, bytes) else str(v)) for k, v in self._f.attrs.items()}
            self._h = None
            return
        H = self._h = _HDF5.get()
        self._fid = H.Fopen(path.encode(), 0, 0)
        if self._fid < 0:
            raise OSError(f"cannot open HDF5 file {path}")
        if H.Lexists(self._fid, b"cls", 0) <= 0:
            H.Fclose(self._fid)
            raise KeyError(f"{path} has no 'cls' dataset")
        self._did = H.Dopen2(self._fid, b"cls", 0)
        sp = H.Dget_space(self._did)
        nd = H.Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        H.Sget_simple_extent_dims(sp, dims, None)
        H.Sclose(sp)
        self.shape = tuple(int(x) for x in dims[:nd])
        t = H.Dget_type(self._did)
        self.itemsize = int(H.Tget_size(t))
        self.is_half = self.itemsize == 2 and H.Tget_class(t) == 1        # H5T_FLOAT
        H.Tclose(t)
        self.attrs = {}
        for k in ("encoder_model_identifier", "schema_version", "encoder_precision"):
            if H.Aexists(self._fid, k.encode()) > 0:
                self.attrs[k] = self._read_str_attr(k)

    def _read_str_attr(self, name: str) -> str:
        H = self._h
        a = H.Aopen(self._fid, name.encode(), 0)
        t = H.Aget_type(a)
        try:
            if H.Tis_variable_str(t) > 0:
                p = C.c_char_p()
                mt = H.vlen_utf8()
                H.Aread(a, mt, C.byref(p))
                H.Tclose(mt)
                s = p.value.decode("utf-8", "replace") if p.value is not None else ""
                if p.value is not None:
                    H.free_memory(C.cast(p, C.c_void_p))
                return s
            n = int(H.Tget_size(t))
            buf = C.create_string_buffer(n + 1)
            H.Aread(a, t, buf)
            return buf.raw[:n].split(b"\0")[0].decode("utf-8", "replace")
        finally:
            H.Tclose(t)
            H.Aclose(a)

    def read(self, start: int, stop: int) -> np.ndarray:
        start, stop = max(0, int(start)), min(int(stop), self.shape[0])
        n = max(0, stop - start)
        if self._h is None:
            a = self._d[start:stop]
            return a if a.dtype == np.float16 else a.astype(np.float32)
        out = np.empty((n, self.shape[1]), np.float16 if self.is_half else np.float32)
        if n == 0:
            return out
        H = self._h
        fs = H.Dget_space(self._did)
        H.Sselect_hyperslab(fs, 0, _dims(start, 0), None, _dims(n, self.shape[1]), None)
        ms = H.Screate_simple(2, _dims(n, self.shape[1]), None)
        # file f2 -> memory f2 is a byte copy; any other numeric file type is converted to float32 by the library
        rc = H.Dread(self._did, H.T_F16 if self.is_half else H.T_IEEE_F32LE, ms, fs, 0, out.ctypes.data)
        H.Sclose(ms); H.Sclose(fs)
        if rc < 0:
            raise OSError(f"H5Dread failed on {self.path}")
        return out

    def close(self) -> None:
        if self._h is None:
            if self._f is not None:
                self._f.close()
                self._f = None
        elif self._fid is not None:
            self._h.Dclose(self._did)
            self._h.Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def backend_name() -> str:
    return "h5py" if _h5py is not None else "libhdf5-ctypes"
