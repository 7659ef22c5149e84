"""Host mirror of the reference's `writetonc` (R/dataprep.R:1063-1260): the solver's outputs as a netCDF file of
int32 variables on (east, north, time).  `NcWriter` is the streaming form (day chunks straight from a `Plan`'s device
ring, packed and byte-ordered on the GPU); `writetonc` has the reference's call shape for a finished `mout`.

Two containers (`format=`): "classic" — netCDF classic / 64-bit offsets, `time` as record dimension, uncompressed, needs
nothing on the host (mcf_ncfile.hpp) — and "netcdf4", the reference's own (HDF5 by the netCDF-4 conventions, chunked,
deflate 9 unless `deflate_level` says otherwise; needs an HDF5 library on the host at run time, mcf_nc4file.hpp).
"""
from __future__ import annotations

import calendar
import ctypes as C
from typing import Mapping, Sequence

import numpy as np

from . import _abi

# writetonc's default `vars` per height (dataprep.R:1108, 1179, 1232)
DEFAULT_VARS_ABOVE = ("Tz", "tleaf", "relhum", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
DEFAULT_VARS_SURFACE = ("Tz", "soilm", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
DEFAULT_VARS_BELOW = ("Tz", "soilm")


FORMATS = {"classic": 0, "netcdf4": 1}      # MCF_NC_CLASSIC, MCF_NC_NETCDF4


def default_vars(reqhgt: float) -> tuple:
    return DEFAULT_VARS_ABOVE if reqhgt > 0 else DEFAULT_VARS_SURFACE if reqhgt == 0 else DEFAULT_VARS_BELOW


def coords_from_extent(xmin, xmax, ymin, ymax, xres, yres=None):
    """`est` / `nth` of dataprep.R:1072-1073: cell centres, both ASCENDING (R's seq(from, to, by))."""
    yres = xres if yres is None else yres
    ne = int(np.floor((xmax - xres / 2 - (xmin + xres / 2)) / xres + 1e-10)) + 1
    nn = int(np.floor((ymax - yres / 2 - (ymin + yres / 2)) / yres + 1e-10)) + 1
    return xmin + xres / 2 + xres * np.arange(ne), ymin + yres / 2 + yres * np.arange(nn)


def hours_since_epoch(obstime: Mapping) -> np.ndarray:
    """as.numeric(as.POSIXct(tme)) / 3600 for a UTC obstime table (year, month, day, hour)."""
    y, m, d = (np.asarray(obstime[k]).astype(int) for k in ("year", "month", "day"))
    h = np.asarray(obstime["hour"], dtype=np.float64)
    return np.array([calendar.timegm((yy, mm, dd, 0, 0, 0)) / 3600.0 for yy, mm, dd in zip(y, m, d)]) + h


class NcWriter:
    def __init__(self, fileout: str, rows: int, cols: int, time_hours, east, north, reqhgt: float,
                 vars: Sequence[str] | None = None, crs_wkt: str = "", reference_puts_only: bool = False,
                 format: str = "classic", deflate_level: int = 0):
        self._lib = _abi.load()
        self.vars = tuple(default_vars(reqhgt) if vars is None else vars)
        unknown = [v for v in self.vars if v not in _abi.OUT_NAMES]
        if unknown:
            raise ValueError(f"unknown variables {unknown}")
        self._t = np.ascontiguousarray(time_hours, dtype=np.float64)
        self._e = np.ascontiguousarray(east, dtype=np.float64)
        self._n = np.ascontiguousarray(north, dtype=np.float64)
        if self._e.shape != (cols,) or self._n.shape != (rows,):
            raise ValueError("east / north must have one entry per raster column / row")
        sp = _abi.NcSpec()
        sp.rows, sp.cols, sp.nsteps = rows, cols, len(self._t)
        sp.east = self._e.ctypes.data_as(_abi.c_double_p)
        sp.north = self._n.ctypes.data_as(_abi.c_double_p)
        sp.time_hours = self._t.ctypes.data_as(_abi.c_double_p)
        self._wkt = crs_wkt.encode()
        sp.crs_wkt = self._wkt
        sp.reqhgt = float(reqhgt)
        for v in self.vars:
            sp.vars[_abi.OUT_NAMES.index(v)] = 1
        sp.reference_puts_only = 1 if reference_puts_only else 0
        if format not in FORMATS:
            raise ValueError(f"format must be one of {sorted(FORMATS)}")
        sp.format = FORMATS[format]
        sp.deflate_level = int(deflate_level)       # netcdf4: 0 = writetonc's compression = 9, -1 = none
        self.rows, self.cols, self.nsteps = rows, cols, len(self._t)
        self._h = C.c_void_p()
        _abi.check(self._lib.mcf_nc_create(str(fileout).encode(), C.byref(sp), C.byref(self._h)))

    def write_host(self, step0: int, arrays: Mapping[str, np.ndarray]):
        """arrays[name]: [rows, cols, n] as runmicro*Cpp return them"""
        ptrs = (_abi.c_double_p * 10)()
        keep, n = [], None
        for name in self.vars:
            if name not in arrays:
                continue
            a = np.asfortranarray(arrays[name], dtype=np.float64)
            if a.shape[:2] != (self.rows, self.cols) or (n is not None and a.shape[2] != n):
                raise ValueError(f"{name}: expected [rows, cols, n]")
            n = a.shape[2]
            keep.append(a)
            ptrs[_abi.OUT_NAMES.index(name)] = a.ctypes.data_as(_abi.c_double_p)
        if n is None:
            raise ValueError("no variable of the file given")
        _abi.check(self._lib.mcf_nc_write_host(self._h, int(step0), int(n), C.byref(ptrs)))

    def write_plan(self, plan, slot: int, slot_step0: int, file_step0: int, nsteps: int, timing: bool = False):
        ms = C.c_float()
        _abi.check(self._lib.mcf_nc_write_plan(self._h, plan._p, int(slot), int(slot_step0), int(file_step0), int(nsteps),
                                               C.byref(ms) if timing else None))
        return ms.value if timing else None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            h, self._h = self._h, C.c_void_p()
            _abi.check(self._lib.mcf_nc_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def writetonc(mout: Mapping, fileout: str, dtm: Mapping, reqhgt: float, vars: Sequence[str] | None = None,
              reference_puts_only: bool = False, format: str = "classic", deflate_level: int = 0):
    """`writetonc(mout, fileout, dtm, reqhgt, vars)`: `mout` holds the output arrays and `tme` (an obstime table or
    hours since 1970); `dtm` = {"xmin","xmax","ymin","ymax","res", optional "crs"} stands for the SpatRaster."""
    names = tuple(default_vars(reqhgt) if vars is None else vars)
    first = np.asarray(mout[next(v for v in names if v in mout)])
    rows, cols = first.shape[:2]
    tme = mout["tme"]
    hours = hours_since_epoch(tme) if isinstance(tme, Mapping) else np.asarray(tme, dtype=np.float64)
    res = dtm["res"]
    xres, yres = (res, res) if np.isscalar(res) else res
    east, north = coords_from_extent(dtm["xmin"], dtm["xmax"], dtm["ymin"], dtm["ymax"], xres, yres)
    with NcWriter(fileout, rows, cols, hours, east, north, reqhgt, names, dtm.get("crs", ""), reference_puts_only,
                  format=format, deflate_level=deflate_level) as w:
        w.write_host(0, {k: mout[k] for k in names if k in mout})
