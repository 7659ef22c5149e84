"""Marshalling of R-style arguments (named lists of column-major arrays) into the
flat C structs of include/mcf.h.

Mirrors what Rcpp does for the reference at src/microclimfCpp.cpp:2056-2111
(runmicro1Cpp) and :2344-2399 (runmicro2Cpp): lookup is BY NAME, year/month/day
are coerced to int, every array is read as column-major fp64 with the raster
row as the fastest index.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Mapping, Sequence

import numpy as np

from . import _abi


class Marshalled:
    """GridInputs/Options plus the numpy arrays that back their pointers."""

    def __init__(self):
        self.inputs = _abi.GridInputs()
        self.options = _abi.Options()
        self._keep = []
        self.rows = self.cols = self.tsteps = 0

    def _f64(self, a, shape=None, name="?"):
        arr = np.asarray(a, dtype=np.float64)
        if shape is not None and tuple(arr.shape) != tuple(shape):
            if arr.size != int(np.prod(shape)):
                raise ValueError(f"{name}: expected shape {tuple(shape)}, got {arr.shape}")
            arr = arr.reshape(shape, order="F")
        arr = np.asfortranarray(arr)
        self._keep.append(arr)
        return arr.ctypes.data_as(_abi.c_double_p)

    def _i32(self, a, n, name):
        arr = np.ascontiguousarray(np.asarray(a).astype(np.int32))
        if arr.shape != (n,):
            raise ValueError(f"{name}: expected length {n}")
        self._keep.append(arr)
        return arr.ctypes.data_as(_abi.c_int32_p)


def _get(d: Mapping, *names):
    for n in names:
        if n in d:
            return d[n]
    raise KeyError(f"none of {names} present (have {sorted(d)})")


def marshal(obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping,
            soilc: Mapping, reqhgt: float, zref: float, lat, lon, Sminp: float,
            Smaxp: float, tfact: float, complete: bool, mat: float,
            out: Sequence, array_forcing: bool, device: int = 0,
            days_per_chunk: int = 0, cells_per_block: int = 0, dfsel: Mapping | None = None,
            coarse: Mapping | None = None) -> Marshalled:
    """`coarse` = {"rowpos": [rows], "colpos": [cols]} switches to coarse array forcing (mcf.h, array_forcing == 2):
    climdata = {temp, relhum, pres, swdown, difrad, lwdown, windspeed, winddir} and pointm are then
    [coarse_rows, coarse_cols, tsteps] arrays."""
    m = Marshalled()
    hgt = np.asarray(vegp["hgt"], dtype=np.float64)
    if dfsel is None:
        if hgt.ndim != 2:
            raise ValueError("vegp$hgt must be a rows x cols matrix")
        L = 1
    else:                       # runmicro3Cpp/4Cpp: vegetation arrays are [rows, cols, layers]
        if hgt.ndim != 3:
            raise ValueError("vegp$hgt must be a rows x cols x layers array")
        # the reference walks dfsel's rows and indexes layer `row` of the arrays (cpp:2632, 2771): arrays may be
        # deeper than dfsel (`.runmodel3Cpp` renumbers the layers it uses, R/internal.R:1399) — the first L are read
        L = len(np.asarray(dfsel["st"]))
        if hgt.shape[2] < L:
            raise ValueError("dfsel has more rows than the vegetation arrays have layers")
    R, Cc = hgt.shape[:2]
    T = len(np.asarray(obstime["year"]))
    m.rows, m.cols, m.tsteps = R, Cc, T
    gi = m.inputs
    gi.rows, gi.cols, gi.tsteps = R, Cc, T
    gi.array_forcing = 2 if coarse is not None else 1 if array_forcing else 0
    cshape = None
    if coarse is not None:
        t0 = np.asarray(_get(climdata, "temp", "tc"))
        if t0.ndim != 3 or t0.shape[2] != T:
            raise ValueError("coarse climate arrays must be [coarse_rows, coarse_cols, tsteps]")
        cshape = t0.shape
        gi.coarse_rows, gi.coarse_cols = int(cshape[0]), int(cshape[1])
        gi.coarse_rowpos = m._f64(coarse["rowpos"], (R,), "coarse$rowpos")
        gi.coarse_colpos = m._f64(coarse["colpos"], (Cc,), "coarse$colpos")
        gi.coarse_relhum = m._f64(climdata["relhum"], cshape, "climdata$relhum")
        gi.coarse_winddir = m._f64(climdata["winddir"], cshape, "climdata$winddir")
        gi.coarse_altcorrect = int(coarse.get("altcorrect", 0))
        if gi.coarse_altcorrect:
            gi.coarse_dtm = m._f64(coarse["dtmc"], cshape[:2], "coarse$dtmc")
            gi.fine_dtm = m._f64(coarse["dtm"], (R, Cc), "coarse$dtm")
    gi.obstime.year = m._i32(obstime["year"], T, "obstime$year")
    gi.obstime.month = m._i32(obstime["month"], T, "obstime$month")
    gi.obstime.day = m._i32(obstime["day"], T, "obstime$day")
    gi.obstime.hour = m._f64(obstime["hour"], (T,), "obstime$hour")
    fshape = cshape if coarse is not None else (R, Cc, T) if array_forcing else (T,)
    # climdata: data.frame names (1Cpp) or list names (2Cpp)
    names = {"tc": ("temp", "tc"), "pk": ("pres", "pk")}
    for f in _abi.CLIM_FIELDS:
        if coarse is not None and f in ("es", "ea", "tdew", "winddir"):     # derived inside the solver
            setattr(gi.clim, f, None)
            continue
        src = _get(climdata, *names.get(f, (f,)))
        shape = (T,) if f == "winddir" else fshape
        setattr(gi.clim, f, m._f64(src, shape, f"climdata${f}"))
    need_tgp = (reqhgt < 0) and (not complete)
    for f in _abi.POINTM_FIELDS:
        if f in ("Tg", "Tbp") and not need_tgp:
            setattr(gi.pointm, f, None)
            continue
        src = _get(pointm, *(("G", "Gp") if f == "G" else (f,)))
        if f == "Tbp" and np.ndim(src) == 0:       # `pointm$Tbp <- 0` (R/internal.R:1096)
            src = np.full(fshape, float(src))
        setattr(gi.pointm, f, m._f64(src, fshape, f"pointm${f}"))
    for f in _abi.VEGP_FIELDS:
        src = vegp[f]
        if dfsel is not None:
            src = np.asfortranarray(np.asarray(src, dtype=np.float64))
            if src.ndim != 3 or src.shape[2] < L:
                raise ValueError(f"vegp${f}: expected [rows, cols, >= {L} layers]")
            src = src[:, :, :L]            # a contiguous prefix in column-major order: no copy
        setattr(gi.vegp, f, m._f64(src, (R, Cc) if dfsel is None else (R, Cc, L), f"vegp${f}"))
    if dfsel is None:
        gi.veg_layers = 0
        gi.lyr_st = gi.lyr_ed = None
    else:
        gi.veg_layers = L
        gi.lyr_st = m._i32(dfsel["st"], L, "dfsel$st")
        gi.lyr_ed = m._i32(dfsel["ed"], L, "dfsel$ed")
    for f in _abi.SOILC_FIELDS:
        shape = (R, Cc, 8) if f == "wsa" else (R, Cc, 24) if f == "hor" else (R, Cc)
        setattr(gi.soilc, f, m._f64(soilc[f], shape, f"soilc${f}"))
    if array_forcing:
        gi.lats = m._f64(lat, (R, Cc), "lats")
        gi.lons = m._f64(lon, (R, Cc), "lons")
        gi.lat = gi.lon = float("nan")
    else:
        gi.lat, gi.lon = float(lat), float(lon)
        gi.lats = gi.lons = None
    op = m.options
    op.reqhgt, op.zref = float(reqhgt), float(zref)
    op.Sminp, op.Smaxp = float(Sminp), float(Smaxp)
    op.tfact, op.mat = float(tfact), float(mat)
    op.complete = 1 if complete else 0
    out = list(out)
    if len(out) != _abi.NOUT:
        raise ValueError("out must have 10 entries")
    for v in range(_abi.NOUT):
        op.out[v] = 1 if out[v] else 0     # may arrive as numeric 0/1 (R/internal.R:1161)
    op.device = device
    op.days_per_chunk = days_per_chunk
    op.cells_per_block = cells_per_block
    return m


def alloc_outputs(m: Marshalled):
    """Host output arrays [rows, cols, tsteps] (column-major) for requested vars."""
    outs = _abi.Outputs()
    arrays = {}
    for v, name in enumerate(_abi.OUT_NAMES):
        if m.options.out[v]:
            if os.environ.get("MCF_TEST_PLAIN_OUTPUTS"):
                # (measurement aid, tools/oneshot_rate.py: an anonymous mapping without numpy's own huge-page advice —
                # what an R vector's malloc gives the library to write into)
                import mmap
                buf = mmap.mmap(-1, m.rows * m.cols * m.tsteps * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
                a = np.frombuffer(buf, dtype=np.float64).reshape((m.rows, m.cols, m.tsteps), order="F")
            else:
                a = np.empty((m.rows, m.cols, m.tsteps), dtype=np.float64, order="F")
            arrays[name] = a
            outs.var[v] = a.ctypes.data_as(_abi.c_double_p)
        else:
            outs.var[v] = None
    return outs, arrays
