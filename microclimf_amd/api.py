"""Host-side mirror of the reference's operator interface for the grid solver.

`runmicro1Cpp` / `runmicro2Cpp` take the same 15 positional arguments, with the
same names and meaning, as the R functions of the reference
(R/RcppExports.R:72-78) that `.runmodel1Cpp` / `.runmodel2Cpp` call
(R/internal.R:1168, 1342): R named lists / data.frames become Python mappings
of numpy arrays, R's column-major arrays become Fortran-ordered numpy arrays,
and the returned named list becomes a dict holding only the requested
variables, in the reference's order (src/microclimfCpp.cpp:2326-2335), each of
shape (rows, cols, tsteps).

All arithmetic happens in libmcfhip.so (hand-written HIP, gfx950); this module
only marshals pointers.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Sequence

import numpy as np

from . import _abi
from .marshal import Marshalled, alloc_outputs, marshal


def _run(fn_name, array_forcing, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
         Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, cells_per_block, dfsel=None, coarse=None,
         devices=None, n_blocks=0):
    lib = _abi.load()
    m = marshal(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp,
                tfact, complete, mat, out, array_forcing, device, days_per_chunk, cells_per_block, dfsel, coarse)
    outs, arrays = alloc_outputs(m)
    if devices is not None or n_blocks:
        # one process, several devices (include/mcf.h mcf_runmicro1_multi): row blocks dealt to the listed devices
        mu = _abi.Multi()
        devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
        mu.n_devices, mu.devices, mu.n_blocks = int(devs.size), devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
        _abi.check(getattr(lib, fn_name + "_multi")(C.byref(m.inputs), C.byref(m.options), C.byref(mu), C.byref(outs)))
        return arrays
    _abi.check(getattr(lib, fn_name)(C.byref(m.inputs), C.byref(m.options), C.byref(outs)))
    return arrays


def runmicro1Cpp(obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping, soilc: Mapping,
                 reqhgt: float, zref: float, lat: float, lon: float, Sminp: float, Smaxp: float,
                 tfact: float, complete: bool, mat: float, out: Sequence, *, device: int = 0,
                 days_per_chunk: int = 0, cells_per_block: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Grid microclimate model, hourly, static vegetation, data.frame (vector) climate.

    Drop-in for the reference's runmicro1Cpp (src/microclimfCpp.cpp:2052-2337).  `devices` (a list of HIP ordinals, [] =
    all visible) / `n_blocks`: the raster in row blocks over several devices from this one process, same bits."""
    return _run("mcf_runmicro1", False, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, cells_per_block, devices=devices, n_blocks=n_blocks)


def runmicro2Cpp(obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping, soilc: Mapping,
                 reqhgt: float, zref: float, lats, lons, Sminp: float, Smaxp: float, tfact: float,
                 complete: bool, mat: float, out: Sequence, *, device: int = 0,
                 days_per_chunk: int = 0, cells_per_block: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Grid microclimate model, hourly, static vegetation, array climate inputs.

    Drop-in for the reference's runmicro2Cpp (src/microclimfCpp.cpp:2340-2621); `devices` / `n_blocks` as runmicro1Cpp."""
    return _run("mcf_runmicro2", True, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, cells_per_block, devices=devices, n_blocks=n_blocks)


def coarse_positions(n_fine: int, n_coarse: int):
    """Position of each of `n_fine` equally spaced fine cells in units of `n_coarse` coarse cells covering the same
    extent (0 = centre of the first coarse cell), clamped to the coarse centres (edge replication)."""
    pos = (np.arange(n_fine) + 0.5) * (n_coarse / n_fine) - 0.5
    return np.clip(pos, 0.0, n_coarse - 1.0)


def runmicro2Cpp_coarse(obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping, soilc: Mapping,
                        reqhgt: float, zref: float, lats, lons, Sminp: float, Smaxp: float, tfact: float,
                        complete: bool, mat: float, out: Sequence, *, rowpos=None, colpos=None, altcorrect: int = 0,
                        dtmc=None, dtm=None, device: int = 0, days_per_chunk: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """`.runmodel2Cpp` with the resampling fused into the solver (include/mcf.h, array_forcing == 2): `climdata` =
    {temp, relhum, pres, swdown, difrad, lwdown, windspeed, winddir} and `pointm` = {soilm, Gp, umu, kp, muGp, dtrp} as
    COARSE arrays [coarse_rows, coarse_cols, tsteps] — what `.cca(..., dtmc, dtmc)` gives before `resample` — instead of
    the full-resolution arrays runmicro2Cpp takes.  `rowpos` / `colpos`: see `coarse_positions` (default: the coarse
    grid covers the raster's extent)."""
    R, Cc = np.shape(vegp["hgt"])[:2]
    cr, cc = np.shape(climdata["temp"])[:2]
    coarse = {"rowpos": coarse_positions(R, cr) if rowpos is None else rowpos,
              "colpos": coarse_positions(Cc, cc) if colpos is None else colpos}
    if altcorrect:                     # `.runmodel2Cpp`'s altcorrect 1 / 2 with the coarse (dtmc) and fine (dtm) elevations
        coarse.update(altcorrect=int(altcorrect), dtmc=dtmc, dtm=dtm)
    return _run("mcf_runmicro2", True, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, 0, None, coarse, devices=devices, n_blocks=n_blocks)


def runmicro3Cpp(dfsel: Mapping, obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping,
                 soilc: Mapping, reqhgt: float, zref: float, lat: float, lon: float, Sminp: float,
                 Smaxp: float, tfact: float, complete: bool, mat: float, out: Sequence, *, device: int = 0,
                 days_per_chunk: int = 0, cells_per_block: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Hourly, changing vegetation, data.frame climate: drop-in for the reference's runmicro3Cpp
    (src/microclimfCpp.cpp:2624-2924).  `dfsel` has columns lyr, st, ed (0-based step ranges of
    each vegetation layer, R/internal.R:1391-1399); vegp entries are [rows, cols, layers].  `devices` / `n_blocks`: as runmicro1Cpp."""
    return _run("mcf_runmicro3", False, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, cells_per_block, dfsel,
                devices=devices, n_blocks=n_blocks)


def runmicro4Cpp(dfsel: Mapping, obstime: Mapping, climdata: Mapping, pointm: Mapping, vegp: Mapping,
                 soilc: Mapping, reqhgt: float, zref: float, lats, lons, Sminp: float, Smaxp: float,
                 tfact: float, complete: bool, mat: float, out: Sequence, *, device: int = 0,
                 days_per_chunk: int = 0, cells_per_block: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Hourly, changing vegetation, array climate: drop-in for the reference's runmicro4Cpp
    (src/microclimfCpp.cpp:2926-3226).  `devices` / `n_blocks`: as runmicro1Cpp."""
    return _run("mcf_runmicro4", True, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                Sminp, Smaxp, tfact, complete, mat, out, device, days_per_chunk, cells_per_block, dfsel,
                devices=devices, n_blocks=n_blocks)


BIOCLIM_DFSEL = {"lyr": np.arange(1, 15), "st": np.arange(14) * 24, "ed": np.arange(14) * 24 + 23}   # cpp:3634-3646


def _bioclim(fn_name, array_forcing, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp,
             Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air, device, layered=False, devices=None, n_blocks=0):
    lib = _abi.load()
    m = marshal(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, True, mat,
                [1] * 10, array_forcing, device, dfsel=BIOCLIM_DFSEL if layered else None)
    sel = _abi.BioclimSel()
    keep = []
    for name, q in (("wet", wetq), ("dry", dryq), ("hot", hotq), ("col", colq)):
        arr = np.ascontiguousarray(np.asarray(q).astype(np.int32))
        keep.append(arr)
        setattr(sel, name + "q", arr.ctypes.data_as(_abi.c_int32_p))
        setattr(sel, "n" + name, arr.size)
    sel.air = 1 if air else 0
    out = list(out)
    if len(out) != _abi.NBIO:
        raise ValueError("out must have 19 entries")
    bo = _abi.BioclimOut()
    res = {}
    for v in range(_abi.NBIO):
        sel.out[v] = 1 if out[v] else 0
        if out[v]:
            a = np.empty((m.rows, m.cols), dtype=np.float64, order="F")
            res[f"bio{v + 1}"] = a
            bo.bio[v] = a.ctypes.data_as(_abi.c_double_p)
        else:
            bo.bio[v] = None
    if devices is not None or n_blocks:
        # one process, several devices (include/mcf.h mcf_runbioclim1_multi): row blocks dealt to the listed devices, same bits
        mu = _abi.Multi()
        devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
        mu.n_devices, mu.devices, mu.n_blocks = int(devs.size), devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
        _abi.check(getattr(lib, fn_name + "_multi")(C.byref(m.inputs), C.byref(m.options), C.byref(sel), C.byref(mu), C.byref(bo)))
    else:
        _abi.check(getattr(lib, fn_name)(C.byref(m.inputs), C.byref(m.options), C.byref(sel), C.byref(bo)))
    return res


def runbioclim1Cpp(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, mat, out,
                   wetq, dryq, hotq, colq, air, *, device: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Drop-in for the reference's runbioclim1Cpp (src/microclimfCpp.cpp:3563-3588): the grid solver on
    the selected days followed by the 19 per-cell bioclim reductions, both on the device; only the
    requested [rows, cols] matrices are copied back.  `devices` / `n_blocks`: row blocks over several devices, same bits."""
    return _bioclim("mcf_runbioclim1", False, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                    Sminp, Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air, device, devices=devices, n_blocks=n_blocks)


def runbioclim2Cpp(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, mat, out,
                   wetq, dryq, hotq, colq, air, *, device: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Drop-in for the reference's runbioclim2Cpp (src/microclimfCpp.cpp:3590-3616), array climate."""
    return _bioclim("mcf_runbioclim2", True, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                    Sminp, Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air, device, devices=devices, n_blocks=n_blocks)


class Plan:
    """HBM-resident solver plan (include/mcf.h plan API): inputs uploaded once,
    day chunks solved into a device output ring."""

    def __init__(self, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp,
                 tfact, complete, mat, out, *, array_forcing=False, ring_days=1, ring_slots=1,
                 device=0, cells_per_block=0, dfsel=None, coarse=None):
        self._lib = _abi.load()
        self._m: Marshalled = marshal(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                                      Sminp, Smaxp, tfact, complete, mat, out, array_forcing or coarse is not None, device,
                                      0, cells_per_block, dfsel, coarse)
        self._p = C.c_void_p()
        _abi.check(self._lib.mcf_plan_create(C.byref(self._m.inputs), C.byref(self._m.options),
                                             int(ring_days), int(ring_slots), C.byref(self._p)))
        self.rows, self.cols, self.tsteps = self._m.rows, self._m.cols, self._m.tsteps
        self.ndays = self.tsteps // 24
        self.array_forcing = bool(array_forcing) or coarse is not None

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self._lib.mcf_plan_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def valid_cells(self) -> int:
        return int(self._lib.mcf_plan_valid_cells(self._p))

    @property
    def device_bytes(self) -> int:
        return int(self._lib.mcf_plan_bytes(self._p))

    def twi_partial(self):
        s, n = C.c_double(), C.c_int64()
        _abi.check(self._lib.mcf_plan_twi_partial(self._p, C.byref(s), C.byref(n)))
        return s.value, n.value

    def set_twi_mean(self, mean: float):
        _abi.check(self._lib.mcf_plan_set_twi_mean(self._p, float(mean)))

    def upload_forcing_days(self, day0: int, ndays: int, slot: int = 0):
        _abi.check(self._lib.mcf_plan_upload_forcing_days(self._p, C.byref(self._m.inputs), day0, ndays, slot))

    def run_days(self, day0: int, ndays: int, slot: int = 0):
        _abi.check(self._lib.mcf_plan_run_days(self._p, day0, ndays, slot))

    def run_days_at(self, day0: int, ndays: int, slot: int, slot_day0: int):
        """run_days with the days written at day `slot_day0` of the slot (include/mcf.h mcf_plan_run_days_at)."""
        _abi.check(self._lib.mcf_plan_run_days_at(self._p, day0, ndays, slot, slot_day0))

    def run_days_masked(self, day0: int, ndays: int, slot: int, slot_day0: int, skip_tile):
        """run_days_at leaving out the tiles with skip_tile[t] != 0 (include/mcf.h mcf_plan_run_days_masked)."""
        sk = np.ascontiguousarray(skip_tile, dtype=np.uint8)
        _abi.check(self._lib.mcf_plan_run_days_masked(self._p, day0, ndays, slot, slot_day0, sk.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                      int(sk.size)))

    def run_days_cells(self, day0: int, ndays: int, slot: int, slot_day0: int, need_cell_dev: int) -> int:
        """run_days_at for the cells marked in `need_cell_dev` — the ADDRESS of one byte per cell in this device's memory, e.g.
        SnowPlan.free_cells' or a torch uint8 tensor's data_ptr() — gathered into tiles of their own; every other cell's values
        in the slot stay (include/mcf.h mcf_plan_run_days_cells).  -> number of marked cells"""
        n = C.c_int64(0)
        _abi.check(self._lib.mcf_plan_run_days_cells(self._p, day0, ndays, slot, slot_day0, C.c_void_p(int(need_cell_dev)),
                                                     int(self.rows * self.cols), C.byref(n)))
        return int(n.value)

    @property
    def n_tiles(self) -> int:
        lay = self.ring_layout()
        return (lay["cells"] + lay["cells_per_tile"] - 1) // lay["cells_per_tile"]

    def set_mxtc(self, mxtc: float):
        """Replace the series' maximum air temperature (the snow branch solves a subset of the days)."""
        _abi.check(self._lib.mcf_plan_set_mxtc(self._p, float(mxtc)))

    def belowground(self):
        _abi.check(self._lib.mcf_plan_belowground(self._p))

    def sync(self):
        _abi.check(self._lib.mcf_plan_sync(self._p))

    def fetch(self, slot: int, var, step0: int, nsteps: int) -> np.ndarray:
        v = _abi.OUT_NAMES.index(var) if isinstance(var, str) else int(var)
        a = np.empty((self.rows, self.cols, nsteps), dtype=np.float64, order="F")
        _abi.check(self._lib.mcf_plan_fetch(self._p, slot, v, step0, nsteps,
                                            a.ctypes.data_as(_abi.c_double_p)))
        return a

    def fetch_cells(self, slot: int, var, step0: int, nsteps: int, cells) -> np.ndarray:
        """[len(cells), nsteps] values of `var` for the listed cells (0-based column-major indices i + rows*j)."""
        v = _abi.OUT_NAMES.index(var) if isinstance(var, str) else int(var)
        cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int64))
        a = np.empty((cells.size, nsteps), dtype=np.float64, order="F")
        _abi.check(self._lib.mcf_plan_fetch_cells(self._p, slot, v, step0, nsteps,
                                                  cells.ctypes.data_as(C.POINTER(C.c_int64)), cells.size,
                                                  a.ctypes.data_as(_abi.c_double_p)))
        return a

    # writetonc's scale per variable (R/dataprep.R:1158-1167): x100 for temperatures, soil moisture, wind
    NC_SCALE = {"Tz": 100.0, "tleaf": 100.0, "relhum": 1.0, "soilm": 100.0, "windspeed": 100.0, "Rdirdown": 1.0,
                "Rdifdown": 1.0, "Rlwdown": 1.0, "Rswup": 1.0, "Rlwup": 1.0}

    def fetch_packed(self, slot: int, var, step0: int, nsteps: int, scale: float | None = None,
                     timing: bool = False):
        """`writetonc`-packed fetch: int32 [cols, rows, nsteps] (east fastest) = round(value * scale),
        NA -> INT32_MIN (R's NA_integer_).  Returns the array (and the pack kernel's ms with timing)."""
        v = _abi.OUT_NAMES.index(var) if isinstance(var, str) else int(var)
        if scale is None:
            scale = self.NC_SCALE[_abi.OUT_NAMES[v]]
        a = np.empty((self.cols, self.rows, nsteps), dtype=np.int32, order="F")
        ms = C.c_float()
        _abi.check(self._lib.mcf_plan_fetch_packed(self._p, slot, v, step0, nsteps, float(scale),
                                                   a.ctypes.data_as(_abi.c_int32_p),
                                                   C.byref(ms) if timing else None))
        return (a, ms.value) if timing else a

    def ring_layout(self) -> dict:
        """How a ring slot variable is addressed on the device (include/mcf.h mcf_ring_layout)."""
        lay = _abi.RingLayout()
        _abi.check(self._lib.mcf_plan_ring_layout(self._p, C.byref(lay)))
        return {n: int(getattr(lay, n)) for n, _ in lay._fields_}

    def timer_start(self):
        _abi.check(self._lib.mcf_plan_timer_start(self._p))

    def timer_stop(self) -> float:
        ms = C.c_float()
        _abi.check(self._lib.mcf_plan_timer_stop(self._p, C.byref(ms)))
        return ms.value

    def kernel_timing(self, enable: bool = True):
        _abi.check(self._lib.mcf_plan_kernel_timing(self._p, 1 if enable else 0))

    def dispatch_stats(self) -> dict:
        """Which clamp variant the launches ran (include/mcf.h mcf_dispatch_stats): diagnostics, results are identical."""
        st = _abi.DispatchStats()
        _abi.check(self._lib.mcf_plan_dispatch_stats(self._p, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}

    def kernel_stats(self):
        ms, n = C.c_double(), C.c_int64()
        _abi.check(self._lib.mcf_plan_kernel_stats(self._p, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def runbioclim3Cpp(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon, Sminp, Smaxp, tfact, mat, out,
                   wetq, dryq, hotq, colq, air, *, device: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Drop-in for runbioclim3Cpp (src/microclimfCpp.cpp:3620-3658): vegetation arrays [rows, cols, 14], one layer per
    selected day (twelve monthly days, the hottest, the coldest); steps past the 336th stay NA as in the reference."""
    return _bioclim("mcf_runbioclim3", False, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lat, lon,
                    Sminp, Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air, device, True, devices=devices, n_blocks=n_blocks)


def runbioclim4Cpp(obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons, Sminp, Smaxp, tfact, mat, out,
                   wetq, dryq, hotq, colq, air, *, device: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """Drop-in for runbioclim4Cpp (src/microclimfCpp.cpp:3660-3700), array climate."""
    return _bioclim("mcf_runbioclim4", True, obstime, climdata, pointm, vegp, soilc, reqhgt, zref, lats, lons,
                    Sminp, Smaxp, tfact, mat, out, wetq, dryq, hotq, colq, air, device, True, devices=devices, n_blocks=n_blocks)
