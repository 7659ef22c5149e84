"""Host-side mirror of the reference's snow-branch operators.

`gridmodelsnow1/2` and `gridmicrosnow1/2` take the arguments of the R functions of the
same names (R/RcppExports.R:108-114, 124-130; bodies src/microclimfCpp.cpp:4172-4673,
4894-5214): R named lists / data.frames become mappings of numpy arrays, column-major,
and the returned named list becomes a dict.  All arithmetic happens in libmcfhip.so
(mcf_snow.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Sequence

import numpy as np

from . import _abi


class SnowMarshalled:
    def __init__(self):
        self.inputs = _abi.SnowInputs()
        self._keep = []
        self.rows = self.cols = self.tsteps = 0

    def f64(self, a, shape, name):
        arr = np.asarray(a, dtype=np.float64)
        if tuple(arr.shape) != tuple(shape):
            if arr.size != int(np.prod(shape)):
                raise ValueError(f"{name}: expected shape {tuple(shape)}, got {arr.shape}")
            arr = arr.reshape(shape, order="F")
        arr = np.asfortranarray(arr)
        self._keep.append(arr)
        return arr.ctypes.data_as(_abi.c_double_p)

    def i32(self, a, shape, name):
        # Rcpp's as<IntegerVector/IntegerMatrix>() truncates doubles towards zero
        # (NA cells: NA_integer_, as Rcpp's conversion gives)
        v = np.trunc(np.asarray(a, dtype=np.float64))
        arr = np.asfortranarray(np.where(np.isnan(v), float(np.iinfo(np.int32).min), v).astype(np.int32))
        if tuple(arr.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {arr.shape}")
        self._keep.append(arr)
        return arr.ctypes.data_as(_abi.c_int32_p)


def _get(d: Mapping, *names):
    for n in names:
        if n in d:
            return d[n]
    raise KeyError(f"none of {names} present (have {sorted(d)})")


def marshal_snow(obstime: Mapping, climdata: Mapping, vegp: Mapping, other: Mapping, array_forcing: bool,
                 pointm: Mapping | None = None, snowenv: str = "Alpine", micro: bool = False) -> SnowMarshalled:
    """Builds mcf_snow_inputs.  `pointm` (gridmodelsnow) and `micro` (gridmicrosnow: paia, leafd,
    leafden, umu, Smax) select which optional members are filled."""
    m = SnowMarshalled()
    pai = np.asarray(vegp["pai"], dtype=np.float64)
    if pai.ndim != 2:
        raise ValueError("vegp$pai must be a rows x cols matrix")
    R, Cc = pai.shape
    T = len(np.asarray(obstime["year"]))
    m.rows, m.cols, m.tsteps = R, Cc, T
    si = m.inputs
    si.rows, si.cols, si.tsteps = R, Cc, T
    si.array_forcing = 1 if array_forcing else 0
    si.snowenv = _abi.SNOWENV.get(snowenv, 0)     # unknown names fall back to the default (cpp:3743)
    si.obstime.year = m.i32(obstime["year"], (T,), "obstime$year")
    si.obstime.month = m.i32(obstime["month"], (T,), "obstime$month")
    si.obstime.day = m.i32(obstime["day"], (T,), "obstime$day")
    si.obstime.hour = m.f64(obstime["hour"], (T,), "obstime$hour")
    fshape = (R, Cc, T) if array_forcing else (T,)
    for f in _abi.SNOW_CLIM_FIELDS:
        if f == "umu" and not micro:
            setattr(si.clim, f, None)
            continue
        src = _get(climdata, *(("precip", "prec") if f == "precip" else (f,)))   # cpp:5083 reads "prec"
        setattr(si.clim, f, m.f64(src, (T,) if f == "winddir" else fshape, f"climdata${f}"))
    for f in _abi.SNOW_POINTM_FIELDS:
        setattr(si.pointm, f, m.f64(pointm[f], fshape, f"pointm${f}") if pointm is not None else None)
    for f in _abi.SNOW_VEGP_FIELDS:
        if f in ("paia", "leafd", "leafden") and not micro:
            setattr(si.vegp, f, None)
            continue
        setattr(si.vegp, f, m.f64(vegp[f], (R, Cc), f"vegp${f}"))
    o = si.other
    o.slope = m.f64(other["slope"], (R, Cc), "other$slope")
    o.aspect = m.f64(other["aspect"], (R, Cc), "other$aspect")
    o.skyview = m.f64(other["skyview"], (R, Cc), "other$skyview")
    o.wsa = m.f64(other["wsa"], (R, Cc, 8), "other$wsa")
    o.hor = m.f64(other["hor"], (R, Cc, 24), "other$hor")
    o.zref = float(other["zref"])
    if array_forcing:
        o.lats = m.f64(_get(other, "lats", "lat"), (R, Cc), "other$lats")    # cpp:4457 "lats", cpp:5091 "lat"
        o.lons = m.f64(_get(other, "lons", "lon"), (R, Cc), "other$lons")
        o.lat = o.lon = float("nan")
    else:
        o.lat, o.lon = float(other["lat"]), float(other["lon"])
        o.lats = o.lons = None
    if micro:
        o.Smax = m.f64(other["Smax"], (R, Cc), "other$Smax")
        o.isnowdc = o.isnowdg = None
        o.isnowac = o.isnowag = None
    else:
        o.Smax = None
        o.isnowdc = m.f64(other["isnowdc"], (R, Cc), "other$isnowdc")
        o.isnowdg = m.f64(other["isnowdg"], (R, Cc), "other$isnowdg")
        o.isnowac = m.i32(other["isnowac"], (R, Cc), "other$isnowac")
        o.isnowag = m.i32(other["isnowag"], (R, Cc), "other$isnowag")
    return m


def alloc_snowmodel_out(m: SnowMarshalled):
    out = _abi.SnowModelOut()
    arrays = {}
    for f in _abi.SNOWMODEL_OUT3:
        a = np.empty((m.rows, m.cols, m.tsteps), dtype=np.float64, order="F")
        arrays[f] = a
        setattr(out, f, a.ctypes.data_as(_abi.c_double_p))
    for f in _abi.SNOWMODEL_OUT2:
        a = np.empty((m.rows, m.cols), dtype=np.float64, order="F")
        arrays[f] = a
        setattr(out, f, a.ctypes.data_as(_abi.c_double_p))
    return out, arrays


def marshal_snowm(m: SnowMarshalled, snowm: Mapping):
    s = _abi.Snowm()
    shape = (m.rows, m.cols, m.tsteps)
    for f in _abi.SNOWM_FIELDS:
        setattr(s, f, m.f64(snowm[f], shape, f"snowm${f}"))
    return s


def marshal_micro(m: SnowMarshalled, micro: Mapping, out: Sequence):
    """Copies of the requested `micro` fields (the reference updates its argument's storage and
    returns it; here the caller's arrays are left alone and the updated copies are returned)."""
    out = list(out)
    if len(out) != _abi.NOUT:
        raise ValueError("out must have 10 entries")
    sel = (C.c_int32 * _abi.NOUT)(*[1 if v else 0 for v in out])
    outs = _abi.Outputs()
    arrays = {}
    shape = (m.rows, m.cols, m.tsteps)
    for v, name in enumerate(_abi.OUT_NAMES):
        if out[v]:
            a = np.array(np.asarray(micro[name], dtype=np.float64).reshape(shape, order="F"), order="F", copy=True)
            arrays[name] = a
            outs.var[v] = a.ctypes.data_as(_abi.c_double_p)
        else:
            outs.var[v] = None
    return sel, outs, arrays


def _model(fn_name, af, obstime, climdata, pointm, vegp, other, snowenv, device):
    lib = _abi.load()
    m = marshal_snow(obstime, climdata, vegp, other, af, pointm=pointm, snowenv=snowenv)
    out, arrays = alloc_snowmodel_out(m)
    _abi.check(getattr(lib, fn_name)(C.byref(m.inputs), C.byref(out), device))
    return arrays


def gridmodelsnow1(obstime, climdata, pointm, vegp, other, snowenv, *, device: int = 0) -> dict:
    """Snowpack energy and mass balance, data.frame climate: drop-in for the reference's
    gridmodelsnow1 (src/microclimfCpp.cpp:4172-4423).  Returns Tc, Tg, sdepc, sdepg, sden
    [rows, cols, tsteps] and agec, ageg, meltc, meltg [rows, cols]."""
    return _model("mcf_gridmodelsnow1", False, obstime, climdata, pointm, vegp, other, snowenv, device)


def gridmodelsnow2(obstime, climdata, pointm, vegp, other, snowenv, *, device: int = 0) -> dict:
    """Array-climate variant: drop-in for gridmodelsnow2 (src/microclimfCpp.cpp:4426-4673)."""
    return _model("mcf_gridmodelsnow2", True, obstime, climdata, pointm, vegp, other, snowenv, device)


def _micro(fn_name, af, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out, device):
    lib = _abi.load()
    m = marshal_snow(obstime, climdata, vegp, other, af, micro=True)
    sm = marshal_snowm(m, snowm)
    sel, outs, arrays = marshal_micro(m, micro, out)
    _abi.check(getattr(lib, fn_name)(C.byref(m.inputs), C.byref(sm), float(reqhgt), float(mat), C.byref(sel),
                                     C.byref(outs), device))
    return arrays


def gridmicrosnow1(reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out, *, device: int = 0) -> dict:
    """Microclimate of snow-covered cell-steps written over `micro`: drop-in for the reference's
    gridmicrosnow1 (src/microclimfCpp.cpp:4894-5056)."""
    return _micro("mcf_gridmicrosnow1", False, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out,
                  device)


def gridmicrosnow2(reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out, *, device: int = 0) -> dict:
    """Array-climate variant: drop-in for gridmicrosnow2 (src/microclimfCpp.cpp:5059-5214)."""
    return _micro("mcf_gridmicrosnow2", True, reqhgt, obstime, climdata, snowm, micro, vegp, other, mat, out,
                  device)


def snowmodel1_chunks(obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact=0.02, *,
                      chunk_steps: int = 120, device: int = 0, devices=None, n_blocks: int = 0) -> dict:
    """The chunk loop of the reference's `.snowmodel1` (R/internal.R:2553-2617) resident on the
    device: per 5-day chunk terrain refresh from dtm + snow, gridmodelsnow1, `.tpicalc`
    redistribution, hand-over of depths and ages.  Arguments are what the loop works with:
    `pointm` is pointmodelsnow's output, `vegp` the snow-season means of `.sortl`, `other`
    holds lat, lon, zref and the initial isnowdc / isnowdg / isnowac / isnowag.  Returns the
    list `.snowmodel1` returns (R/internal.R:2619) minus `umu`.
    `devices` (a list of HIP ordinals, [] = all visible) / `n_blocks`: the raster in row blocks over several devices from
    this one process (include/mcf.h mcf_snowmodel1_multi); equal up to the summation order of the two raster-wide means."""
    lib = _abi.load()
    R, Cc = np.shape(vegp["pai"])
    oth = dict(other)
    for k, shp in (("slope", (R, Cc)), ("aspect", (R, Cc)), ("skyview", (R, Cc)), ("wsa", (R, Cc, 8)),
                   ("hor", (R, Cc, 24))):
        oth.setdefault(k, np.zeros(shp))          # recomputed on the device every chunk
    m = marshal_snow(obstime, climdata, vegp, oth, False, pointm=pointm, snowenv=snowenv)
    din = _abi.SnowDriverIn()
    din.base = m.inputs
    din.dtm = m.f64(dtm, (R, Cc), "dtm")
    din.res, din.tfact, din.chunk_steps = float(res), float(tfact), int(chunk_steps)
    out = _abi.SnowDriverOut()
    arrays = {}
    for f in _abi.SNOWDRIVER_OUT:
        a = np.empty((R, Cc, m.tsteps), dtype=np.float64, order="F")
        arrays[f] = a
        setattr(out, f, a.ctypes.data_as(_abi.c_double_p))
    if devices is not None or n_blocks:
        mu = _abi.Multi()
        devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
        mu.n_devices, mu.devices, mu.n_blocks = int(devs.size), devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
        _abi.check(lib.mcf_snowmodel1_multi(C.byref(din), C.byref(out), C.byref(mu)))
    else:
        _abi.check(lib.mcf_snowmodel1(C.byref(din), C.byref(out), device))
    return arrays


def _driver_in_array(obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact, af_wind, wsa_s, chunk_steps):
    """mcf_snowdriver_in for array weather at the raster's resolution (include/mcf.h mcf_snowmodel2)"""
    R, Cc = np.shape(vegp["pai"])
    oth = dict(other)
    for k, shp in (("slope", (R, Cc)), ("aspect", (R, Cc)), ("skyview", (R, Cc)), ("wsa", (R, Cc, 8)), ("hor", (R, Cc, 24))):
        oth.setdefault(k, np.zeros(shp))          # recomputed on the device every chunk
    m = marshal_snow(obstime, climdata, vegp, oth, True, pointm=pointm, snowenv=snowenv)
    din = _abi.SnowDriverIn()
    din.base = m.inputs
    din.dtm = m.f64(dtm, (R, Cc), "dtm")
    din.res, din.tfact, din.chunk_steps = float(res), float(tfact), int(chunk_steps)
    din.af_wsa_s = int(wsa_s)
    din.af_wind = m.f64(af_wind, (m.tsteps,), "af_wind")
    return m, din


def snowmodel2_device(obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact=0.02, *, af_wind, wsa_s: int = 0,
                      chunk_steps: int = 120, device: int = 0) -> dict:
    """The chunk loop of `.snowmodel2` (R/internal.R:2950-3008) resident on the device, given what the loop works with: the
    climate and snow point-model arrays ALREADY at the raster's resolution ([rows, cols, T]; `climdata$winddir` [T]),
    `other` = zref, lats, lons and the initial depths / ages, `af_wind` = sqrt(wuv^2 + wvv^2) per step (the coarse wind
    components' spatial means, :2907-2908), `wsa_s` the wind-shelter factor (:2963-2964; 0: from res).  A chunk's slices are
    uploaded as the loop reaches them.  Returns `.snowmodel2`'s list minus umu (include/mcf.h mcf_snowmodel2)."""
    lib = _abi.load()
    m, din = _driver_in_array(obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact, af_wind, wsa_s, chunk_steps)
    R, Cc = m.rows, m.cols
    out = _abi.SnowDriverOut()
    arrays = {}
    for f in _abi.SNOWDRIVER_OUT:
        a = np.empty((R, Cc, m.tsteps), dtype=np.float64, order="F")
        arrays[f] = a
        setattr(out, f, a.ctypes.data_as(_abi.c_double_p))
    _abi.check(lib.mcf_snowmodel2(C.byref(din), C.byref(out), device))
    return arrays


def canintfrac(hgt, pai, uf: float, prec: float, tc: float, Li: float = 0.0) -> np.ndarray:
    """`canintfrac` (src/microclimfCpp.cpp:5417-5450): the canopy's share of a snowfall of `prec` mm per cell (host code)"""
    lib = _abi.load()
    h = np.asfortranarray(np.asarray(hgt, dtype=np.float64))
    p = np.asfortranarray(np.asarray(pai, dtype=np.float64))
    if h.shape != p.shape:
        raise ValueError("hgt and pai differ in shape")
    out = np.empty(h.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_canintfrac(C.c_int64(h.size), h.ctypes.data_as(_abi.c_double_p), p.ctypes.data_as(_abi.c_double_p),
                                  C.c_double(uf), C.c_double(prec), C.c_double(tc), C.c_double(Li),
                                  out.ctypes.data_as(_abi.c_double_p)))
    return out


def meltmu(skyview, stemp, tc) -> np.ndarray:
    """`meltmu` (src/microclimfCpp.cpp:5454-5492): per-cell multiplier of the point model's temperature melt (host code)"""
    lib = _abi.load()
    sv = np.asfortranarray(np.asarray(skyview, dtype=np.float64))
    st = np.ascontiguousarray(stemp, dtype=np.float64)
    ta = np.ascontiguousarray(tc, dtype=np.float64)
    if st.shape != ta.shape or st.ndim != 1:
        raise ValueError("stemp and tc must be vectors of one length")
    out = np.empty(sv.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_meltmu(C.c_int64(sv.size), sv.ctypes.data_as(_abi.c_double_p), C.c_int64(st.size),
                              st.ctypes.data_as(_abi.c_double_p), ta.ctypes.data_as(_abi.c_double_p),
                              out.ctypes.data_as(_abi.c_double_p)))
    return out


def tpicalc(af: int, dtm, tfact: float, *, device: int = 0) -> np.ndarray:
    """`.tpicalc(af, min(dim), dtm, tfact)` (R/internal.R:2483-2496) on the device"""
    lib = _abi.load()
    z = np.asfortranarray(np.asarray(dtm, dtype=np.float64))
    out = np.empty(z.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_tpicalc(C.c_int64(z.shape[0]), C.c_int64(z.shape[1]), z.ctypes.data_as(_abi.c_double_p), C.c_int32(int(af)),
                               C.c_double(tfact), out.ctypes.data_as(_abi.c_double_p), C.c_int32(device)))
    return out


def r_colon(a: int, b: int) -> np.ndarray:
    """R's `a:b` (counts down when a > b) as 0-based indices of 1-based positions"""
    return (np.arange(a, b + 1) if a <= b else np.arange(a, b - 1, -1)) - 1


def snowmodelq1_days(obstime, climdata, pointm, pmod, temp_all, snow_all, subs, vegp, other, snowenv, dtm, res, tfact=0.02, *,
                     device: int = 0) -> dict:
    """The day loop of the reference's fast snow method `.snowmodelq1` (R/internal.R:2690-2776).  `obstime`, `climdata`,
    `pointm`: the selected hours only (whole days); `pmod`: pointmodelsnow's output over the whole series with `temp_all`
    (air temperature) and `snow_all` (precipitation of the hours at or below 2 degC) beside it; `subs`: 1-based positions of
    the selected hours; `vegp`: `.sortl`'s means; `other`: zref, lat, lon, isnowdc, isnowac, isnowag.  Between two selected
    days the pack of every cell moves by the point model's balance, its temperature melt scaled by `meltmu`; each selected
    day runs gridmodelsnow1 on the device over terrain of the bare dtm and spreads the ground-snow change by `.tpicalc`.
    Reference behaviours kept: the ground depths stacked on the dtm for the position index are still zero when they are
    read (:2750), snow ages are not handed on, and a first selected day that is the first day of the series fails (`sbtn`
    undefined there)."""
    from . import terrain as T
    z = np.asarray(dtm, dtype=np.float64)
    subs = np.asarray(subs, dtype=np.int64)
    n = subs.size
    if n % 24 or n == 0:
        raise ValueError("the fast snow method works on whole selected days")
    if subs[0] - 1 <= 1:
        raise ValueError("the fast snow method cannot start on the first day of the series (the reference fails there: "
                         "`sbtn` not found)")
    R, Cc = z.shape
    zref = float(other["zref"])
    oth = dict(other)
    oth.update(T.snow_terrain(z, res, zref, device=device))
    temp_s = np.asarray(climdata["temp"], dtype=np.float64)
    wind_s = np.asarray(climdata["windspeed"], dtype=np.float64)
    snow_all = np.asarray(snow_all, dtype=np.float64)
    pos = snow_all[snow_all > 0]
    msnow = float(pos.mean()) if pos.size else float("nan")
    intfrac = canintfrac(vegp["hgt"], vegp["pai"], 2.0, msnow, float(temp_s.mean()), 0.0)
    isnowdc = np.array(oth["isnowdc"], dtype=np.float64)
    isnowdg = (1 - intfrac) * isnowdc
    out = {k: np.full((R, Cc, n), np.nan, order="F") for k in ("Tc", "Tg", "sdepc", "snowden")}
    out["sdepg"] = np.zeros((R, Cc, n), order="F")
    pai = np.asarray(vegp["pai"], dtype=np.float64)
    ped = 0
    with np.errstate(invalid="ignore"):
        for day in range(n // 24):
            sl = slice(day * 24, day * 24 + 24)
            first = int(subs[day * 24])
            if first - 1 > 1:
                sbtn = r_colon(ped + 1, first - 1)
                mu = meltmu(oth["skyview"], pmod["sstemp"][sbtn], np.asarray(temp_all)[sbtn])
                melt = pmod["sublmelt"][sbtn].sum() + pmod["rainmelt"][sbtn].sum() + mu * pmod["tempmelt"][sbtn].sum()
                fall = (snow_all[sbtn] / 1000).sum()
                balancec, balanceg = fall - melt, (1 - intfrac) * fall - np.exp(-pai) * melt
            else:
                balancec = balanceg = 0.0                          # sbtn of the day before stays in force
            isnowdc = isnowdc + balancec * (1000 / pmod["sdenc"][sbtn].mean())
            isnowdg = isnowdg + balanceg * (1000 / pmod["sdeng"][sbtn].mean())
            isnowdc[isnowdc < 0] = 0
            isnowdg[isnowdg < 0] = 0
            oth["isnowdc"], oth["isnowdg"] = isnowdc, isnowdg
            smod = gridmodelsnow1({k: np.asarray(v)[sl] for k, v in obstime.items()},
                                  {k: np.asarray(v)[sl] for k, v in climdata.items()},
                                  {k: np.asarray(v)[sl] for k, v in pointm.items()}, vegp, oth, snowenv, device=device)
            dsnow = smod["sdepc"] - isnowdc[:, :, None]
            dsnowg = smod["sdepg"] - isnowdg[:, :, None]
            af = int(np.round(10 * wind_s[sl].mean() ** 0.5 / res))    # half to even, like R's round
            tpi = tpicalc(af, z, tfact, device=device)
            dsnowg2 = dsnowg * tpi[:, :, None]
            sdc = (dsnow - dsnowg) + dsnowg2 + isnowdc[:, :, None]
            sdg = dsnowg2 + isnowdg[:, :, None]
            sdc[sdc < 0] = 0
            sdg[sdg < 0] = 0
            out["Tc"][:, :, sl], out["Tg"][:, :, sl], out["snowden"][:, :, sl] = smod["Tc"], smod["Tg"], smod["sden"]
            out["sdepc"][:, :, sl], out["sdepg"][:, :, sl] = sdc, sdg
            ped = int(subs[day * 24 + 23])
            isnowdc, isnowdg = sdc[:, :, 23].copy(), sdg[:, :, 23].copy()
        swe = out["sdepc"] * out["snowden"]
    return {"Tc": out["Tc"], "Tg": out["Tg"], "groundsnowdepth": out["sdepg"], "totalSWE": swe, "snowden": out["snowden"]}


def _fine_snow_inputs(clim_c, pointm_c, sl, z, zc, rowpos, colpos, altcorrect, wu_c, wv_c, winddir):
    """Steps `sl` of the coarse climate and snow point-model arrays on the fine raster, as `.snowmodel2` / `.snowmodelq2`
    prepare them (R/internal.R:2862-2925, 3108-3170): `.cca` = bilinear and masked by the dtm, pressure and wind
    components unmasked, altitude correction 0 / 1 / 2, relative humidity capped at 100"""
    from .rformulas import lapserate_R, satvap_R, upsample_coarse
    hole = np.isnan(z)
    up = lambda a: upsample_coarse(np.asarray(a)[:, :, sl], rowpos, colpos)            # noqa: E731
    cca = lambda a: np.where(hole[:, :, None], np.nan, up(a))                          # noqa: E731
    temp, relhum = cca(clim_c["temp"]), cca(clim_c["relhum"])
    if altcorrect == 0:
        pres = up(clim_c["pres"])
    else:
        ea = satvap_R(temp) * relhum / 100
        pres = up(np.asarray(clim_c["pres"]) / (((293 - 0.0065 * zc) / 293) ** 5.26)[:, :, None]) * (((293 - 0.0065 * z) / 293) ** 5.26)[:, :, None]
        elevd = (upsample_coarse(zc, rowpos, colpos) - z)[:, :, None]
        lr = 5 / 1000 if altcorrect == 1 else lapserate_R(temp, ea, pres)
        temp = lr * elevd + temp
        relhum = (ea / satvap_R(temp)) * 100
    relhum = np.where(relhum > 100, 100.0, relhum)
    clim = {"temp": temp, "relhum": relhum, "pres": pres, "swdown": cca(clim_c["swdown"]), "difrad": cca(clim_c["difrad"]),
            "lwdown": cca(clim_c["lwdown"]), "precip": cca(clim_c["precip"]),
            "windspeed": np.sqrt(up(wu_c) ** 2 + up(wv_c) ** 2), "winddir": winddir[sl]}
    return clim, {k: cca(pointm_c[k]) for k in ("Gp", "Tc", "RswabsG", "RlwabsG", "umu", "tr")}


def meltmu2(mu, stemp, tc) -> np.ndarray:
    """`meltmu2` (src/microclimfCpp.cpp:5495-5527): as `meltmu` with the snow surface and air temperatures given per
    cell [rows, cols, n]; 0.5 where the surface never thaws (host code)"""
    lib = _abi.load()
    m = np.asfortranarray(np.asarray(mu, dtype=np.float64))
    st = np.asfortranarray(np.asarray(stemp, dtype=np.float64))
    ta = np.asfortranarray(np.asarray(tc, dtype=np.float64))
    if st.shape != ta.shape or st.shape[:2] != m.shape:
        raise ValueError("stemp and tc must be [rows, cols, n] over mu's raster")
    out = np.empty(m.shape, dtype=np.float64, order="F")
    _abi.check(lib.mcf_meltmu2(C.c_int64(m.size), C.c_int64(st.shape[2]), m.ctypes.data_as(_abi.c_double_p),
                               st.ctypes.data_as(_abi.c_double_p), ta.ctypes.data_as(_abi.c_double_p),
                               out.ctypes.data_as(_abi.c_double_p)))
    return out


def snowmodelq2_days(obstime, clim_c, pointm_c, pm2_c, subs, vegp, other, snowenv, dtm, dtmc, res, tfact=0.02, *, rowpos, colpos,
                     altcorrect: int = 0, device: int = 0) -> dict:
    """The second half of the reference's fast array-weather snow method `.snowmodelq2` (R/internal.R:3108-3283).
    `obstime`, `clim_c`, `pointm_c`: the selected hours (coarse arrays as for `snowmodel2_chunks`); `pm2_c`: coarse arrays
    over the WHOLE series — sublmelt, tempmelt, rainmelt, snow, sstemp, tc, sdenc, sdeng; `subs`: 1-based positions of the
    selected hours.  Between two selected days each cell's pack moves by the resampled point-model balance (`meltmu2`
    scaling the temperature melt by sky view), each selected day runs gridmodelsnow2 on terrain of the bare dtm and
    spreads the ground-snow change by `.tpicalc`.  Unlike `.snowmodelq1` a first selected day that is the first day of
    the series is accepted (no adjustment then)."""
    from . import terrain as T
    from .rformulas import upsample_coarse
    z = np.asarray(dtm, dtype=np.float64)
    hole = np.isnan(z)
    R, Cc = z.shape
    subs = np.asarray(subs, dtype=np.int64)
    n = subs.size
    if n % 24 or n == 0:
        raise ValueError("the fast snow method works on whole selected days")
    zc = np.nan_to_num(np.asarray(dtmc, dtype=np.float64), nan=0.0)
    wd = np.asarray(clim_c["winddir"], dtype=np.float64) * np.pi / 180
    wu_c = np.asarray(clim_c["windspeed"], dtype=np.float64) * np.cos(wd)
    wv_c = np.asarray(clim_c["windspeed"], dtype=np.float64) * np.sin(wd)
    wuv, wvv = np.nanmean(wu_c, axis=(0, 1)), np.nanmean(wv_c, axis=(0, 1))
    winddir = (np.arctan2(wvv, wuv) * 180 / np.pi) % 360
    oth = dict(other)
    oth.update(T.snow_terrain(z, res, float(other["zref"]), device=device))
    vg = dict(vegp)
    vg["leaft"] = np.where(np.isnan(vg["leaft"]), 0.01, vg["leaft"])
    pai = np.asarray(vg["pai"], dtype=np.float64)
    up = lambda a: upsample_coarse(a, rowpos, colpos)                                      # noqa: E731
    cca = lambda a: np.where(hole[:, :, None], np.nan, up(a))                              # noqa: E731
    snow_c = np.asarray(pm2_c["snow"], dtype=np.float64)
    pos = snow_c[snow_c > 0]
    msnow = float(pos.mean()) if pos.size else float("nan")
    mtemp = float(np.nanmean(cca(pm2_c["tc"])))
    intfrac = canintfrac(vg["hgt"], vg["pai"], 2.0, msnow, mtemp, 0.0)
    isnowdc = np.array(oth["isnowdc"], dtype=np.float64)
    isnowdg = (1 - intfrac) * isnowdc
    names = ("Tc", "Tg", "sdepc", "snowden", "umu")
    out = {k: np.full((R, Cc, n), np.nan, order="F") for k in names}
    out["sdepg"] = np.zeros((R, Cc, n), order="F")
    ped = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        for day in range(n // 24):
            sl = slice(day * 24, day * 24 + 24)
            first = int(subs[day * 24])
            if first - 1 > 1:
                sbtn = r_colon(ped + 1, first - 1)
                tot = lambda k: up(np.asarray(pm2_c[k])[:, :, sbtn].sum(axis=2))          # noqa: E731  `.resamplemelt`
                mu = meltmu2(oth["skyview"], cca(np.asarray(pm2_c["sstemp"])[:, :, sbtn]), cca(np.asarray(pm2_c["tc"])[:, :, sbtn]))
                melt = tot("sublmelt") + tot("rainmelt") + mu * tot("tempmelt")
                fall = tot("snow") / 1000
                isnowdc = isnowdc + (fall - melt) * (1000 / (tot("sdenc") / sbtn.size))
                isnowdg = isnowdg + ((1 - intfrac) * fall - np.exp(-pai) * melt) * (1000 / (tot("sdeng") / sbtn.size))
            isnowdc[isnowdc < 0] = 0
            isnowdg[isnowdg < 0] = 0
            oth["isnowdc"], oth["isnowdg"] = isnowdc, isnowdg
            clim, pointm = _fine_snow_inputs(clim_c, pointm_c, sl, z, zc, rowpos, colpos, altcorrect, wu_c, wv_c, winddir)
            smod = gridmodelsnow2({k: np.asarray(v)[sl] for k, v in obstime.items()}, clim, pointm, vg, oth, snowenv, device=device)
            dsnow = smod["sdepc"] - isnowdc[:, :, None]
            dsnowg = smod["sdepg"] - isnowdg[:, :, None]
            af = int(np.round(10 * np.mean(np.sqrt(wuv[sl] ** 2 + wvv[sl] ** 2)) ** 0.5 / res))
            tpi = tpicalc(af, z, tfact, device=device)
            dsnowg2 = dsnowg * tpi[:, :, None]
            sdc = (dsnow - dsnowg) + dsnowg2 + isnowdc[:, :, None]
            sdg = dsnowg2 + isnowdg[:, :, None]
            sdc[sdc < 0] = 0
            sdg[sdg < 0] = 0
            out["Tc"][:, :, sl], out["Tg"][:, :, sl], out["snowden"][:, :, sl] = smod["Tc"], smod["Tg"], smod["sden"]
            out["sdepc"][:, :, sl], out["sdepg"][:, :, sl], out["umu"][:, :, sl] = sdc, sdg, pointm["umu"]
            ped = int(subs[day * 24 + 23])
            isnowdc, isnowdg = sdc[:, :, 23].copy(), sdg[:, :, 23].copy()
        res_ = {"Tc": out["Tc"], "Tg": out["Tg"], "groundsnowdepth": out["sdepg"], "totalSWE": out["sdepc"] * out["snowden"],
                "snowden": out["snowden"], "umu": out["umu"]}
    for v in res_.values():                                             # `.cleansmod`
        v[hole] = np.nan
    return res_


def snowmodel2_chunks(obstime, clim_c, pointm_c, vegp, other, snowenv, dtm, dtmc, res, tfact=0.02, *, rowpos, colpos,
                      altcorrect: int = 0, agg: int = 10, chunk_steps: int = 120, device: int = 0) -> dict:
    """The second half of the reference's `.snowmodel2` (R/internal.R:2862-3013): coarse climate and snow point-model
    arrays [crows, ccols, T] brought to the fine raster chunk by chunk (bilinear, `.cca`; altitude correction 0 / 1 / 2 of
    :2871-2890; wind from resampled components), then per 5-day chunk the terrain of dtm + ground snow, gridmodelsnow2
    and `.tpicalc` on the device, the redistribution and the hand-over of depths and ages.  `clim_c`: temp, relhum, pres,
    swdown, difrad, lwdown, precip, windspeed [crows, ccols, T] and winddir [T] (degrees, the same in every cell as `.todf`
    makes it); `pointm_c`: Gp, Tc, RswabsG, RlwabsG, umu, tr; `vegp`: `.sortl`'s means; `other`: zref, lats, lons, isnowdc,
    isnowdg, isnowac, isnowag.  Reference behaviours kept: `other$isnowdg` is never updated, the aggregation factor of the
    position index is at least 2, `1:n5days` truncates."""
    from . import terrain as T
    from .rformulas import upsample_coarse
    z = np.asarray(dtm, dtype=np.float64)
    R, Cc = z.shape
    h = len(np.asarray(obstime["year"]))
    nch = h // chunk_steps
    if nch < 1:
        raise ValueError("the array snow model needs at least one whole 5-day chunk")
    hole = np.isnan(z)
    zc = np.nan_to_num(np.asarray(dtmc, dtype=np.float64), nan=0.0)
    wd = np.asarray(clim_c["winddir"], dtype=np.float64) * np.pi / 180
    wu_c = np.asarray(clim_c["windspeed"], dtype=np.float64) * np.cos(wd)
    wv_c = np.asarray(clim_c["windspeed"], dtype=np.float64) * np.sin(wd)
    wuv, wvv = np.nanmean(wu_c, axis=(0, 1)), np.nanmean(wv_c, axis=(0, 1))
    winddir = (np.arctan2(wvv, wuv) * 180 / np.pi) % 360
    names = ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden", "umu")
    out = {k: np.full((R, Cc, h), np.nan, order="F") for k in names}
    oth = dict(other)
    vg = dict(vegp)
    vg["leaft"] = np.where(np.isnan(vg["leaft"]), 0.001, vg["leaft"])
    isnowdg = np.asarray(other["isnowdg"], dtype=np.float64)
    dtms = z + isnowdg
    with np.errstate(invalid="ignore"):
        for ch in range(nch):
            sl = slice(ch * chunk_steps, min((ch + 1) * chunk_steps, h))
            oth.update(T.snow_terrain(dtms, res, float(other["zref"]), agg=agg, mask=z, device=device))
            clim, pointm = _fine_snow_inputs(clim_c, pointm_c, sl, z, zc, rowpos, colpos, altcorrect, wu_c, wv_c, winddir)
            smod = gridmodelsnow2({k: np.asarray(v)[sl] for k, v in obstime.items()}, clim, pointm, vg, oth, snowenv, device=device)
            af = max(int(np.round(10 * np.mean(np.sqrt(wuv[sl] ** 2 + wvv[sl] ** 2)) ** 0.5 / res)), 2)
            tpi = tpicalc(af, dtms, tfact, device=device)[:, :, None]
            asd = isnowdg[:, :, None]
            dsnow = smod["sdepg"] - asd
            dsnow2 = np.where(dsnow < 0, dsnow, dsnow * tpi)
            asc = np.asarray(oth["isnowdc"], dtype=np.float64)[:, :, None]
            tot = asc + (smod["sdepc"] - asc - dsnow) + dsnow2
            out["Tc"][:, :, sl], out["Tg"][:, :, sl], out["snowden"][:, :, sl] = smod["Tc"], smod["Tg"], smod["sden"]
            out["totalSWE"][:, :, sl] = tot * smod["sden"]
            out["groundsnowdepth"][:, :, sl] = asd + dsnow2
            out["umu"][:, :, sl] = pointm["umu"]                       # `umu = pointm$umu`
            oth["isnowdc"] = tot[:, :, -1]
            oth["isnowac"], oth["isnowag"] = smod["agec"], smod["ageg"]
            dtms = z + out["groundsnowdepth"][:, :, sl.stop - 1]
    tail = slice(nch * chunk_steps, h)                                  # steps past the last whole chunk: only umu is filled
    if tail.start < h:
        out["umu"][:, :, tail] = upsample_coarse(np.asarray(pointm_c["umu"])[:, :, tail], rowpos, colpos)
    for k in names:                                                     # `.cleansmod`
        out[k][hole] = np.nan
    return out


APPLY_FUNS = {"mean": 0, "sum": 1, "max": 2, "min": 3}


def applycpp3(a, fun_name: str, *, device: int = 0, with_count: bool = False):
    """Reduction over space per time step, NA skipped: drop-in for the reference's applycpp3
    (src/microclimfCpp.cpp:5553-5588).  `with_count=True` also returns the non-NA cell counts
    (what a row-block rank contributes to a raster-wide mean, see distributed.allreduce_apply3)."""
    if fun_name not in APPLY_FUNS:
        raise ValueError("Unknown function name")           # the reference's stop() message
    lib = _abi.load()
    arr = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if arr.ndim != 3:
        raise ValueError("applycpp3 needs a [rows, cols, tsteps] array")
    R, Cc, T = arr.shape
    res = np.empty(T)
    cnt = np.empty(T) if with_count else None
    _abi.check(lib.mcf_applycpp3(arr.ctypes.data_as(_abi.c_double_p), R, Cc, T, APPLY_FUNS[fun_name],
                                 res.ctypes.data_as(_abi.c_double_p),
                                 cnt.ctypes.data_as(_abi.c_double_p) if with_count else None, device))
    return (res, cnt) if with_count else res


def snowdaysfun(maxsnowdepth, minsnowdepth) -> dict:
    """Host-side mirror of the reference's snowdaysfun (src/microclimfCpp.cpp:5531-5550): a day is a
    snow day if any hour has snow somewhere (max > 0) and a no-snow day if any hour has a snow-free
    cell (min == 0); a day can be both."""
    mx = np.asarray(maxsnowdepth, dtype=np.float64)
    mn = np.asarray(minsnowdepth, dtype=np.float64)
    days = mx.size // 24
    with np.errstate(invalid="ignore"):
        snow = (mx[:days * 24].reshape(days, 24) > 0.0).any(axis=1)
        nosnow = (mn[:days * 24].reshape(days, 24) == 0.0).any(axis=1)
    return {"snowdays": snow.astype(np.int32), "nosnowdays": nosnow.astype(np.int32)}


def merge_snow_outputs(moutn: Mapping, mouts: Mapping, snowdays, nosnowdays, rows: int, cols: int) -> dict:
    """Step (5) of `.runmicrosnow1/2` (R/internal.R:3633-3656): days with snow anywhere take the snow
    microclimate (`mouts`, which already carries the no-snow solver's values on its snow-free cell-steps),
    the other days the no-snow solver's output (`moutn`).  `snowdays` / `nosnowdays` are 1-based day
    numbers as in R; `moutn` covers `nosnowdays` in order, `mouts` covers `snowdays` in order."""
    snowdays = np.asarray(snowdays, dtype=np.int64)
    nosnowdays = np.asarray(nosnowdays, dtype=np.int64)
    if nosnowdays.size == 0:
        return dict(mouts)
    if snowdays.size == 0:
        return dict(moutn)
    tdays = np.unique(np.concatenate([snowdays, nosnowdays]))
    nosnow = np.setdiff1d(tdays, snowdays)
    s1 = np.repeat(np.isin(nosnowdays, nosnow), 24)
    hours = np.arange(24)
    nosnowh = (np.repeat((nosnow - 1) * 24, 24) + np.tile(hours, nosnow.size)).astype(np.int64)
    snowh = (np.repeat((snowdays - 1) * 24, 24) + np.tile(hours, snowdays.size)).astype(np.int64)
    n = nosnowh.size + snowh.size
    out = {}
    for k, v in moutn.items():
        a = np.full((rows, cols, n), np.nan, order="F")
        a[:, :, nosnowh] = np.asarray(v)[:, :, s1]
        a[:, :, snowh] = np.asarray(mouts[k])
        out[k] = a
    return out


class SnowPlan:
    """One rank's row block of `.snowmodel1`'s chunk loop, resident on the device (include/mcf.h
    mcf_snowplan_*).  Arguments as snowmodel1_chunks, for the block's own rows; `row0` / `rows_total`
    place the block in the raster."""

    def __init__(self, obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact=0.02, *, chunk_steps=120,
                 row0=0, rows_total=0, device=0, keep_results=True):
        self._lib = _abi.load()
        R, Cc = np.shape(vegp["pai"])
        oth = dict(other)
        for k, shp in (("slope", (R, Cc)), ("aspect", (R, Cc)), ("skyview", (R, Cc)), ("wsa", (R, Cc, 8)),
                       ("hor", (R, Cc, 24))):
            oth.setdefault(k, np.zeros(shp))
        self._m = marshal_snow(obstime, climdata, vegp, oth, False, pointm=pointm, snowenv=snowenv)
        din = _abi.SnowDriverIn()
        din.base = self._m.inputs
        din.dtm = self._m.f64(dtm, (R, Cc), "dtm")
        din.res, din.tfact, din.chunk_steps = float(res), float(tfact), int(chunk_steps)
        self._p = C.c_void_p()
        _abi.check(self._lib.mcf_snowplan_create(C.byref(din), int(row0), int(rows_total), device, C.byref(self._p)))
        self.rows, self.cols, self.tsteps = R, Cc, self._m.tsteps
        self.device = int(device)
        self.chunks = int(self._lib.mcf_snowplan_chunks(self._p))
        self._chunk_steps = int(chunk_steps) if chunk_steps else 120
        self._out = _abi.SnowDriverOut()
        self.result = {}
        for f in _abi.SNOWDRIVER_OUT:          # keep_results=False: the series stay on the device (no [rows, cols, tsteps] host arrays)
            if keep_results:
                a = np.empty((R, Cc, self.tsteps), dtype=np.float64, order="F")
                self.result[f] = a
                setattr(self._out, f, a.ctypes.data_as(_abi.c_double_p))
            else:
                setattr(self._out, f, None)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self._lib.mcf_snowplan_destroy(self._p)
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

    def surface(self, out=None) -> np.ndarray:
        """dtm + ground snow depth of the own rows; `out`: a column-major [rows, cols] array to fill (reused across chunks)"""
        a = np.empty((self.rows, self.cols), dtype=np.float64, order="F") if out is None else out
        if a.shape != (self.rows, self.cols) or not a.flags.f_contiguous:
            raise ValueError("out must be a column-major [rows, cols] array")
        _abi.check(self._lib.mcf_snowplan_surface(self._p, a.ctypes.data_as(_abi.c_double_p)))
        return a

    def handover(self) -> np.ndarray:
        """The pack depth handed to the next chunk (`other$isnowdc`), own rows."""
        a = np.empty((self.rows, self.cols), dtype=np.float64, order="F")
        _abi.check(self._lib.mcf_snowplan_handover(self._p, a.ctypes.data_as(_abi.c_double_p)))
        return a

    def apply3(self, chunk: int, fun_name: str):
        """applycpp3 of the chunk's totalSWE on the device: (result, non-NA counts), each [steps of the chunk]"""
        fun = {"mean": 0, "sum": 1, "max": 2, "min": 3}[fun_name]
        n = min(self._chunk_steps, self.tsteps - chunk * self._chunk_steps)
        r, c = np.empty(n), np.empty(n)
        _abi.check(self._lib.mcf_snowplan_apply3(self._p, int(chunk), fun, r.ctypes.data_as(_abi.c_double_p),
                                                 c.ctypes.data_as(_abi.c_double_p)))
        return r, c

    def surface_partial(self):
        s, n = C.c_double(), C.c_double()
        _abi.check(self._lib.mcf_snowplan_surface_partial(self._p, C.byref(s), C.byref(n)))
        return s.value, n.value

    def prepare_chunk(self, chunk: int, ext=None, halo_north: int = 0, halo_south: int = 0, surface_mean: float = 0.0):
        s, n = C.c_double(), C.c_double()
        p = None
        if ext is not None:
            e = np.asfortranarray(np.asarray(ext, dtype=np.float64))
            if e.shape != (halo_north + self.rows + halo_south, self.cols):
                raise ValueError("ext must be [halo_north + rows + halo_south, cols]")
            p = e.ctypes.data_as(_abi.c_double_p)
        _abi.check(self._lib.mcf_snowplan_prepare_chunk(self._p, int(chunk), p, int(halo_north), int(halo_south),
                                                        float(surface_mean), C.byref(s), C.byref(n)))
        return s.value, n.value

    def pack_halo(self, north=None, south=None):
        """The own block's first / last surface rows into DEVICE tensors `north` [cols, h] / `south` (torch, float64, contiguous:
        the memory is a column-major [h, cols] piece) — what the neighbouring ranks receive as their halos.  No host staging."""
        hn, pn = _dev_piece(north, self.cols)
        hs, ps = _dev_piece(south, self.cols)
        _abi.check(self._lib.mcf_snowplan_pack_halo(self._p, hn, pn, hs, ps))

    def prepare_chunk_dev(self, chunk: int, north=None, south=None, surface_mean: float = 0.0):
        """prepare_chunk with the halo pieces as device tensors ([cols, h], as pack_halo writes them)."""
        hn, pn = _dev_piece(north, self.cols)
        hs, ps = _dev_piece(south, self.cols)
        s, n = C.c_double(), C.c_double()
        _abi.check(self._lib.mcf_snowplan_prepare_chunk_dev(self._p, int(chunk), pn, hn, ps, hs, float(surface_mean), C.byref(s),
                                                            C.byref(n)))
        return s.value, n.value

    def run_chunk(self, chunk: int, tpic_mean: float):
        _abi.check(self._lib.mcf_snowplan_run_chunk(self._p, int(chunk), float(tpic_mean), C.byref(self._out)))

    # ---- the snow-day microclimate inside the chunk loop (include/mcf.h: two passes over the year) ------------------
    def reset(self):
        """Back to the series' start (hand-over depths, ages, snow surface): the second pass."""
        _abi.check(self._lib.mcf_snowplan_reset(self._p))

    FETCH = {"isnowdc": 0, "isnowac": 1, "isnowag": 2, "slope": 3, "aspect": 4, "skyview": 5, "wsa": 6, "hor": 7,
             "Tc": 8, "Tg": 9, "groundsnowdepth": 10, "snowden": 11, "totalSWE": 12}

    def fetch_cells(self, what: str, cells) -> np.ndarray:
        """[len(cells), depth] values of one of the plan's device arrays (FETCH) for a sample of cells (0-based column-major
        indices within the block): the hand-over state, the chunk's terrain, the series of the chunk run last."""
        c = np.ascontiguousarray(cells, dtype=np.int64)
        depth = {"wsa": 8, "hor": 24}.get(what, self._chunk_steps if self.FETCH[what] >= 8 else 1)
        out = np.empty((c.size, depth), dtype=np.float64, order="F")
        d = C.c_int32()
        _abi.check(self._lib.mcf_snowplan_fetch_cells(self._p, self.FETCH[what], c.ctypes.data_as(C.POINTER(C.c_int64)),
                                                      int(c.size), out.ctypes.data_as(_abi.c_double_p), C.byref(d)))
        assert d.value == depth
        return out

    def keep_chunk(self, chunk: int, reserve_bytes: int = 32 << 30) -> bool:
        """After run_chunk (and apply3 / meand_accumulate): the chunk's series stay on the device for the second pass if
        `reserve_bytes` of device memory remain free; True if kept (then pass 2 only calls microsnow for it)."""
        k = C.c_int32()
        _abi.check(self._lib.mcf_snowplan_keep_chunk(self._p, int(chunk), int(reserve_bytes), C.byref(k)))
        return bool(k.value)

    def can_keep(self, reserve_bytes: int = 32 << 30) -> bool:
        """Would keep_chunk keep a chunk now? (include/mcf.h mcf_snowplan_can_keep)"""
        k = C.c_int32()
        _abi.check(self._lib.mcf_snowplan_can_keep(self._p, int(reserve_bytes), C.byref(k)))
        return bool(k.value)

    SERIES_ALL, SERIES_PASS1 = 31, 4 | 16       # all five; totalSWE (day classes) + density (mean damping depth)

    def set_series(self, mask: int):
        """Which of the five device series run_chunk writes from now on (include/mcf.h mcf_snowplan_set_series)."""
        _abi.check(self._lib.mcf_snowplan_set_series(self._p, int(mask)))

    def release_kept(self):
        _abi.check(self._lib.mcf_snowplan_release_kept(self._p))

    def checkpoint(self, chunk: int):
        """Keeps the state `chunk` starts from on the device (call before its prepare_chunk, in the first pass)."""
        _abi.check(self._lib.mcf_snowplan_checkpoint(self._p, int(chunk)))

    def restore(self, chunk: int):
        """Puts the checkpointed start state of `chunk` back: the second pass re-runs only the chunks with a snow day."""
        _abi.check(self._lib.mcf_snowplan_restore(self._p, int(chunk)))

    def meand_accumulate(self, chunk: int, snowday):
        sd = np.ascontiguousarray(snowday, dtype=np.int32)
        _abi.check(self._lib.mcf_snowplan_meand_accumulate(self._p, int(chunk), sd.ctypes.data_as(_abi.c_int32_p)))

    def micro_setup(self, reqhgt, obstime, climdata, vegp, other, mat, out, sub_of_day, reuse_static: bool = False):
        """gridmicrosnow1's inputs for the snow-day SUBSET series (what `.prepsnowinputs1` hands it: subset weather incl.
        umu, `.sortl2` vegetation, bare-ground terrain + Smax) and, per day of the whole series, its day in the subset
        (-1: not a snow day).  `reuse_static`: the matrices (vegp, other) of the previous set-up stay on the device — only the
        series and the day map are new."""
        if reuse_static and getattr(self, "_micro_static", None) is not None:
            m = self._micro_static                     # the marshalled matrices (kept alive; not uploaded again)
            T = len(np.asarray(obstime["year"]))
            si = m.inputs
            si.tsteps = m.tsteps = T
            si.obstime.year = m.i32(obstime["year"], (T,), "obstime$year")
            si.obstime.month = m.i32(obstime["month"], (T,), "obstime$month")
            si.obstime.day = m.i32(obstime["day"], (T,), "obstime$day")
            si.obstime.hour = m.f64(obstime["hour"], (T,), "obstime$hour")
            for f in _abi.SNOW_CLIM_FIELDS:
                src = _get(climdata, *(("precip", "prec") if f == "precip" else (f,)))
                setattr(si.clim, f, m.f64(src, (T,), f"climdata${f}"))
        else:
            m = marshal_snow(obstime, climdata, vegp, other, False, micro=True)
            self._micro_static = m
        sod = np.ascontiguousarray(sub_of_day, dtype=np.int32)
        sel = (C.c_int32 * _abi.NOUT)(*[1 if v else 0 for v in out])
        _abi.check(self._lib.mcf_snowplan_micro_setup(self._p, C.byref(m.inputs), sod.ctypes.data_as(_abi.c_int32_p),
                                                      int(sod.size), float(reqhgt), float(mat), C.byref(sel), 1 if reuse_static else 0))

    def covered_tiles(self, plan, chunk: int, day: int, ndays: int):
        """uint8 [tiles of `plan`]: 1 where every cell of the tile lies under snow at every step of the chunk's days
        [day, day + ndays) — mcf_snowplan_microsnow overwrites all its values, the solver may leave the tile out
        (include/mcf.h mcf_snowplan_covered_tiles); and the number of such tiles"""
        sk = np.zeros(plan.n_tiles, np.uint8)
        n = C.c_int64(0)
        _abi.check(self._lib.mcf_snowplan_covered_tiles(self._p, plan._p, int(chunk), int(day), int(ndays),
                                                        sk.ctypes.data_as(C.POINTER(C.c_uint8)), int(sk.size), C.byref(n)))
        return sk, int(n.value)

    def free_cells(self, plan, chunk: int, day: int, ndays: int):
        """-> (device address of one byte per cell, number of ones): 1 where the cell is NOT under snow at every step of the
        chunk's days [day, day + ndays) (or has no vegetation height) — the cells whose solver values survive the merge, for
        Plan.run_days_cells (include/mcf.h mcf_snowplan_free_cells)"""
        ptr, n = C.c_void_p(0), C.c_int64(0)
        _abi.check(self._lib.mcf_snowplan_free_cells(self._p, plan._p, int(chunk), int(day), int(ndays), C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def microsnow(self, plan, chunk: int, slot: int, nosnowday):
        """gridmicrosnow1 on the chunk's snow days, written over the solver's outputs in ring slot `slot` of `plan`."""
        nd = np.ascontiguousarray(nosnowday, dtype=np.int32)
        _abi.check(self._lib.mcf_snowplan_microsnow(self._p, plan._p, int(chunk), int(slot), nd.ctypes.data_as(_abi.c_int32_p)))


def _dev_piece(t, cols):
    """(rows, device pointer) of a halo piece held as a torch tensor [cols, h] on the GPU"""
    if t is None or t.numel() == 0:
        return 0, None
    if not t.is_cuda or str(t.dtype) != "torch.float64" or not t.is_contiguous() or t.dim() != 2 or t.shape[0] != cols:
        raise ValueError("a halo piece is a contiguous float64 CUDA tensor of shape [cols, rows]")
    return int(t.shape[1]), C.c_void_p(t.data_ptr())


class DeviceHalo:
    """Halo exchange of a SnowPlan's surface between neighbouring ranks without host staging: the 128 boundary rows are packed
    on the device, sent / received point-to-point as device tensors (RCCL when the backend is nccl; any other backend moves
    those rows — not the block — through the host) and handed to prepare_chunk_dev.  Buffers are allocated once."""

    def __init__(self, plan, rank: int, world: int, halo: int | None = None):
        import torch
        import torch.distributed as dist
        from .terrain import HALO
        self.plan, self.rank, self.world = plan, rank, world
        halo = HALO if halo is None else halo
        dev = torch.device("cuda", plan.device)
        counts = torch.zeros(world, dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        counts[rank] = plan.rows
        dist.all_reduce(counts)
        counts = [int(v) for v in counts.tolist()]
        mk = lambda h: torch.empty((plan.cols, h), dtype=torch.float64, device=dev)      # noqa: E731
        own = min(halo, plan.rows)
        self.send_n = mk(own) if rank > 0 else None
        self.send_s = mk(own) if rank < world - 1 else None
        self.recv_n = mk(min(halo, counts[rank - 1])) if rank > 0 else None
        self.recv_s = mk(min(halo, counts[rank + 1])) if rank < world - 1 else None
        self.on_device = dist.get_backend() == "nccl"

    def exchange(self):
        """-> (north piece, south piece) as device tensors (None at the raster's edge)"""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return None, None
        self.plan.pack_halo(self.send_n, self.send_s)
        st = (lambda t: t) if self.on_device else (lambda t: t.cpu())
        ops, rn, rs = [], None, None
        if self.rank > 0:
            rn = self.recv_n if self.on_device else torch.empty(self.recv_n.shape, dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, rn, self.rank - 1))
            ops.append(dist.P2POp(dist.isend, st(self.send_n), self.rank - 1))
        if self.rank < self.world - 1:
            rs = self.recv_s if self.on_device else torch.empty(self.recv_s.shape, dtype=torch.float64)
            ops.append(dist.P2POp(dist.isend, st(self.send_s), self.rank + 1))
            ops.append(dist.P2POp(dist.irecv, rs, self.rank + 1))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.on_device:
            torch.cuda.current_stream().synchronize()      # the library's kernels run on the null stream
        else:
            if rn is not None:
                self.recv_n.copy_(rn)
            if rs is not None:
                self.recv_s.copy_(rs)
            torch.cuda.synchronize()
        return self.recv_n, self.recv_s


def snowmodel1_chunks_tiled(plan, rank: int, world: int, *, exchange=None, allreduce=None) -> dict:
    """Drives one rank's SnowPlan through the chunk loop of a row-tiled raster: per chunk a halo exchange of
    the snow surface (terrain.exchange_halo, point-to-point between neighbouring ranks) and two (sum, count)
    all-reduces (the raster means of the surface and of tpic).  `exchange` / `allreduce` are injectable for
    tests; the defaults use torch.distributed (RCCL when the backend is nccl)."""
    from .terrain import exchange_halo
    from .distributed import allreduce_twi_mean
    if exchange is None:
        exchange = lambda blk: exchange_halo(blk, rank, world)          # noqa: E731
    if allreduce is None:
        allreduce = allreduce_twi_mean                                    # (sum, count) -> global mean
    for ch in range(plan.chunks):
        surf = plan.surface()
        ext, hn, hs = exchange(surf)
        s, n = plan.surface_partial()
        smean = allreduce(s, n)
        ts, tn = plan.prepare_chunk(ch, ext if (hn or hs) else None, hn, hs, smean)
        plan.run_chunk(ch, allreduce(ts, tn))
    return plan.result


# ---- `runmicro(..., snow = TRUE)` as one library call (include/mcf.h mcf_runmicrosnow1 / mcf_snowrun_*) -------------------
class SnowRun:
    """`.snowmodel1` + `.runmicrosnow1` (R/internal.R:2498-2619, 3581-3659) device-resident, staged:
    `pass1()` walks the snow model's chunk loop and returns the day classes, `pass2(micro inputs)` the merged microclimate.

    grid   the fifteen arguments of runmicro1Cpp for the WHOLE series (mapping with the names of api.runmicro1Cpp's parameters)
    snow   {"obstime", "climdata", "pointm", "vegp", "other", "snowenv", "dtm", "res", "tfact"[, "chunk_steps"]} as for
           `SnowPlan` / snowmodel1_chunks
    devices / n_blocks: row blocks over several devices from this one process (None: one block on `device`)."""

    def __init__(self, grid: Mapping, snow: Mapping, *, device: int = 0, devices=None, n_blocks: int = 0, cells_per_block: int = 0):
        self._marshal_only(grid, snow, device, cells_per_block)
        mu = None
        if devices is not None or n_blocks:
            mu = _abi.Multi()
            self._devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
            mu.n_devices, mu.devices, mu.n_blocks = int(self._devs.size), self._devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
        self._p = C.c_void_p()
        _abi.check(self._lib.mcf_snowrun_create(C.byref(self._in), C.byref(self._gm.options), C.byref(mu) if mu is not None else None,
                                                C.byref(self._p)))
        self.days = int(self._lib.mcf_snowrun_days(self._p))

    def _marshal_only(self, grid: Mapping, snow: Mapping, device: int, cells_per_block: int):
        from .marshal import alloc_outputs, marshal
        self._lib = _abi.load()
        g = grid
        # array weather (mcf_runmicrosnow2): `snow` carries "af_wind" (and optionally "wsa_s"), every weather array is [rows, cols, T]
        self.array_weather = "af_wind" in snow
        self._gm = marshal(g["obstime"], g["climdata"], g["pointm"], g["vegp"], g["soilc"], g["reqhgt"], g["zref"],
                           g["lats"] if "lats" in g else g["lat"], g["lons"] if "lons" in g else g["lon"],
                           g.get("Sminp", 0.0), g.get("Smaxp", 0.0), g["tfact"], g.get("complete", True), g.get("mat", 0.0),
                           g.get("out", (1,) * 10), self.array_weather, device, 0, cells_per_block, g.get("dfsel"))
        self._alloc_outputs = lambda: alloc_outputs(self._gm)
        R, Cc = np.shape(snow["vegp"]["pai"])
        oth = dict(snow["other"])
        for k, shp in (("slope", (R, Cc)), ("aspect", (R, Cc)), ("skyview", (R, Cc)), ("wsa", (R, Cc, 8)), ("hor", (R, Cc, 24))):
            oth.setdefault(k, np.zeros(shp))
        if self.array_weather:
            self._sm, self._din = _driver_in_array(snow["obstime"], snow["climdata"], snow["pointm"], snow["vegp"], oth,
                                                   snow.get("snowenv", "Alpine"), snow["dtm"], snow["res"], snow.get("tfact", 0.02),
                                                   snow["af_wind"], snow.get("wsa_s", 0), snow.get("chunk_steps", 120))
        else:
            self._sm = marshal_snow(snow["obstime"], snow["climdata"], snow["vegp"], oth, False, pointm=snow["pointm"],
                                    snowenv=snow.get("snowenv", "Alpine"))
            self._din = _abi.SnowDriverIn()
            self._din.base = self._sm.inputs
            self._din.dtm = self._sm.f64(snow["dtm"], (R, Cc), "dtm")
            self._din.res, self._din.tfact = float(snow["res"]), float(snow.get("tfact", 0.02))
            self._din.chunk_steps = int(snow.get("chunk_steps", 120))
        self._in = _abi.MicrosnowIn()
        self._in.grid = C.pointer(self._gm.inputs)
        self._in.snow = C.pointer(self._din)
        self._in.micro = None
        self._in.mat = 0.0
        self.rows, self.cols, self.tsteps = R, Cc, self._sm.tsteps
        self._mm = None

    def keep(self, gigabytes: float):
        """Keep pass 1's snow chunks in device memory for pass 2, up to this much (include/mcf.h mcf_snowrun_keep; 0: off).
        Pays from the handle's second period on — the sets are pooled — or when the snow series are fetched anyway."""
        _abi.check(self._lib.mcf_snowrun_keep(self._p, int(gigabytes * 2 ** 30)))

    def stats(self) -> dict:
        """What pass 2 was spared (include/mcf.h mcf_snowrun_stats)."""
        st = (C.c_int64 * 4)()
        _abi.check(self._lib.mcf_snowrun_stats(self._p, st))
        return dict(zip(("tile_days", "tile_days_left_out", "chunks_kept", "chunks_rerun"), (int(v) for v in st)))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self._lib.mcf_snowrun_destroy(self._p)
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

    def pass1(self, want_smod: bool = False):
        """-> (snowdays, nosnowdays[, smod]): one 0/1 flag per day; smod = `.snowmodel1`'s five [rows, cols, tsteps] arrays"""
        sd, nd = np.zeros(self.days, np.int32), np.zeros(self.days, np.int32)
        so, smod = None, None
        if want_smod:
            so, smod = _abi.SnowDriverOut(), {}
            for f in _abi.SNOWDRIVER_OUT:
                a = np.empty((self.rows, self.cols, self.tsteps), dtype=np.float64, order="F")
                smod[f] = a
                setattr(so, f, a.ctypes.data_as(_abi.c_double_p))
        _abi.check(self._lib.mcf_snowrun_pass1(self._p, C.byref(so) if so is not None else None, sd.ctypes.data_as(_abi.c_int32_p),
                                               nd.ctypes.data_as(_abi.c_int32_p)))
        return (sd, nd, smod) if want_smod else (sd, nd)

    def pass2(self, micro: Mapping | None, mat: float) -> dict:
        """micro = {"obstime", "climdata" (with umu), "vegp" (`.sortl2`), "other" (bare-ground terrain, lat, lon, zref, Smax)} for
        the WHOLE series, or None when the year has no snow day -> the ten merged outputs"""
        mi = None
        if micro is not None:
            self._mm = marshal_snow(micro["obstime"], micro["climdata"], micro["vegp"], micro["other"], self.array_weather, micro=True)
            mi = C.byref(self._mm.inputs)
        outs, arrays = self._alloc_outputs()
        _abi.check(self._lib.mcf_snowrun_pass2(self._p, mi, float(mat), C.byref(outs)))
        return arrays


def runmicrosnow1(grid: Mapping, snow: Mapping, micro: Mapping | None, mat: float, *, device: int = 0, devices=None, n_blocks: int = 0,
                  want_smod: bool = False, cells_per_block: int = 0):
    """mcf_runmicrosnow1 / mcf_runmicrosnow1_multi: the whole snow run as ONE library call (arguments as `SnowRun`, `micro` as
    `SnowRun.pass2`) -> the merged outputs[, smod]"""
    from .marshal import alloc_outputs
    with SnowRun.__new__(SnowRun) as run:
        run._p = None
        SnowRun._marshal_only(run, grid, snow, device, cells_per_block)
        mi = None
        if micro is not None:
            run._mm = marshal_snow(micro["obstime"], micro["climdata"], micro["vegp"], micro["other"], run.array_weather, micro=True)
            run._in.micro = C.pointer(run._mm.inputs)
        run._in.mat = float(mat)
        outs, arrays = alloc_outputs(run._gm)
        so, smod = None, None
        if want_smod:
            so, smod = _abi.SnowDriverOut(), {}
            for f in _abi.SNOWDRIVER_OUT:
                a = np.empty((run.rows, run.cols, run.tsteps), dtype=np.float64, order="F")
                smod[f] = a
                setattr(so, f, a.ctypes.data_as(_abi.c_double_p))
        sop = C.byref(so) if so is not None else None
        lib = _abi.load()
        if devices is not None or n_blocks:
            mu = _abi.Multi()
            devs = np.ascontiguousarray([] if devices is None else list(devices), dtype=np.int32)
            mu.n_devices, mu.devices, mu.n_blocks = int(devs.size), devs.ctypes.data_as(_abi.c_int32_p), int(n_blocks)
            _abi.check(lib.mcf_runmicrosnow1_multi(C.byref(run._in), C.byref(run._gm.options), C.byref(mu), C.byref(outs), sop))
        else:
            fn = lib.mcf_runmicrosnow2 if run.array_weather else lib.mcf_runmicrosnow1
            _abi.check(fn(C.byref(run._in), C.byref(run._gm.options), C.byref(outs), sop))
    return (arrays, smod) if want_smod else arrays


def runmicrosnow2(grid: Mapping, snow: Mapping, micro: Mapping | None, mat: float, **kw):
    """mcf_runmicrosnow2: `.snowmodel2`'s loop + `.runmicrosnow2` as ONE library call.  `grid` = runmicro2Cpp's arguments for the
    whole series (array climate / point-model arrays, "lats", "lons"), `snow` as for snowmodel2_device (with "af_wind"[, "wsa_s"]),
    `micro` = gridmicrosnow2's inputs for the whole series -> the merged outputs[, smod]"""
    if "af_wind" not in snow:
        raise ValueError("runmicrosnow2: snow['af_wind'] (the chunk wind series of `.snowmodel2`) is missing")
    return runmicrosnow1(grid, snow, micro, mat, **kw)
