"""Host-side mirror of the reference's point-model operators: `soilmCpp`, `BigLeafCpp`,
`pointmprocess`, `weatherhgtCpp` with the argument lists of R/RcppExports.R, plus `runpointmodel_chain`, the way
`runpointmodel` strings them together into the grid solver's `pointm` (R/Cppwrappers.R:119-138).

They are O(tsteps) serial series for one point and run on the host inside libmcfhip (mcf_pointmodel.cpp), as
they run on the host in the reference; the grid solver itself has no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping

import numpy as np

from . import _abi


def _vec(a, n=None, name="?"):
    v = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if n is not None and v.shape != (n,):
        raise ValueError(f"{name}: expected length {n}")
    return v


def _weather(climdata: Mapping, n: int, need_precip: bool):
    w = _abi.PointWeather()
    keep = []
    for f in _abi.POINT_WEATHER_FIELDS:
        if f == "precip" and not need_precip and "precip" not in climdata:
            setattr(w, f, None)
            continue
        v = _vec(climdata[f], n, f"climdata${f}")
        keep.append(v)
        setattr(w, f, v.ctypes.data_as(_abi.c_double_p))
    return w, keep


def _obstime(obstime: Mapping, n: int):
    t = _abi.Obstime()
    keep = []
    for f in ("year", "month", "day"):
        v = np.ascontiguousarray(np.asarray(obstime[f]).astype(np.int32))
        keep.append(v)
        setattr(t, f, v.ctypes.data_as(_abi.c_int32_p))
    h = _vec(obstime["hour"], n, "obstime$hour")
    keep.append(h)
    t.hour = h.ctypes.data_as(_abi.c_double_p)
    return t, keep


def BigLeafCpp(obstime, climdata, vegp, groundp, soilm, lat, lon, dTmx=25.0, zref=2.0, maxiter=100, bwgt=0.5,
               tol=0.5, gmn=0.1, yearG=True) -> dict:
    """Drop-in for the reference's BigLeafCpp (src/microclimfCpp.cpp:710-881).  `gmn` is accepted and, as in the
    reference, unused."""
    lib = _abi.load()
    n = len(np.asarray(climdata["temp"]))
    t, k1 = _obstime(obstime, n)
    w, k2 = _weather(climdata, n, False)
    vp, gp, sm = _vec(vegp), _vec(groundp), _vec(soilm, n, "soilm")
    if vp.size < 9 or gp.size < 12:
        raise ValueError("vegp needs >= 9 and groundp 12 entries")
    out = _abi.BigLeafOut()
    res = {}
    for f in _abi.BIGLEAF_FIELDS:
        res[f] = np.zeros(n)
        setattr(out, f, res[f].ctypes.data_as(_abi.c_double_p))
    _abi.check(lib.mcf_bigleaf(n, C.byref(t), C.byref(w), vp.ctypes.data_as(_abi.c_double_p),
                               gp.ctypes.data_as(_abi.c_double_p), sm.ctypes.data_as(_abi.c_double_p), float(lat),
                               float(lon), float(dTmx), float(zref), int(maxiter), float(bwgt), float(tol),
                               1 if yearG else 0, C.byref(out)))
    res["err"] = out.err
    res["iters"] = out.iters
    return res


def soilmCpp(climdata, rmu, mult, pwr, Smax, Smin, Ksat, a) -> np.ndarray:
    """Drop-in for soilmCpp (src/microclimfCpp.cpp:931-972): daily soil moisture of the two-layer bucket model."""
    lib = _abi.load()
    n = len(np.asarray(climdata["temp"]))
    w, keep = _weather({**{f: np.zeros(n) for f in ("relhum", "pres", "difrad", "windspeed")}, **climdata}, n, True)
    out = np.zeros(max(n // 24, 1))
    nd = C.c_int64()
    _abi.check(lib.mcf_soilm(n, C.byref(w), float(rmu), float(mult), float(pwr), float(Smax), float(Smin), float(Ksat),
                             float(a), out.ctypes.data_as(_abi.c_double_p), C.byref(nd)))
    return out[:nd.value]


def pointmprocess(pointvars, zref, h, pai, rho, Vm, Vq, Mc) -> dict:
    """Drop-in for pointmprocess (src/microclimfCpp.cpp:5265-5323); `pointvars` has windspeed, tc, rh, pk, uf,
    soilm, RabsG."""
    lib = _abi.load()
    n = len(np.asarray(pointvars["tc"]))
    ins = [_vec(pointvars[k], n, k) for k in ("windspeed", "tc", "rh", "pk", "uf", "soilm", "RabsG")]
    res = {k: np.zeros(n) for k in ("umu", "kp", "muGp", "DDp", "T0p", "dtrp")}
    _abi.check(lib.mcf_pointmprocess(n, *[v.ctypes.data_as(_abi.c_double_p) for v in ins], float(zref), float(h),
                                     float(pai), float(rho), float(Vm), float(Vq), float(Mc),
                                     *[res[k].ctypes.data_as(_abi.c_double_p) for k in ("umu", "kp", "muGp", "DDp", "T0p",
                                                                                        "dtrp")]))
    return res


def weatherhgtCpp(obstime, climdata, zin, uzin, zout, lat, lon) -> dict:
    """Drop-in for weatherhgtCpp (src/microclimfCpp.cpp:884-929): a copy of `climdata` with temp, relhum and
    windspeed moved from zin / uzin to zout."""
    lib = _abi.load()
    n = len(np.asarray(climdata["temp"]))
    t, k1 = _obstime(obstime, n)
    w, k2 = _weather(climdata, n, False)
    res = {k: np.zeros(n) for k in ("temp", "relhum", "windspeed")}
    _abi.check(lib.mcf_weatherhgt(n, C.byref(t), C.byref(w), float(zin), float(uzin), float(zout), float(lat), float(lon),
                                  *[res[k].ctypes.data_as(_abi.c_double_p) for k in ("temp", "relhum", "windspeed")]))
    out = {k: np.array(v, dtype=np.float64, copy=True) for k, v in climdata.items()}
    out.update(res)
    return out


def pointmodelsnow(obstime, climdata, vegp, other, snowenv, tol: float = 0.5, maxiter: float = 100) -> dict:
    """Drop-in for pointmodelsnow (src/microclimfCpp.cpp:4000-4169): vegp = (pai, hgt, ltra, clump), other = (slope,
    aspect, lat, lon, zref, initial snow depth, initial snow age), snowenv a name as in the reference."""
    lib = _abi.load()
    n = len(np.asarray(climdata["temp"]))
    t, k1 = _obstime(obstime, n)
    w, k2 = _weather(climdata, n, True)
    vp, ot = _vec(vegp), _vec(other)
    if vp.size < 4 or ot.size < 7:
        raise ValueError("vegp needs 4 and other 7 entries")
    out = _abi.PointSnowOut()
    res = {}
    for f in _abi.POINTSNOW_FIELDS:
        res[f] = np.zeros(n + 1 if f in ("sdepc", "sdepg") else n)
        setattr(out, f, res[f].ctypes.data_as(_abi.c_double_p))
    env = lib.mcf_snowenv_from_name(str(snowenv).encode())
    _abi.check(lib.mcf_pointmodelsnow(n, C.byref(t), C.byref(w), vp.ctypes.data_as(_abi.c_double_p),
                                      ot.ctypes.data_as(_abi.c_double_p), env, float(tol), float(maxiter), C.byref(out)))
    res["mxdif"] = out.mxdif
    res["iters"] = out.iters
    return res


def manCpp(x, n: int) -> np.ndarray:
    """Drop-in for manCpp (src/microclimfCpp.cpp:597-627)."""
    lib = _abi.load()
    v = _vec(x)
    out = np.zeros(len(v))
    _abi.check(lib.mcf_man(len(v), v.ctypes.data_as(_abi.c_double_p), int(n), out.ctypes.data_as(_abi.c_double_p)))
    return out


def runpointmodel_chain(obstime, weather, vegp_p, groundp_p, lat, lon, zref=2.0, soilparams=None, soilm=None,
                        dTmx=25.0, maxiter=100, yearG=True) -> dict:
    """The chain of `runpointmodel` (R/Cppwrappers.R:117-138) from a weather table to the grid solver's `pointm`:
    wind floor 0.5 m/s, soilmCpp (unless `soilm` is given; its daily values are interpolated LINEARLY to hours —
    R uses stats::spline), BigLeafCpp, pointmprocess.  Returns {"pointm": ..., "bigleaf": ..., "weather": ...}."""
    w = {k: np.array(v, dtype=np.float64, copy=True) for k, v in weather.items()}
    n = len(w["temp"])
    w["windspeed"] = np.maximum(w["windspeed"], 0.5)
    if soilm is None:
        if soilparams is None:
            raise ValueError("give soilm or soilparams (rmu, mult, pwr, Smax, Smin, Ksat, a)")
        sd = soilmCpp(w, **soilparams)
        soilm = np.interp(np.linspace(0, max(len(sd) - 1, 0), n), np.arange(len(sd)), sd)
    soilm = _vec(soilm, n, "soilm")
    bl = BigLeafCpp(obstime, w, vegp_p, groundp_p, soilm, lat, lon, dTmx, zref, maxiter, 0.5, 0.5, 0.1, yearG)
    pv = {"windspeed": w["windspeed"], "tc": w["temp"], "rh": w["relhum"], "pk": w["pres"], "uf": bl["uf"],
          "soilm": soilm, "RabsG": bl["RabsG"]}
    pp = pointmprocess(pv, zref, vegp_p[0], vegp_p[1], groundp_p[4], groundp_p[5], groundp_p[6], groundp_p[7])
    pointm = {"soilm": soilm, "Tg": bl["Tg"], "T0p": pp["T0p"], "Tbp": np.zeros(n), "G": bl["G"], "DDp": pp["DDp"],
              "umu": pp["umu"], "kp": pp["kp"], "muGp": pp["muGp"], "dtrp": pp["dtrp"]}
    return {"pointm": pointm, "bigleaf": bl, "weather": w}
