"""pointm from the point-model chain (TEST INFRASTRUCTURE): soilmCpp -> BigLeafCpp -> pointmprocess, the way
`runpointmodel` assembles the grid solver's `pointm` (R/Cppwrappers.R:119-138), on top of the oracle's
restatements (oracle/pointmodel.c).  Used to give parity cases point-model inputs that come from the physics
instead of the SURVEY §8d synthetic recipe.  One deliberate simplification, because it only shapes a test
input: R expands the daily soil moisture with stats::spline (FMM cubic); here it is interpolated linearly.
BigLeafCpp runs with yearG = FALSE: its annual ground-heat-flux cycle takes a 91-day circular mean (mayCpp ->
maCpp, cpp:561-594) that indexes out of bounds for series shorter than 91 days, in the reference as in the oracle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as O
from .replay_reference_tests import bigleaf

DP = C.POINTER(C.c_double)

# soilparamsp row "Loam"-like constants of the reference's test-soilmCpp.R call
SOILM_PARAMS = dict(rmu=0.021303, mult=0.000191202, pwr=1.134773, Smax=0.419, Smin=0.091, Ksat=5.89, a=0.059765)
# vegp_p = (h, pai, x, clump, lref, ltra, leafd, em, gsmax, q50); groundp_p = (gref, slope, aspect, em, rho, Vm, Vq,
# Mc, b, psi_e, Smax, Smin) — positional, as BigLeafCpp reads them (cpp:717-740)
VEGP_P = np.array([0.5, 2.0, 1.0, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100.0])
GROUNDP_P = np.array([0.15, 0.0, 180.0, 0.97, 1.53, 0.509, 0.06, 0.5422, 5.2, -5.6, 0.42, 0.074])


def _d(a):
    return a.ctypes.data_as(DP)


def pointm_chain(obstime, weather, lat, lon, zref=2.0, vegp_p=VEGP_P, groundp_p=GROUNDP_P, maxiter=100):
    """weather: temp, relhum, pres, swdown, difrad, lwdown, windspeed, precip (hourly vectors).  Returns the
    pointm dict of the grid solver (soilm, Tg, T0p, Tbp, G, DDp, umu, kp, muGp, dtrp) and BigLeafCpp's err."""
    lib = O.load()
    n = len(weather["temp"])
    w = {k: np.ascontiguousarray(np.asarray(v, dtype=np.float64)) for k, v in weather.items()}
    w["windspeed"] = np.maximum(w["windspeed"], 0.5)                    # R/Cppwrappers.R:118
    nd = n // 24
    sd = np.zeros(max(nd, 1))
    lib.orc_soilm.restype = C.c_int
    p = SOILM_PARAMS
    lib.orc_soilm(C.c_int(n), _d(w["temp"]), _d(w["swdown"]), _d(w["lwdown"]), _d(w["precip"]), C.c_double(p["rmu"]),
                  C.c_double(p["mult"]), C.c_double(p["pwr"]), C.c_double(p["Smax"]), C.c_double(p["Smin"]),
                  C.c_double(p["Ksat"]), C.c_double(p["a"]), _d(sd))
    soilm = np.interp(np.linspace(0, max(nd - 1, 0), n), np.arange(max(nd, 1)), sd)
    obst = {"year": np.ascontiguousarray(obstime["year"], dtype=np.int32),
            "month": np.ascontiguousarray(obstime["month"], dtype=np.int32),
            "day": np.ascontiguousarray(obstime["day"], dtype=np.int32),
            "hour": np.ascontiguousarray(obstime["hour"], dtype=np.float64)}
    bl = bigleaf(obst, w, np.ascontiguousarray(vegp_p), np.ascontiguousarray(groundp_p), np.ascontiguousarray(soilm),
                 float(lat), float(lon), 25.0, float(zref), int(maxiter), 0.5, 0.5, 0.1, False)
    out = {k: np.zeros(n) for k in ("umu", "kp", "muGp", "DDp", "T0p", "dtrp")}
    lib.orc_pointmprocess.restype = None
    lib.orc_pointmprocess(C.c_int(n), _d(w["windspeed"]), _d(w["temp"]), _d(w["relhum"]), _d(w["pres"]), _d(bl["uf"]),
                          _d(soilm), _d(bl["RabsG"]), C.c_double(zref), C.c_double(vegp_p[0]), C.c_double(vegp_p[1]),
                          C.c_double(groundp_p[4]), C.c_double(groundp_p[5]), C.c_double(groundp_p[6]),
                          C.c_double(groundp_p[7]), _d(out["umu"]), _d(out["kp"]), _d(out["muGp"]), _d(out["DDp"]),
                          _d(out["T0p"]), _d(out["dtrp"]))
    pointm = {"soilm": soilm, "Tg": bl["Tg"], "T0p": out["T0p"], "Tbp": np.zeros(n), "G": bl["G"], "DDp": out["DDp"],
              "umu": out["umu"], "kp": out["kp"], "muGp": out["muGp"], "dtrp": out["dtrp"]}
    return pointm, bl["err"]
