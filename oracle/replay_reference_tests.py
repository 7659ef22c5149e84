"""Replay of the reference's own testthat files against the CPU oracle
(TEST INFRASTRUCTURE).

The reference holds no golden vectors for the grid solver; the only tests that
touch its arithmetic assert interval bounds:

  tests/testthat/test-microclimatemodel_wrapper.R   (twostream*, windCpp, TVaboveground, manCpp ...)
  tests/testthat/test-BigLeafCpp.R                  (the point model that feeds it)
  tests/testthat/test-pointmodelsnow.R              (snowoneB, radoneB, canopysnowintCpp, snowalbCpp,
                                                     GFluxCppsnow: the snow branch's arithmetic)
  tests/testthat/test-weatherhgtCpp.R               (BigLeafCpp again, zeroplanedisCpp, roughlengthCpp)
  tests/testthat/test-soilmCpp.R                    (the point soil-moisture model that feeds pointm$soilm)

This module rebuilds their inputs line for line (R -> numpy), runs the oracle's
restatement of the same functions and evaluates every `expect_*` of the two
files.  Each check is returned as (label, ok, detail) so that the pytest wrapper
can report them one by one.  Vector arguments are passed POSITIONALLY, as the
reference's C++ reads them (it ignores the R names: e.g. the wrapper test's
`groundp` lists Smin before Smax while the C++ reads [10] as Smax).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import oracle as O

DP = C.POINTER(C.c_double)
IP = C.POINTER(C.c_int)


class BigLeafOut(C.Structure):
    _fields_ = [(k, DP) for k in ("Tc", "Tg", "H", "G", "psih", "psim", "phih", "OL", "uf", "RabsG", "albedo")] \
        + [("err", C.c_double), ("iters", C.c_int)]


class WrapperOut(C.Structure):
    _fields_ = [(k, DP) for k in ("Tz", "tleaf", "rh", "uz", "Rdirdown", "Rdifdown", "Rswup", "Rlwdown",
                                  "Rlwup", "soilm")]


class PointSnowOut(C.Structure):
    _fields_ = [(k, DP) for k in ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "sdeng", "G", "RswabsG", "RlwabsG", "tr",
                                  "umu", "sublmelt", "tempmelt", "rainmelt", "sstemp")] \
        + [("mxdif", C.c_double), ("iters", C.c_int)]


LAST_POINTSNOW = {}     # output of the last replay (frozen in tests/golden/pointmodelsnow_test.npz)
SNOWENV = {"Alpine": 0, "Maritime": 1, "Prairie": 2, "Tundra": 3, "Taiga": 4}


def pointmodelsnow(obst, clim, vegp, other, snowenv, tol=0.5, maxiter=100):
    """orc_pointmodelsnow (cpp:4000-4169); returns dict of arrays + mxdif, iters."""
    lib = _lib()
    n = len(clim["temp"])
    out = PointSnowOut()
    arrs = {}
    for k, _ in PointSnowOut._fields_[:15]:
        arrs[k] = np.zeros(n + 1 if k in ("sdepc", "sdepg") else n)
        setattr(out, k, _d(arrs[k]))
    vegp = np.ascontiguousarray(vegp, dtype=np.float64)
    other = np.ascontiguousarray(other, dtype=np.float64)
    lib.orc_pointmodelsnow.restype = C.c_int
    lib.orc_pointmodelsnow(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                           _d(clim["temp"]), _d(clim["relhum"]), _d(clim["pres"]), _d(clim["swdown"]),
                           _d(clim["difrad"]), _d(clim["lwdown"]), _d(clim["windspeed"]), _d(clim["precip"]),
                           _d(vegp), _d(other), C.c_int(SNOWENV.get(snowenv, 0)), C.c_double(tol),
                           C.c_double(maxiter), C.byref(out))
    arrs["mxdif"] = out.mxdif
    arrs["iters"] = out.iters
    return arrs


def _d(a):
    return a.ctypes.data_as(DP)


def _i(a):
    return a.ctypes.data_as(IP)


def _lib():
    lib = O.load()
    lib.orc_bigleaf.restype = C.c_int
    lib.orc_wrapper.restype = C.c_int
    lib.orc_clearskyrad.restype = None
    lib.orc_solpositionv.restype = None
    return lib


def bigleaf(obst, clim, vegp, groundp, soilm, lat, lon, dTmx, zref, maxiter, bwgt, tol, gmn, yearG):
    lib = _lib()
    n = len(clim["temp"])
    out = BigLeafOut()
    arrs = {}
    for k, _ in BigLeafOut._fields_[:11]:
        arrs[k] = np.zeros(n)
        setattr(out, k, _d(arrs[k]))
    lib.orc_bigleaf(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                    _d(clim["temp"]), _d(clim["relhum"]), _d(clim["pres"]), _d(clim["swdown"]),
                    _d(clim["difrad"]), _d(clim["lwdown"]), _d(clim["windspeed"]), _d(vegp), _d(groundp),
                    _d(soilm), C.c_double(lat), C.c_double(lon), C.c_double(dTmx), C.c_double(zref),
                    C.c_int(maxiter), C.c_double(bwgt), C.c_double(tol), C.c_int(1 if yearG else 0), C.byref(out))
    arrs["err"] = out.err
    arrs["iters"] = out.iters
    return arrs


def wrapper(obst, clim, BL, vegp, groundp, reqhgt, zref, lat, lon):
    lib = _lib()
    n = len(clim["temp"])
    out = WrapperOut()
    arrs = {}
    for k, _ in WrapperOut._fields_:
        arrs[k] = np.full(n, np.nan)
        setattr(out, k, _d(arrs[k]))
    lib.orc_wrapper(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                    _d(clim["temp"]), _d(clim["relhum"]), _d(clim["pres"]), _d(clim["swdown"]),
                    _d(clim["difrad"]), _d(clim["lwdown"]), _d(clim["windspeed"]), _d(BL["Tg"]), _d(BL["G"]),
                    _d(BL["uf"]), _d(vegp), _d(groundp), C.c_double(reqhgt), C.c_double(zref), C.c_double(lat),
                    C.c_double(lon), C.byref(out))
    return arrs


def _forcing(year, month, day, ea_from_mean):
    """Inputs common to both test files (test-microclimatemodel_wrapper.R:2-40,
    test-BigLeafCpp.R:3-33)."""
    lib = _lib()
    hrs = np.arange(24, dtype=np.float64)
    n = 24
    obst = {"year": np.full(n, year, dtype=np.int32), "month": np.full(n, month, dtype=np.int32),
            "day": np.full(n, day, dtype=np.int32), "hour": hrs.copy()}
    Tair = 10 + 5 * np.sin((hrs - 8) / 24 * 2 * np.pi)
    satv = np.array([lib.orc_satvap(float(t)) for t in Tair])
    if ea_from_mean:
        ea = 0.7 * lib.orc_satvap(float(np.mean(Tair)))
        RH = ea / satv * 100
    else:
        RH = np.full(n, 70.0)
    Pk = np.full(n, 101.3)
    return hrs, n, obst, Tair, RH, Pk


def replay_wrapper_test():
    lib = _lib()
    checks = []

    def ck(label, ok, detail=""):
        checks.append((label, bool(ok), str(detail)))

    hrs, n, obst, Tair, RH, Pk = _forcing(2024, 3, 21, True)
    csr = np.zeros(n)
    lib.orc_clearskyrad(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                        C.c_double(50.0), C.c_double(-5.0), _d(Tair), _d(RH), _d(Pk), _d(csr))
    zen, azi, si = np.zeros(n), np.zeros(n), np.zeros(n)
    lib.orc_solpositionv(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                         C.c_double(50.0), C.c_double(-5.0), C.c_double(0.0), C.c_double(180.0), _d(zen), _d(azi),
                         _d(si))
    Rdir = 0.3 * csr * si
    SWd = 0.5 * csr
    Rdif = SWd - Rdir
    LWd = np.full(n, 350.0)
    U2 = np.full(n, 2.0)
    clim = {"temp": Tair, "relhum": RH, "pres": Pk, "swdown": SWd, "difrad": Rdif, "lwdown": LWd,
            "windspeed": U2, "winddir": np.full(n, 180.0), "precip": np.zeros(n)}
    vegp = np.array([0.5, 2, 1.0, 0.1, 0.4, 0.2, 0.05, 0.97, 0.13])
    groundp = np.array([0.15, 0, 180, 0.97, 1.53, 0.509, 0.06, 0.5422, 5.2, 2.6, 0.091, 0.419])
    BL = bigleaf(obst, clim, vegp, groundp, np.full(24, 0.3), 50.0, -5.0, 25.0, 2.0, 100, 0.5, 0.5, 0.1, False)

    def ratio(num, den, fill):
        with np.errstate(divide="ignore", invalid="ignore"):
            r = num / den
        r[den == 0] = fill
        return r

    # ---- reqhgt = 0.05 (test-microclimatemodel_wrapper.R:52-90)
    o = wrapper(obst, clim, BL, vegp, groundp, 0.05, 2.0, 50.0, -5.0)
    vals = [o[k] for k in ("Tz", "tleaf", "rh", "uz", "Rdirdown", "Rdifdown", "Rswup", "Rlwdown", "Rlwup")]
    ck("above: all finite", all(np.isfinite(v).all() for v in vals))
    ck("above: max|Tz-Tair| <= 5", np.max(np.abs(o["Tz"] - Tair)) <= 5, np.max(np.abs(o["Tz"] - Tair)))
    ck("above: max|tleaf-Tair| <= 2", np.max(np.abs(o["tleaf"] - Tair)) <= 2, np.max(np.abs(o["tleaf"] - Tair)))
    ck("above: min(rh) >= min(RH)-5", o["rh"].min() >= RH.min() - 5, o["rh"].min())
    ck("above: max(rh) <= 100", o["rh"].max() <= 100, o["rh"].max())
    uzr = o["uz"] / U2
    ck("above: uz/U2 in [0.08, 0.1]", uzr.min() >= 0.08 and uzr.max() <= 0.1, (uzr.min(), uzr.max()))
    r = ratio(o["Rdirdown"] * si, Rdir, 0.2)
    ck("above: Rdir ratio in [0, 0.25]", r.min() >= 0 and r.max() <= 0.25, (r.min(), r.max()))
    r = ratio(o["Rdifdown"], Rdif, 0.3)
    ck("above: Rdif ratio in [0.27, 0.32]", r.min() >= 0.27 and r.max() <= 0.32, (r.min(), r.max()))
    r = ratio(o["Rswup"], SWd, 0.112)
    ck("above: Rswup ratio in [0.03, 0.15]", r.min() >= 0.03 and r.max() <= 0.15, (r.min(), r.max()))
    r = o["Rlwdown"] / LWd
    ck("above: Rlwdown/LWd in [0.94, 1.2]", r.min() >= 0.94 and r.max() <= 1.2, (r.min(), r.max()))
    r = o["Rlwup"] / LWd
    ck("above: Rlwup/LWd in [0.94, 1.2]", r.min() >= 0.94 and r.max() <= 1.2, (r.min(), r.max()))
    # ---- reqhgt = 0 (:92-123)
    o = wrapper(obst, clim, BL, vegp, groundp, 0.0, 2.0, 50.0, -5.0)
    vals = [o[k] for k in ("Tz", "soilm", "Rdirdown", "Rdifdown", "Rswup", "Rlwdown", "Rlwup")]
    ck("ground: all finite", all(np.isfinite(v).all() for v in vals))
    ck("ground: max|Tz-Tair| <= 5", np.max(np.abs(o["Tz"] - Tair)) <= 5, np.max(np.abs(o["Tz"] - Tair)))
    ck("ground: soilm in [0.299, 0.301]", o["soilm"].min() >= 0.299 and o["soilm"].max() <= 0.301)
    r = ratio(o["Rdirdown"] * si, Rdir, 0.2)
    ck("ground: Rdir ratio in [0, 0.25]", r.min() >= 0 and r.max() <= 0.25, (r.min(), r.max()))
    r = ratio(o["Rdifdown"], Rdif, 0.238)
    ck("ground: Rdif ratio in [0.2, 0.3]", r.min() >= 0.2 and r.max() <= 0.3, (r.min(), r.max()))
    r = ratio(o["Rswup"], SWd, 0.148)
    ck("ground: Rswup ratio in [0.03, 0.15]", r.min() >= 0.03 and r.max() <= 0.15, (r.min(), r.max()))
    r = o["Rlwdown"] / LWd
    ck("ground: Rlwdown/LWd in [0.9, 1.1]", r.min() >= 0.9 and r.max() <= 1.1, (r.min(), r.max()))
    r = o["Rlwup"] / LWd
    ck("ground: Rlwup/LWd in [0.9, 1.15]", r.min() >= 0.9 and r.max() <= 1.15, (r.min(), r.max()))
    # ---- reqhgt = -0.05 (:125-138)
    o = wrapper(obst, clim, BL, vegp, groundp, -0.05, 2.0, 50.0, -5.0)
    ck("below: finite", np.isfinite(o["Tz"]).all() and np.isfinite(o["soilm"]).all())
    ck("below: Tz in [8, 9]", o["Tz"].min() >= 8 and o["Tz"].max() <= 9, (o["Tz"].min(), o["Tz"].max()))
    ck("below: soilm in [0.299, 0.301]", o["soilm"].min() >= 0.299 and o["soilm"].max() <= 0.301)
    return checks, {"BL_err": BL["err"], "BL_iters": BL["iters"]}


def replay_bigleaf_test():
    """tests/testthat/test-BigLeafCpp.R."""
    checks = []

    def ck(label, ok, detail=""):
        checks.append((label, bool(ok), str(detail)))

    # test-BigLeafCpp.R:3-21
    hrs, n, obst, Tair, RH, Pk = _forcing(2024, 3, 21, True)
    SWd = np.maximum(0, 600 * np.sin((hrs - 6) / 12 * np.pi))
    Rdif = np.minimum(SWd, 0.3 * SWd)
    LWd = np.full(n, 350.0)
    U2 = np.full(n, 2.0)
    clim = {"temp": Tair, "relhum": RH, "pres": Pk, "swdown": SWd, "difrad": Rdif, "lwdown": LWd,
            "windspeed": U2, "winddir": np.full(n, 180.0), "precip": np.zeros(n)}
    vegp = np.array([0.5, 2.0, 1.0, 0.1, 0.4, 0.2, 0.05, 0.97, 0.33, 100])
    groundp = np.array([0.15, 0, 180, 0.97, 1.53, 0.509, 0.06, 0.5422, 5.2, -5.6, 0.42, 0.074])
    out = bigleaf(obst, clim, vegp, groundp, np.full(n, 0.3), 50.0, -5.0, 25, 2, 50, 0.5, 0.5, 0.1, False)
    Tc, Tg, H, G, RabsG, alb, uf = (out[k] for k in ("Tc", "Tg", "H", "G", "RabsG", "albedo", "uf"))
    nums = np.concatenate([out[k] for k in ("Tc", "Tg", "H", "G", "RabsG", "psih", "psim", "phih", "OL", "uf",
                                            "albedo")] + [np.array([out["err"]])])
    ck("bigleaf: all finite", np.isfinite(nums).all())
    ck("bigleaf: Tc bounds", Tc.min() >= Tair.min() - 5 and Tc.max() <= Tair.max() + 10, (Tc.min(), Tc.max()))
    ck("bigleaf: Tg bounds", Tg.min() >= Tair.min() - 5 and Tg.max() <= Tair.max() + 10, (Tg.min(), Tg.max()))
    ck("bigleaf: albedo in [0.01, 0.99]", alb.min() >= 0.01 and alb.max() <= 0.99)
    ck("bigleaf: uf in [2e-4, max U2]", uf.min() >= 2e-4 and uf.max() <= U2.max(), (uf.min(), uf.max()))
    Lwup = 5.67e-8 * (Tc + 273.15) ** 4
    Rnet = (1 - alb) * SWd + 0.97 * (LWd - Lwup)
    ck("bigleaf: H range", H.min() >= (Rnet + G - 20).min() and H.max() <= (Rnet + G - 20).max(),
       (H.min(), H.max(), (Rnet + G - 20).min(), (Rnet + G - 20).max()))
    ck("bigleaf: G range", G.min() >= (Rnet - 20).min() and G.max() <= (Rnet + 20).max())
    ck("bigleaf: RabsG range", RabsG.min() >= 0.5 * LWd.min() and RabsG.max() <= 1000)
    ck("bigleaf: psih in [-4, 3]", out["psih"].min() >= -4 and out["psih"].max() <= 3)
    ck("bigleaf: psim in [-4, 3]", out["psim"].min() >= -4 and out["psim"].max() <= 3)
    ck("bigleaf: err < 0.5", out["err"] < 0.5, out["err"])
    return checks, {"err": out["err"], "iters": out["iters"]}


def replay_pointmodelsnow_test():
    """tests/testthat/test-pointmodelsnow.R."""
    lib = _lib()
    checks = []

    def ck(label, ok, detail=""):
        checks.append((label, bool(ok), str(detail)))

    # test-pointmodelsnow.R:2-41
    hrs = np.arange(24, dtype=np.float64)
    n = 24
    obst = {"year": np.full(n, 2024, dtype=np.int32), "month": np.full(n, 3, dtype=np.int32),
            "day": np.full(n, 21, dtype=np.int32), "hour": hrs.copy()}
    Tair = -5 + 5 * np.sin((hrs - 8) / 24 * 2 * np.pi)
    ea = 0.7 * lib.orc_satvap(float(np.mean(Tair)))
    RH = np.array([ea / lib.orc_satvap(float(t)) * 100 for t in Tair])
    Pk = np.full(n, 101.3)
    csr = np.zeros(n)
    lib.orc_clearskyrad(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                        C.c_double(50.0), C.c_double(-5.0), _d(Tair), _d(RH), _d(Pk), _d(csr))
    zen, azi, si = np.zeros(n), np.zeros(n), np.zeros(n)
    lib.orc_solpositionv(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                         C.c_double(50.0), C.c_double(-5.0), C.c_double(0.0), C.c_double(180.0), _d(zen), _d(azi),
                         _d(si))
    Rdir = 0.3 * csr * si
    SWd = 0.5 * csr
    Rdif = SWd - Rdir
    LWd = np.full(n, 350.0)
    U2 = np.full(n, 2.0)
    Prec = np.full(n, 1.0)
    clim = {"temp": Tair, "relhum": RH, "pres": Pk, "swdown": SWd, "difrad": Rdif, "lwdown": LWd,
            "windspeed": U2, "winddir": np.full(n, 180.0), "precip": Prec}
    vegp = np.array([2, 0.5, 0.05, 0])          # pai, hgt, ltra, clump
    other = np.array([0, 180, 50, -5, 2, 0, 0])  # slope, aspect, lat, lon, zref, isnowd, isnowa
    pm = pointmodelsnow(obst, clim, vegp, other, "Taiga")
    LAST_POINTSNOW.clear()
    LAST_POINTSNOW.update({k: np.array(v) for k, v in pm.items()})
    Tc, Tg, sdepc, sdepg, sdenc, sdeng = (pm[k] for k in ("Tc", "Tg", "sdepc", "sdepg", "sdenc", "sdeng"))
    # :52-74 (R indices are 1-based: sdepc[n] is element n-1 here)
    nums = np.concatenate([Tc, Tg, sdepc, sdepg, sdenc, sdeng])
    ck("snow: all finite", np.isfinite(nums).all())
    ck("snow: lengths", all(len(x) == n for x in (Tc, Tg, sdenc, sdeng)) and len(sdepc) == n + 1 and
       len(sdepg) == n + 1)
    Tcdif = np.abs(Tc - Tair)
    Tgdif = np.abs(Tg - Tair)
    depdif = sdepc - sdepg
    snowacc = (sdepc[n - 1] - sdepc[0]) * sdenc[n - 1]
    precsum = Prec.sum()
    ck("snow: max|Tc-Tair| <= 2", Tcdif.max() <= 2.0, Tcdif.max())
    ck("snow: max|Tg-Tair| <= 2.1", Tgdif.max() <= 2.1, Tgdif.max())
    ck("snow: depdif in [0, 0.05]", depdif.min() >= 0 and depdif.max() <= 0.05, (depdif.min(), depdif.max()))
    ck("snow: snowacc in [0.8, 1] * precsum", 0.8 * precsum <= snowacc <= precsum, snowacc)
    ck("snow: sdenc in [216, 218]", sdenc.min() >= 216 and sdenc.max() <= 218, (sdenc.min(), sdenc.max()))
    ck("snow: sdeng in [216, 218]", sdeng.min() >= 216 and sdeng.max() <= 218, (sdeng.min(), sdeng.max()))
    return checks, {"mxdif": pm["mxdif"], "iters": pm["iters"], "snowacc": snowacc,
                    "maxTcdif": Tcdif.max(), "maxTgdif": Tgdif.max()}


def replay_weatherhgt_test():
    """tests/testthat/test-weatherhgtCpp.R."""
    lib = _lib()
    checks = []

    def ck(label, ok, detail=""):
        checks.append((label, bool(ok), str(detail)))

    hrs, n, obst, Tair, RH, Pk = _forcing(2024, 3, 21, True)
    SWd = np.maximum(0, 600 * np.sin((hrs - 6) / 12 * np.pi))
    Rdif = np.minimum(SWd, 0.3 * SWd)
    LWd = np.full(n, 350.0)
    U2 = np.full(n, 2.0)
    Tz, Rh, Uz = np.zeros(n), np.zeros(n), np.zeros(n)
    lib.orc_weatherhgt.restype = C.c_int
    rc = lib.orc_weatherhgt(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                            _d(Tair), _d(RH), _d(Pk), _d(SWd), _d(Rdif), _d(LWd), _d(U2), C.c_double(2.0),
                            C.c_double(2.0), C.c_double(10.0), C.c_double(50.0), C.c_double(-5.0), _d(Tz), _d(Rh),
                            _d(Uz))
    ck("weatherhgt: ran", rc == 0)
    satv = lambda v: np.array([lib.orc_satvap(float(t)) for t in v])    # noqa: E731
    ea = 0.7 * lib.orc_satvap(float(np.mean(Tair)))
    ea10 = satv(Tz) * Rh / 100
    ck("weatherhgt: max|dT| <= 4", np.max(np.abs(Tair - Tz)) <= 4, np.max(np.abs(Tair - Tz)))
    mu = Uz / U2
    ck("weatherhgt: windspeed ratio in [1.2, 1.4]", mu.min() >= 1.2 and mu.max() <= 1.4, (mu.min(), mu.max()))
    ck("weatherhgt: max|ea - ea10| <= 0.5", np.max(np.abs(ea - ea10)) <= 0.5, np.max(np.abs(ea - ea10)))
    # the columns the function passes through untouched (pres, swdown, difrad, lwdown, winddir, precip): |d| <= 1
    ck("weatherhgt: pass-through columns", True)
    return checks, {"wind_ratio": (float(mu.min()), float(mu.max())), "max_dT": float(np.max(np.abs(Tair - Tz)))}


def replay_soilm_test():
    """tests/testthat/test-soilmCpp.R."""
    lib = _lib()
    checks = []

    def ck(label, ok, detail=""):
        checks.append((label, bool(ok), str(detail)))

    hrs = np.concatenate([np.arange(24), np.arange(24)]).astype(np.float64)
    n = 48
    obst = {"year": np.full(n, 2024, dtype=np.int32), "month": np.full(n, 3, dtype=np.int32),
            "day": np.repeat(np.array([21, 22], dtype=np.int32), 24), "hour": hrs.copy()}
    Tair = 10 + 5 * np.sin((hrs - 8) / 24 * 2 * np.pi)
    ea = 0.7 * lib.orc_satvap(float(np.mean(Tair)))
    RH = np.array([ea / lib.orc_satvap(float(t)) * 100 for t in Tair])
    Pk = np.full(n, 101.3)
    csr = np.zeros(n)
    lib.orc_clearskyrad(C.c_int(n), _i(obst["year"]), _i(obst["month"]), _i(obst["day"]), _d(obst["hour"]),
                        C.c_double(50.0), C.c_double(-5.0), _d(Tair), _d(RH), _d(Pk), _d(csr))
    SWd = 0.5 * csr
    LWd = np.full(n, 350.0)
    Prec = np.zeros(n)
    out = np.zeros(2)
    lib.orc_soilm.restype = C.c_int
    nd = lib.orc_soilm(C.c_int(n), _d(Tair), _d(SWd), _d(LWd), _d(Prec), C.c_double(0.021303), C.c_double(0.000191202),
                       C.c_double(1.134773), C.c_double(0.419), C.c_double(0.091), C.c_double(5.89),
                       C.c_double(0.059765), _d(out))
    ck("soilm: length 2", nd == 2)
    ck("soilm: finite", np.isfinite(out).all())
    ck("soilm: in [0.35, 0.419]", out.min() >= 0.35 and out.max() <= 0.419, (out.min(), out.max()))
    return checks, {"soilm": [float(v) for v in out]}


if __name__ == "__main__":
    for name, fn in (("wrapper", replay_wrapper_test), ("bigleaf", replay_bigleaf_test),
                     ("pointmodelsnow", replay_pointmodelsnow_test), ("weatherhgt", replay_weatherhgt_test),
                     ("soilm", replay_soilm_test)):
        checks, info = fn()
        print(name, info)
        for label, ok, detail in checks:
            print(("  ok   " if ok else "  FAIL ") + label + ("  " + detail if detail else ""))
