"""Seeded synthetic workloads for tests and bench.py (SURVEY.md §8d recipe).

Counter-based: every value is splitmix64(seed ^ field ^ global_index) -> U[0,1),
so any row block of a large raster can be generated independently on any rank.
The forcing formulas follow the survey's recipe; `es/ea/tdew` use the R-side
definitions the reference's marshaller applies (R/internal.R:501-521).
"""
from __future__ import annotations

import numpy as np

SEED = 20240321
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(field: int, idx: np.ndarray, seed: int = SEED) -> np.ndarray:
    """U[0,1) keyed by (seed, field id, integer index array)."""
    key = np.uint64(seed) ^ (np.uint64(field) << np.uint64(40))
    h = _splitmix64(np.asarray(idx, dtype=np.uint64) ^ key)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(field: int, idx: np.ndarray, seed: int = SEED) -> np.ndarray:
    u1 = uniform(field, idx, seed)
    u2 = uniform(field + 1, idx, seed)
    return np.sqrt(-2.0 * np.log(np.maximum(u1, 1e-300))) * np.cos(2 * np.pi * u2)


# ---- R-side derived climate variables (R/internal.R:501-521) -----------------
from .rformulas import dewpoint_R, satvap_R  # noqa: E402,F401  (the R-side formulas, also used by frontend.py)


def solar_zenith_cos(lat, lon, year, month, day, hour):
    """cos(zenith) by the usual declination / equation-of-time formulas (own
    numpy code; only used to shape synthetic shortwave forcing)."""
    year = np.asarray(year, dtype=np.int64)
    month = np.asarray(month, dtype=np.int64)
    day = np.asarray(day, dtype=np.int64)
    madj = month + (month < 3) * 12
    yadj = year - (month < 3)
    j = np.trunc(365.25 * (yadj + 4716)) + np.trunc(30.6001 * (madj + 1)) + day + 0.5 - 1524.5
    b = 2 - yadj // 100 + (yadj // 100) // 4
    jd = np.trunc(j + (j > 2299160) * b)
    m = 6.24004077 + 0.01720197 * (jd - 2451545.0)
    eot = -7.659 * np.sin(m) + 9.863 * np.sin(2 * m + 3.5932)
    st = hour + (4.0 * lon + eot) / 60.0
    tt = 0.261799 * (st - 12)
    dec = (np.pi * 23.5 / 180) * np.cos(2 * np.pi * ((jd - 159.5) / 365.25))
    latr = np.deg2rad(lat)
    return np.sin(dec) * np.sin(latr) + np.cos(dec) * np.cos(latr) * np.cos(tt)


def calendar(tsteps: int, year: int = 2023, start_doy: int = 1):
    """Hourly calendar: tsteps consecutive hours starting at 00:00 of start_doy."""
    t0 = np.datetime64(f"{year}-01-01T00", "h") + np.timedelta64((start_doy - 1) * 24, "h")
    t = t0 + np.arange(tsteps).astype("timedelta64[h]")
    Y = t.astype("datetime64[Y]").astype(int) + 1970
    Mo = t.astype("datetime64[M]").astype(int) % 12 + 1
    D = (t.astype("datetime64[D]") - t.astype("datetime64[M]").astype("datetime64[D]")).astype(int) + 1
    H = (t - t.astype("datetime64[D]")).astype(int).astype(np.float64)
    doy = (t.astype("datetime64[D]") - t.astype("datetime64[Y]").astype("datetime64[D]")).astype(int) + 1
    obstime = {"year": Y.astype(np.int32), "month": Mo.astype(np.int32), "day": D.astype(np.int32),
               "hour": H}
    return obstime, doy.astype(np.float64)


def forcing_vectors(tsteps: int, lat=50.0, lon=-5.0, year=2023, start_doy=1, seed=SEED,
                    cold: float = 0.0):
    """climdata + pointm as length-T vectors (runmicro1Cpp geometry).
    `cold` shifts the temperature down (to exercise the sub-zero branches)."""
    obstime, doy = calendar(tsteps, year, start_doy)
    k = np.arange(tsteps, dtype=np.uint64) + np.uint64((start_doy - 1) * 24)
    kd = k // np.uint64(24)
    h = obstime["hour"]
    u = uniform(1, k, seed)
    uday = uniform(2, kd, seed)
    temp = 10 + 8 * np.sin(2 * np.pi * (doy - 110) / 365) + 5 * np.sin(2 * np.pi * (h - 9) / 24) \
        + 2 * (u - 0.5) - cold
    relhum = np.clip(80 - 2 * (temp - 10) + 10 * (uniform(3, k, seed) - 0.5), 20, 100)
    cz = solar_zenith_cos(lat, lon, obstime["year"], obstime["month"], obstime["day"], h)
    swdown = np.maximum(0.0, 0.75 * 1352 * cz) * (0.3 + 0.7 * uday)
    difrad = swdown * (0.2 + 0.6 * uniform(4, kd, seed))
    lwdown = 0.8 * 5.67e-8 * (temp + 273.15) ** 4
    wind = np.maximum(0.5, 3 * np.exp(0.5 * normal(5, k, seed)))
    wdir = 360 * uniform(7, k, seed)
    es = satvap_R(temp)
    ea = es * relhum / 100
    tdew = dewpoint_R(ea, temp)
    climdata = {"temp": temp, "es": es, "ea": ea, "tdew": tdew, "pres": np.full(tsteps, 101.3),
                "swdown": swdown, "difrad": difrad, "lwdown": lwdown, "windspeed": wind,
                "winddir": wdir}
    pointm = {
        "soilm": 0.15 + 0.25 * uniform(8, kd, seed),
        "Tg": temp + 0.05 * swdown,
        "T0p": temp.copy(),
        "Tbp": 10 + 6 * np.sin(2 * np.pi * (doy - 130) / 365),
        "G": 60 * np.sin(2 * np.pi * (h - 11) / 24),
        "DDp": 0.08 + 0.07 * uniform(9, kd, seed),
        "umu": 0.8 + 0.4 * uniform(10, k, seed),
        "kp": 0.8 + 0.8 * uniform(11, kd, seed),
        "muGp": 0.08 + 0.07 * uniform(9, kd, seed),
        "dtrp": 5 + 20 * uniform(12, kd, seed),
    }
    return obstime, climdata, pointm


LOAM = dict(Smin=0.074, Smax=0.42, soilb=5.2, Psie=-5.6, Vq=0.06, Vm=0.509, Mc=0.5422, rho=1.53)


def rasters(rows: int, cols: int, row0: int = 0, rows_total: int | None = None, seed=SEED,
            reqhgt: float = 0.05, hgt_range=(0.05, 1.5), bare_frac=0.05, na_frac=0.01,
            variety: bool = False):
    """vegp + soilc for a [rows, cols] block whose first row is global row `row0`
    of a raster with `rows_total` rows.  `variety=True` adds the edge cases of
    SURVEY Appendix D (x==1, clump==0, flat cells, tall horizon, wet/dry soils)."""
    rows_total = rows if rows_total is None else rows_total
    i = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    idx = i + np.uint64(rows_total) * j          # global column-major cell index
    fi, fj = i.astype(np.float64), j.astype(np.float64)

    def U(f, lo=0.0, hi=1.0):
        return lo + (hi - lo) * uniform(f, idx, seed)

    hgt = U(20, *hgt_range)
    pai = U(21, 0.2, 4.0)
    x = U(22, 0.5, 2.0)
    clump = U(26, 0.0, 0.4)
    slope = U(40, 0.0, 35.0)
    aspect = U(41, 0.0, 360.0)
    if variety:
        sel = U(60)
        x = np.where(sel < 0.08, 1.0, x)
        clump = np.where((sel > 0.08) & (sel < 0.2), 0.0, clump)
        slope = np.where((sel > 0.2) & (sel < 0.3), 0.0, slope)
    bare = U(30) < bare_frac
    hgt = np.where(bare, 0.0, hgt)
    pai = np.where(bare, 0.0, pai)
    na = U(31) < na_frac
    hgt = np.where(na, np.nan, hgt)
    # foliage above reqhgt: the survey's simple 0.7*pai / pai/hgt stand-ins
    paia = np.where(reqhgt < hgt, 0.7 * pai, 0.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        leafden = pai / hgt
    vegp = {"hgt": hgt, "pai": pai, "x": x, "gsmax": U(23, 0.2, 0.4), "leafr": U(24, 0.3, 0.45),
            "leaft": U(25, 0.1, 0.25), "clump": clump, "leafd": U(27, 0.01, 0.1), "paia": paia,
            "leafden": leafden}
    soilc = {k: np.full((rows, cols), v) for k, v in LOAM.items()}
    if variety:
        soilc["Smax"] = U(61, 0.36, 0.48)
        soilc["Smin"] = U(62, 0.05, 0.10)
    soilc["gref"] = U(42, 0.1, 0.2)
    soilc["slope"] = slope
    soilc["aspect"] = aspect
    soilc["twi"] = U(43, 1.0, 50.0)
    soilc["svfa"] = U(44, 0.7, 1.0)
    d8 = np.arange(8, dtype=np.uint64)[None, None, :]
    d24 = np.arange(24, dtype=np.uint64)[None, None, :]
    n_tot = np.uint64(rows_total) * np.uint64(cols)
    soilc["wsa"] = 0.5 + 0.5 * uniform(45, idx[:, :, None] + n_tot * d8, seed)
    hor = 0.3 * uniform(46, idx[:, :, None] + n_tot * d24, seed)
    if variety:
        hor = np.where(uniform(63, idx[:, :, None] + n_tot * d24, seed) < 0.05, hor + 0.6, hor)
    soilc["hor"] = hor
    # synthetic DTM of the survey recipe (not read by the solver; kept for the terrain kernels)
    dtm = 100 + 40 * np.sin(2 * np.pi * fi / 257) * np.cos(2 * np.pi * fj / 193) \
        + 12 * np.sin(2 * np.pi * (fi + fj) / 61) + U(47)
    return vegp, soilc, dtm


def workload(rows: int, cols: int, tsteps: int, reqhgt: float = 0.05, zref: float = 2.0,
             row0: int = 0, rows_total: int | None = None, seed=SEED, start_doy: int = 1,
             array_forcing: bool = False, variety: bool = False, cold: float = 0.0,
             hgt_range=(0.05, 1.5), lat: float = 50.0, lon: float = -5.0, out=None,
             complete: bool = True, year: int = 2023, na_frac: float = 0.01):
    """Positional-argument dict for runmicro1Cpp / runmicro2Cpp."""
    obstime, climdata, pointm = forcing_vectors(tsteps, lat, lon, year, start_doy, seed, cold)
    vegp, soilc, _ = rasters(rows, cols, row0, rows_total, seed, reqhgt, hgt_range, na_frac=na_frac, variety=variety)
    # column-major like the R matrices / arrays the reference receives (marshal() then takes them without a copy)
    vegp = {k: np.asfortranarray(v) for k, v in vegp.items()}
    soilc = {k: np.asfortranarray(v) for k, v in soilc.items()}
    args = dict(obstime=obstime, climdata=climdata, pointm=pointm, vegp=vegp, soilc=soilc,
                reqhgt=reqhgt, zref=zref, lat=lat, lon=lon, Sminp=0.074, Smaxp=0.42, tfact=1.5,
                complete=complete, mat=10.0, out=[True] * 10 if out is None else list(out))
    if array_forcing:
        rows_total = rows if rows_total is None else rows_total
        i = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        j = np.arange(cols, dtype=np.uint64)[None, :]
        idx = (i + np.uint64(rows_total) * j)[:, :, None]
        kk = (np.arange(tsteps, dtype=np.uint64) * np.uint64(rows_total * cols))[None, None, :]
        pert = uniform(70, idx + kk, seed) - 0.5          # per cell-step perturbation

        def arr(v, scale):
            return np.asfortranarray(v[None, None, :] + scale * pert)

        temp = arr(climdata["temp"], 1.0)
        es = satvap_R(temp)
        rh = np.clip(100 * climdata["ea"][None, None, :] / es, 20, 100)
        ea = es * rh / 100
        clim2 = {"tc": temp, "es": es, "ea": ea, "tdew": dewpoint_R(ea, temp),
                 "pk": arr(climdata["pres"], 0.5),
                 "swdown": np.asfortranarray(climdata["swdown"][None, None, :] * (1 + 0.1 * pert)),
                 "difrad": np.asfortranarray(climdata["difrad"][None, None, :] * (1 + 0.1 * pert)),
                 "lwdown": arr(climdata["lwdown"], 5.0),
                 "windspeed": np.asfortranarray(climdata["windspeed"][None, None, :] * (1 + 0.2 * pert)),
                 "winddir": climdata["winddir"]}
        pm2 = {}
        for k2, v in pointm.items():
            key = "Gp" if k2 == "G" else k2
            pm2[key] = np.asfortranarray(v[None, None, :] * (1 + 0.05 * pert))
        fi = i.astype(np.float64)
        fj = j.astype(np.float64)
        lats = lat + 0.01 * (fi / max(rows_total, 1)) + 0 * fj
        lons = lon + 0.01 * (fj / max(cols, 1)) + 0 * fi
        args.update(climdata=clim2, pointm=pm2, lat=lats, lon=lons)
    return args


def layered(args: dict, layers: int, cover_days: int | None = None, seed=SEED):
    """Turns a static-vegetation workload into a runmicro3Cpp/4Cpp one: `layers` vegetation layers
    (pai / hgt / x scaled per layer) and the `dfsel` table that deals whole days to them
    (R/internal.R:1391-1399).  `cover_days` < total days leaves the tail uncovered."""
    a = dict(args)
    T = len(a["obstime"]["year"])
    ndays = T // 24 if cover_days is None else cover_days
    rows, cols = np.shape(a["vegp"]["hgt"])
    veg = {}
    for k, v in a["vegp"].items():
        stack = []
        for l in range(layers):
            f = 1.0
            if k == "pai" or k == "paia":
                f = 0.6 + 0.8 * l / max(layers - 1, 1)
            elif k == "hgt":
                f = 0.9 + 0.2 * l / max(layers - 1, 1)
            stack.append(v * f)
        veg[k] = np.stack(stack, axis=2)
    with np.errstate(invalid="ignore", divide="ignore"):
        veg["leafden"] = veg["pai"] / veg["hgt"]
    veg["paia"] = np.where(a["reqhgt"] < veg["hgt"], veg["paia"], 0.0)
    a["vegp"] = veg
    edges = np.linspace(0, ndays, layers + 1).round().astype(int)
    a["dfsel"] = {"lyr": np.arange(1, layers + 1), "st": edges[:-1] * 24, "ed": edges[1:] * 24 - 1}
    return a


def snow_workload(rows: int, cols: int, tsteps: int, seed=SEED, array_forcing: bool = False, start_doy: int = 15,
                  cold: float = 11.0, lat: float = 57.0, lon: float = -4.0, zref: float = 2.0,
                  hgt_range=(0.05, 3.0), na_frac: float = 0.02, bare_frac: float = 0.1, snowenv: str = "Alpine",
                  year: int = 2023, row0: int = 0, rows_total: int | None = None):
    """Seeded inputs of the snow branch: the argument lists of gridmodelsnow1/2
    (src/microclimfCpp.cpp:4172, 4426).  Winter calendar, temperatures straddling 0 and 2 degC so
    that snowfall, rain-on-snow, melt and snow-free steps all occur; cells start with 0 - 0.6 m of
    snow (a fifth of them bare) and vegetation from below to well above the pack."""
    obstime, clim, pm = forcing_vectors(tsteps, lat, lon, year, start_doy, seed, cold)
    vegp0, soilc, _ = rasters(rows, cols, row0, rows_total, seed, 0.05, hgt_range, bare_frac=bare_frac, na_frac=na_frac,
                              variety=True)
    k = np.arange(tsteps, dtype=np.uint64) + np.uint64((start_doy - 1) * 24)
    h = obstime["hour"]
    relhum = 100 * clim["ea"] / clim["es"]
    wet = uniform(80, k // np.uint64(6), seed) < 0.3          # 6-hour wet spells
    precip = np.where(wet, 0.2 + 2.5 * uniform(81, k, seed), 0.0)
    climdata = {"temp": clim["temp"], "relhum": relhum, "pres": clim["pres"], "swdown": clim["swdown"],
                "difrad": clim["difrad"], "lwdown": clim["lwdown"], "windspeed": clim["windspeed"],
                "winddir": clim["winddir"], "precip": precip, "umu": pm["umu"]}
    pointm = {"Gp": 15 * np.sin(2 * np.pi * (h - 11) / 24) + 4 * (uniform(82, k, seed) - 0.5),
              "Tc": np.minimum(clim["temp"] - 0.8 + 0.004 * clim["swdown"], 0.5),
              "RswabsG": 0.2 * clim["swdown"], "RlwabsG": 0.9 * 0.97 * clim["lwdown"], "umu": pm["umu"],
              "tr": np.full(tsteps, 0.5)}
    i = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    idx = i + np.uint64(rows if rows_total is None else rows_total) * j          # global cell index: blocks of one raster agree

    def U(f, lo=0.0, hi=1.0):
        return lo + (hi - lo) * uniform(f, idx, seed)

    vegp = {kk: vegp0[kk] for kk in ("pai", "hgt", "leaft", "clump", "paia", "leafd", "leafden")}
    dc = np.where(U(90) < 0.2, 0.0, U(91, 0.0, 0.6))
    other = {"slope": soilc["slope"], "aspect": soilc["aspect"], "skyview": soilc["svfa"], "wsa": soilc["wsa"],
             "hor": soilc["hor"], "lat": lat, "lon": lon, "zref": zref, "isnowdc": dc, "isnowdg": dc * U(92, 0.3, 1.0),
             "isnowac": np.floor(U(93, 0, 200)), "isnowag": np.floor(U(94, 0, 300)), "Smax": soilc["Smax"]}
    if array_forcing:
        kk = (np.arange(tsteps, dtype=np.uint64) * np.uint64(rows * cols))[None, None, :]
        pert = uniform(70, idx[:, :, None] + kk, seed) - 0.5

        def add(v, s):
            return np.asfortranarray(v[None, None, :] + s * pert)

        def mul(v, s):
            return np.asfortranarray(v[None, None, :] * (1 + s * pert))

        climdata = {"temp": add(climdata["temp"], 1.0), "relhum": np.clip(add(relhum, 6.0), 20, 100),
                    "pres": add(climdata["pres"], 0.5), "swdown": mul(climdata["swdown"], 0.1),
                    "difrad": mul(climdata["difrad"], 0.1), "lwdown": add(climdata["lwdown"], 5.0),
                    "windspeed": mul(climdata["windspeed"], 0.2), "winddir": climdata["winddir"],
                    "precip": np.asfortranarray(np.maximum(precip[None, None, :] * (1 + 1.5 * pert) - 0.15, 0.0)),
                    "umu": mul(pm["umu"], 0.05)}
        pointm = {kk2: (mul(v, 0.05) if kk2 != "Tc" else add(v, 0.3)) for kk2, v in pointm.items()}
        fi, fj = i.astype(np.float64), j.astype(np.float64)
        other["lats"] = lat + 0.01 * (fi / max(rows, 1)) + 0 * fj
        other["lons"] = lon + 0.01 * (fj / max(cols, 1)) + 0 * fi
    return dict(obstime=obstime, climdata=climdata, pointm=pointm, vegp=vegp, other=other, snowenv=snowenv)


def microsnow_inputs(sw: dict, smod: dict, seed=SEED):
    """Arguments of gridmicrosnow1/2 from a snow_workload and a gridmodelsnow result: `snowm` as
    `.snowmodel1` assembles it (R/internal.R:2611-2615: totalSWE = depth * density) and a `micro`
    list of recognisable stand-in fields for the no-snow solver's output."""
    R, Cc, T = smod["Tc"].shape
    with np.errstate(invalid="ignore"):      # NA_real_ is a signalling NaN
        swe = smod["sdepc"] * smod["sden"]
    snowm = {"Tc": smod["Tc"], "Tg": smod["Tg"], "totalSWE": swe,
             "groundsnowdepth": smod["sdepg"], "snowden": smod["sden"]}
    n = R * Cc * T
    base = np.arange(n, dtype=np.uint64)
    names = ("Tz", "tleaf", "relhum", "soilm", "windspeed", "Rdirdown", "Rdifdown", "Rlwdown", "Rswup", "Rlwup")
    micro = {nm: np.asfortranarray((1000.0 * (v + 1) + uniform(100 + v, base, seed)).reshape((R, Cc, T), order="F"))
             for v, nm in enumerate(names)}
    return snowm, micro


def coarse_workload(rows: int, cols: int, tsteps: int, crows: int, ccols: int, reqhgt: float = 0.05, seed=SEED, **kw):
    """Inputs of the coarse array-forcing mode (include/mcf.h, array_forcing == 2): fine rasters as `workload`, climate
    and point-model arrays on a [crows, ccols] grid covering the same extent (what `.runmodel2Cpp` holds before it
    resamples).  Returns (args for runmicro2Cpp_coarse, rowpos, colpos)."""
    fine = workload(rows, cols, tsteps, reqhgt=reqhgt, seed=seed, array_forcing=False, **kw)
    kw2 = {k: v for k, v in kw.items() if k in ("start_doy", "cold", "lat", "lon", "year")}
    co = workload(crows, ccols, tsteps, reqhgt=reqhgt, seed=seed, array_forcing=True, **kw2)
    c = co["climdata"]
    i = np.arange(crows * ccols, dtype=np.uint64).reshape((crows, ccols), order="F")[:, :, None]
    k = (np.arange(tsteps, dtype=np.uint64) * np.uint64(crows * ccols))[None, None, :]
    wd = (c["winddir"][None, None, :] + 40.0 * (uniform(71, i + k, seed) - 0.5)) % 360.0
    clim = {"temp": c["tc"], "relhum": np.asfortranarray(np.clip(100 * c["ea"] / c["es"], 5, 100)), "pres": c["pk"],
            "swdown": c["swdown"], "difrad": c["difrad"], "lwdown": c["lwdown"], "windspeed": c["windspeed"],
            "winddir": np.asfortranarray(wd)}
    fi = (np.arange(rows, dtype=np.float64))[:, None]
    fj = (np.arange(cols, dtype=np.float64))[None, :]
    args = dict(fine)
    args.update(climdata=clim, pointm=co["pointm"],
                lat=fine["lat"] + 0.01 * (fi / max(rows, 1)) + 0 * fj, lon=fine["lon"] + 0.01 * (fj / max(cols, 1)) + 0 * fi)
    pos = lambda n, m: np.clip((np.arange(n) + 0.5) * (m / n) - 0.5, 0.0, m - 1.0)       # noqa: E731
    return args, pos(rows, crows), pos(cols, ccols)
