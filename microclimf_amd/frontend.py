"""Host-side mirror of the reference's user-level calls for data.frame weather — `runpointmodel()`
(R/Cppwrappers.R:59-139) and `runmicro()` → `.runmicronosnow` → `.runmodel1Cpp` / `.runmodel3Cpp`
(R/Cppwrappers.R:376-397, R/internal.R:3290-3349, 1065-1169, 1345-1459) — for hosts without R.

In an R session these R functions keep running unchanged above the C ABI.  Here every step they take
between the user's rasters and the solver call is restated with numpy on top of libmcfhip's own producers:

    step of the reference                          here
    ---------------------------------------------  -----------------------------------------------------
    soilmCpp, BigLeafCpp, pointmprocess, manCpp     libmcfhip, host C++ (pointmodel.py)
    terra::terrain, .horizon, .windsheltera         mcf_precompute_terrain (device kernels, terrain.py)
    .topidx / flowaccCpp                            mcf_topidx (host C++)
    runmicro1Cpp / runmicro3Cpp                     mcf_runmicro1 / mcf_runmicro3 (the hot path)
    .sortvegp, .soilinit, .foliageden, .satvap ...  numpy, below, citing the R lines they follow

Rasters are numpy arrays `[rows, cols]` (or `[rows, cols, layers]`), row 0 = northern edge as in terra; what a
SpatRaster carries besides values is passed as `dtm = {"z": array, "res": xres | (xres, yres), "lat": ., "long": .}`
(the reference gets lat / long from the CRS through sf/PROJ, `.latlongfromraster`, R/internal.R:61-69).
`checkinputs()` (R/dataprep.R) is not mirrored: inputs are taken as checked (`runchecks = FALSE`).
"""
from __future__ import annotations

from typing import Mapping, Sequence

import numpy as np

from . import api, pointmodel, terrain
from .soil_tables import SOILPARAMETERS, SOILPARAMSP
from .rformulas import dewpoint_R as _dewpoint, satvap_R as _satvap   # .satvap / .dewpoint, R/internal.R:501-521

WEATHER = ("temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir", "precip")


# ---- small R helpers --------------------------------------------------------------------------------------------
def getmode(v):
    """`.getmode` (R/internal.R:107-111): most frequent non-NA value, the first seen on ties."""
    v = np.asarray(v, dtype=np.float64).ravel(order="F")
    v = v[~np.isnan(v)]
    if v.size == 0:
        return np.nan
    u, first, counts = np.unique(v, return_index=True, return_counts=True)
    order = np.argsort(first)                      # unique() in R keeps first-appearance order
    return float(u[order][np.argmax(counts[order])])


def r_round(x):
    """R's round(x, 0): half to even"""
    return np.rint(x)


def layer_index(nr: int, n: int):
    """`s <- round(seq(0.50001, nr + 0.5, length.out = n), 0)` clipped to 1..nr (R/internal.R:188-191); 1-based"""
    s = r_round(np.linspace(0.50001, nr + 0.5, n)) if n > 1 else r_round(np.array([0.50001]))
    return np.clip(s, 1, nr).astype(int)


def as3d(a):
    a = np.asarray(a, dtype=np.float64)
    return a[:, :, None] if a.ndim == 2 else a


def intr(a, n: int, subs):
    """`.intr` (R/internal.R:187-195): pick, for each of `n` time steps, the nearest of the raster's layers"""
    a = as3d(a)
    s = layer_index(a.shape[2], n)[np.asarray(subs) - 1]
    return a[:, :, s - 1]


def spline_fmm(y, n: int):
    """`stats::spline(y, n = n)$y` for x = 1..length(y): the cubic spline of Forsythe, Malcolm & Moler (1977), whose
    end conditions fit cubics through the first and the last four points, evaluated at `n` equally spaced abscissae
    (R's default `method = "fmm"`, `xmin = min(x)`, `xmax = max(x)`)."""
    y = np.asarray(y, dtype=np.float64)
    m = len(y)
    x = np.arange(1.0, m + 1.0)
    xo = np.linspace(1.0, float(m), n)
    if m < 2:
        return np.full(n, y[0] if m else np.nan)
    b, c, d = np.zeros(m), np.zeros(m), np.zeros(m)
    if m < 3:
        b[:] = (y[1] - y[0]) / (x[1] - x[0])
    else:
        d[0] = x[1] - x[0]
        c[1] = (y[1] - y[0]) / d[0]
        for i in range(1, m - 1):
            d[i] = x[i + 1] - x[i]
            b[i] = 2.0 * (d[i - 1] + d[i])
            c[i + 1] = (y[i + 1] - y[i]) / d[i]
            c[i] = c[i + 1] - c[i]
        b[0], b[m - 1] = -d[0], -d[m - 2]
        c[0] = c[m - 1] = 0.0
        if m > 3:
            c[0] = c[2] / (x[3] - x[1]) - c[1] / (x[2] - x[0])
            c[m - 1] = c[m - 2] / (x[m - 1] - x[m - 3]) - c[m - 3] / (x[m - 2] - x[m - 4])
            c[0] = c[0] * d[0] * d[0] / (x[3] - x[0])
            c[m - 1] = -c[m - 1] * d[m - 2] * d[m - 2] / (x[m - 1] - x[m - 4])
        for i in range(1, m):                       # Gaussian elimination
            t = d[i - 1] / b[i - 1]
            b[i] = b[i] - t * d[i - 1]
            c[i] = c[i] - t * c[i - 1]
        c[m - 1] = c[m - 1] / b[m - 1]              # back substitution
        for i in range(m - 2, -1, -1):
            c[i] = (c[i] - d[i] * c[i + 1]) / b[i]
        b[m - 1] = (y[m - 1] - y[m - 2]) / d[m - 2] + d[m - 2] * (c[m - 2] + 2.0 * c[m - 1])
        for i in range(m - 1):
            b[i] = (y[i + 1] - y[i]) / d[i] - d[i] * (c[i + 1] + 2.0 * c[i])
            d[i] = (c[i + 1] - c[i]) / d[i]
            c[i] = 3.0 * c[i]
        c[m - 1] = 3.0 * c[m - 1]
        d[m - 1] = d[m - 2]
    i = np.clip(np.searchsorted(x, xo, side="right") - 1, 0, m - 1)
    dx = xo - x[i]
    return y[i] + dx * (b[i] + dx * (c[i] + dx * d[i]))


# ---- vegetation -------------------------------------------------------------------------------------------------
VEG_KEYS = ("hgt", "pai", "x", "gsmax", "leafr", "clump", "leafd", "leaft")


def vegpdmx(vegp) -> int:
    """`.vegpdmx` (R/internal.R:197-208)"""
    return max(as3d(vegp[k]).shape[2] for k in VEG_KEYS)


def sortvegp_point(vegp) -> np.ndarray:
    """`.sortvegp(vegp, method = "P")` (R/internal.R:229-239): the point model's vegetation vector"""
    m = {k: float(np.nanmean(np.asarray(vegp[k], dtype=np.float64))) for k in VEG_KEYS}
    return np.array([m["hgt"], m["pai"], m["x"], m["clump"], m["leafr"], m["leaft"], m["leafd"], 0.97, m["gsmax"], 100.0])


def sortvegp_grid(vegp, n: int, subs):
    """`.sortvegp(vegp, method = "C", n, subs)` (R/internal.R:249-273): every variable expanded to the layers of the
    variable with most layers, and `lsubs`, the layer of each time step with whole days never split"""
    subs = np.asarray(subs)
    dmx = vegpdmx(vegp)
    s = layer_index(dmx, n)
    first = []
    for v in s[subs - 1]:                              # unique(s[subs]) keeps first-appearance order
        if v not in first:
            first.append(int(v))
    out = {k: intr(vegp[k], dmx, np.array(first)) for k in VEG_KEYS}
    ss = s[subs - 1]
    if len(ss) % 24:
        raise ValueError("the grid model needs whole days (matrix(s, ncol = 24) in .sortvegp)")
    sdd = np.array([getmode(row) for row in ss.reshape(-1, 24)]).astype(int)
    out["lsubs"] = np.repeat(sdd, 24)
    return out


def foliageden(z, hgt, pai, paia=None, shape: float = 1.5, rate: float | None = None):
    """`.foliageden` (R/internal.R:937-946): gamma-shaped foliage profile -> leaf density at z and plant area above z"""
    from scipy.stats import gamma
    rate = shape / 7 if rate is None else rate
    g = gamma(a=shape, scale=1.0 / rate)
    with np.errstate(invalid="ignore", divide="ignore"):
        x = ((hgt - z) / hgt) * 10
        td = g.cdf(10)
        rfd = g.pdf(x) / td
        tdf = (pai / hgt) * rfd * 10
        if paia is None:
            paia = g.cdf(x) * (pai / td)
    return tdf, paia


def cleanvars(vegp, soilc, dtm_z):
    """`.cleanvars` (R/internal.R:1020-1063): one NA mask for everything; zero pai where pai or hgt is zero or a
    vegetation parameter is missing.  (`.cleanr(vegp$hgt, 0)` there sets cell 0 — nothing — so hgt keeps its values.)"""
    veg = {k: as3d(vegp[k]).copy() for k in VEG_KEYS}
    soil = {k: np.asarray(v, dtype=np.float64).copy() for k, v in soilc.items()}
    z = np.asarray(dtm_z, dtype=np.float64).copy()
    na = (np.isnan(veg["pai"][:, :, 0]) | np.isnan(veg["hgt"][:, :, 0]) | np.isnan(soil["soiltype"])
          | np.isnan(soil["groundr"]) | np.isnan(z))
    for k in VEG_KEYS:
        veg[k][na] = np.nan
    for k in ("soiltype", "groundr"):
        soil[k][na] = np.nan
    z[na] = np.nan
    with np.errstate(invalid="ignore"):
        zero = (veg["pai"][:, :, 0] == 0) | (veg["hgt"][:, :, 0] == 0)
    for k in ("gsmax", "leafr", "clump", "leafd", "leaft"):
        zero |= np.isnan(veg[k][:, :, 0])
    veg["pai"][zero] = 0.0
    veg["pai"][na] = np.nan                              # mask(vegp$pai, dtm)
    veg["hgt"][na] = np.nan
    return veg, soil, z


# ---- soil -------------------------------------------------------------------------------------------------------
def soilinit(soilc) -> dict:
    """`.soilinit` (R/internal.R:304-336): per-cell soil constants from the soil type unless given explicitly"""
    st = np.asarray(soilc["soiltype"], dtype=np.float64)
    num = np.array(SOILPARAMETERS["Number"])
    out = {}
    for name, col in (("rho", "rho"), ("Vm", "Vm"), ("Vq", "Vq"), ("Mc", "Mc"), ("psi_e", "psi_e"), ("soilb", "b"),
                      ("Smax", "Smax"), ("Smin", "Smin")):
        if col in soilc:
            out[name] = np.asarray(soilc[col], dtype=np.float64)
            continue
        a = np.full(st.shape, np.nan)
        tab = np.array(SOILPARAMETERS[col])
        for u in np.unique(st[~np.isnan(st)]):
            a[st == u] = tab[num == u][0]
        out[name] = a
    return out


def sortsoilc_point(soilc) -> np.ndarray:
    """`.sortsoilc(soilc, "P")` (R/internal.R:338-357): the point model's ground vector (BigLeafCpp reads 12 entries)"""
    sl = soilinit(soilc)
    sn = int(getmode(soilc["soiltype"]))
    return np.array([getmode(soilc["groundr"]), 0.0, 180.0, 0.97, getmode(sl["rho"]), getmode(sl["Vm"]), getmode(sl["Vq"]),
                     getmode(sl["Mc"]), getmode(sl["soilb"]), getmode(sl["psi_e"]), getmode(sl["Smax"]), getmode(sl["Smin"]),
                     SOILPARAMSP["alpha"][sn - 1], SOILPARAMSP["n"][sn - 1], SOILPARAMSP["Ksat"][sn - 1]])


# ---- runpointmodel ----------------------------------------------------------------------------------------------
def soilbelowT(dfo: Mapping, reqhgt: float) -> np.ndarray:
    """`.soilbelowT` (R/internal.R:169-185)"""
    n = -118.35 * reqhgt / dfo["DDp"]
    nmn, nmx = int(np.floor(n.min())), int(np.ceil(n.max()))
    # manCpp's circular mean reads outside its array when the window exceeds the series (the reference then returns
    # whatever lies there); windows are capped at the whole days available
    cap = max(24 * (len(n) // 24), 1) if len(n) >= 48 else len(n)
    nmn, nmx = max(1, min(nmn, cap)), max(1, min(nmx, cap))
    Tnmn, Tnmx = pointmodel.manCpp(dfo["Tg"], nmn), pointmodel.manCpp(dfo["Tg"], nmx)
    if nmx == nmn:                                  # capped windows: one mean serves both
        Tb = Tnmn
    else:
        wgt = np.clip((n - nmn) / (nmx - nmn), 0.0, 1.0)
        Tb = wgt * Tnmx + (1 - wgt) * Tnmn
    wgt2 = 0.041596 * (reqhgt / np.mean(dfo["DDp"])) + 0.87142
    return wgt2 * Tb + (1 - wgt2) * np.mean(dfo["Tg"])


def runpointmodel(weather: Mapping, reqhgt: float, dtm: Mapping, vegp: Mapping, soilc: Mapping, *, zref: float = 2.0,
                  windhgt: float | None = None, soilm=None, matemp: float | None = None, dTmx: float = 25.0,
                  maxiter: int = 20, yearG: bool = True, lat: float | None = None, long: float | None = None,
                  vegp_p=None, groundp_p=None, soiltype: int | None = None, mxhgt: float | None = None) -> dict:
    """`runpointmodel(weather, reqhgt, dtm, vegp, soilc, ...)` (R/Cppwrappers.R:59-139).  `weather`: the columns of the
    reference's `climdata` plus `obstime = {year, month, day, hour}` in place of the POSIX `obs_time`."""
    w = {k: np.array(weather[k], dtype=np.float64, copy=True) for k in WEATHER if k in weather}
    obstime = {k: np.asarray(weather["obstime"][k]) for k in ("year", "month", "day", "hour")}
    n = len(w["temp"])
    windhgt = zref if windhgt is None else windhgt
    if matemp is None:
        matemp = float(np.mean(w["temp"]))
    if zref != windhgt:
        w["windspeed"] = w["windspeed"] * np.log(67.8 * zref - 5.42) / np.log(67.8 * windhgt - 5.42)
    if n < 8760:
        yearG = False
    vegp_p = sortvegp_point(vegp) if vegp_p is None else np.asarray(vegp_p, dtype=np.float64)
    groundp_p = sortsoilc_point(soilc) if groundp_p is None else np.asarray(groundp_p, dtype=np.float64)
    if mxhgt is None:
        mxhgt = float(np.nanmax(np.asarray(vegp["hgt"], dtype=np.float64)))
    zout = mxhgt if mxhgt > 2 else 2.0
    lat = dtm["lat"] if lat is None else lat
    long = dtm["long"] if long is None else long
    if zout > zref:
        w2 = pointmodel.weatherhgtCpp(obstime, w, zref, zout, zout, lat, long)
        if not np.isnan(np.mean(w2["temp"])):
            w = w2
        zref = zout
    w["windspeed"] = np.maximum(w["windspeed"], 0.5)
    if soilm is None:
        ii = (int(getmode(soilc["soiltype"])) if soiltype is None else int(soiltype)) - 1
        p = SOILPARAMSP
        sd = pointmodel.soilmCpp(w, p["rmu"][ii], p["mult"][ii], p["pwr"][ii], p["Smax"][ii], p["Smin"][ii], p["Ksat"][ii],
                                 p["a"][ii])
        soilm = spline_fmm(sd, n)
    soilm = np.asarray(soilm, dtype=np.float64)
    bl = pointmodel.BigLeafCpp(obstime, w, vegp_p, groundp_p, soilm, lat, long, dTmx, zref, maxiter, 0.5, 0.5, 0.1, yearG)
    pp = pointmodel.pointmprocess({"windspeed": w["windspeed"], "tc": w["temp"], "rh": w["relhum"], "pk": w["pres"],
                                   "uf": bl["uf"], "soilm": soilm, "RabsG": bl["RabsG"]},
                                  zref, vegp_p[0], vegp_p[1], groundp_p[4], groundp_p[5], groundp_p[6], groundp_p[7])
    dfo = dict(pp)
    dfo.update(G=bl["G"], soilm=soilm, Tg=bl["Tg"], Tc=bl["Tc"])
    Tbz = soilbelowT(dfo, reqhgt) if reqhgt < 0 else None
    return {"weather": w, "obstime": obstime, "dfo": dfo, "Tbz": Tbz, "lat": lat, "long": long, "zref": zref,
            "subs": np.arange(1, n + 1), "ntme": n, "matemp": matemp, "bigleaf_err": bl["err"]}


# ---- runmicro ---------------------------------------------------------------------------------------------------
def prepare_grid_inputs(micropoint: Mapping, reqhgt: float, vegp: Mapping, soilc: Mapping, dtm: Mapping, *, pai_a=None,
                        out: Sequence = (1,) * 10, slr=None, apr=None, hor=None, twi=None, wsa=None, svf=None,
                        device: int = 0) -> dict:
    """Everything `.runmodel1Cpp` / `.runmodel3Cpp` do before calling the solver (R/internal.R:1067-1166 / 1347-1456):
    returns the keyword arguments of runmicro1Cpp, plus `dfsel` when the vegetation varies in time."""
    res = dtm["res"]
    xres, yres = (res, res) if np.isscalar(res) else res
    if xres != yres:
        raise ValueError("the horizon / wind-shelter pre-compute works on z / res with one resolution, as the reference's "
                         ".horizon does (R/internal.R:909-925): square cells only")
    veg, soil, z = cleanvars(vegp, soilc, dtm["z"])
    w = micropoint["weather"]
    clim = {k: np.asarray(w[k], dtype=np.float64) for k in ("temp", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir")}
    clim["es"] = _satvap(clim["temp"])
    clim["ea"] = clim["es"] * np.asarray(w["relhum"], dtype=np.float64) / 100
    clim["tdew"] = _dewpoint(clim["ea"], clim["temp"])
    dfo = micropoint["dfo"]
    nt = len(clim["temp"])
    pointm = {k: np.asarray(dfo[k], dtype=np.float64) for k in ("soilm", "Tg", "T0p", "G", "DDp", "umu", "kp", "muGp", "dtrp")}
    pointm["Tbp"] = np.asarray(micropoint["Tbz"], dtype=np.float64) if reqhgt < 0 else np.zeros(nt)
    n, subs = micropoint["ntme"], micropoint["subs"]
    sv = sortvegp_grid(veg, n, subs)
    lsubs = sv.pop("lsubs")
    with np.errstate(invalid="ignore"):
        sv["hgt"][sv["pai"] == 0] = 0.0
        sv["pai"][sv["hgt"] == 0] = 0.0
    if pai_a is not None:
        pai_a = intr(pai_a, n, subs)
    sv["leafden"], sv["paia"] = foliageden(reqhgt, sv["hgt"], sv["pai"], pai_a)
    layers = sv["pai"].shape[2]
    dfsel = None
    if layers > 1:                                                       # R/internal.R:1391-1399
        lyr = []
        for v in lsubs:
            if v not in lyr:
                lyr.append(int(v))
        st, ed = [], []
        for v in lyr:
            s = np.nonzero(lsubs == v)[0] + 1                            # 1-based positions
            st.append(int(np.floor(s[0] / 24) * 24))
            ed.append(int(np.floor(s[-1] / 24) * 24 - 1))
        dfsel = {"lyr": np.arange(1, len(lyr) + 1), "st": np.array(st), "ed": np.array(ed)}
    else:
        sv = {k: v[:, :, 0] for k, v in sv.items()}
    sl = soilinit(soil)
    sc = {"gref": soil["groundr"], "Smin": sl["Smin"], "Smax": sl["Smax"], "soilb": sl["soilb"], "Psie": sl["psi_e"],
          "Vq": sl["Vq"], "Vm": sl["Vm"], "Mc": sl["Mc"], "rho": sl["rho"]}
    na = np.isnan(z)
    need = [k for k, v in (("slope", slr), ("aspect", apr), ("hor", hor), ("svfa", svf), ("wsa", wsa)) if v is None]
    ter = terrain.precompute_terrain(z, xres, micropoint["zref"], what=tuple(need), device=device) if need else {}
    for k, given in (("slope", slr), ("aspect", apr)):
        a = np.array(ter[k] if given is None else given, dtype=np.float64, copy=True)
        a[np.isnan(a)] = 0.0
        a[na] = np.nan
        sc[k] = a
    t = np.array(terrain.topidx(z, (xres, yres)) if twi is None else twi, dtype=np.float64, copy=True)
    t[np.isnan(t)] = 1.0
    t[na] = np.nan
    sc["twi"] = t
    sc["hor"] = ter["hor"] if hor is None else np.asarray(hor, dtype=np.float64)
    sc["svfa"] = ter["svfa"] if svf is None else np.asarray(svf, dtype=np.float64)
    sc["wsa"] = ter["wsa"] if wsa is None else np.asarray(wsa, dtype=np.float64)
    out = [int(bool(v)) for v in out]
    if reqhgt == 0:
        out = [a * b for a, b in zip((1, 0, 0, 1, 0, 1, 1, 1, 1, 1), out)]
    if reqhgt < 0:
        out = [a * b for a, b in zip((1, 0, 0, 1, 0, 0, 0, 0, 0, 0), out)]
    args = dict(obstime=micropoint["obstime"], climdata=clim, pointm=pointm, vegp=sv, soilc=sc, reqhgt=float(reqhgt),
                zref=float(micropoint["zref"]), lat=float(micropoint["lat"]), lon=float(micropoint["long"]),
                Sminp=getmode(sc["Smin"]), Smaxp=getmode(sc["Smax"]), tfact=1.5, complete=len(subs) == n,
                mat=float(micropoint["matemp"]), out=out)
    if dfsel is not None:
        args["dfsel"] = dfsel
    return args


def runmicro(micropoint: Mapping, reqhgt: float, vegp: Mapping, soilc: Mapping, dtm: Mapping, *, pai_a=None,
             tfact: float = 1.5, out: Sequence = (1,) * 10, slr=None, apr=None, hor=None, twi=None, wsa=None, svf=None,
             device: int = 0) -> dict:
    """`runmicro(micropoint, reqhgt, vegp, soilc, dtm, ...)` for data.frame weather without snow: time-invariant
    vegetation goes to runmicro1Cpp, layered vegetation to runmicro3Cpp (R/internal.R:3332-3346)."""
    a = prepare_grid_inputs(micropoint, reqhgt, vegp, soilc, dtm, pai_a=pai_a, out=out, slr=slr, apr=apr, hor=hor, twi=twi,
                            wsa=wsa, svf=svf, device=device)
    a["tfact"] = float(tfact)
    dfsel = a.pop("dfsel", None)
    if dfsel is None:
        return api.runmicro1Cpp(**a, device=device)
    return api.runmicro3Cpp(dfsel, **a, device=device)


# ---- array weather: runpointmodela() and runmicro() -> .runmodel2Cpp / .runmodel4Cpp ------------------------------
def block_reduce(a, crows: int, ccols: int, how: str = "mean"):
    """A fine raster [rows, cols(, layers)] summarised per coarse cell (the fine cells whose centres fall in it):
    stands for `.resampler(r2, r)` = terra aggregate + resample (R/internal.R:358-380; terra semantics unpinned)."""
    a = as3d(a)
    R, Cc, L = a.shape
    ri = np.minimum((np.arange(R) + 0.5) * crows / R, crows - 1).astype(int)
    ci = np.minimum((np.arange(Cc) + 0.5) * ccols / Cc, ccols - 1).astype(int)
    out = np.full((crows, ccols, L), np.nan)
    for i in range(crows):
        for j in range(ccols):
            blk = a[np.ix_(ri == i, ci == j)].reshape(-1, L)
            for l in range(L):
                v = blk[:, l]
                v = v[~np.isnan(v)]
                if v.size:
                    out[i, j, l] = getmode(v) if how == "mode" else v.mean()
    return out


def runpointmodela(climarray: Mapping, obstime: Mapping, reqhgt: float, dtm: Mapping, vegp: Mapping, soilc: Mapping, *,
                   lats, lons, matemp: float | None = None, zref: float = 2.0, windhgt: float = 2.0, soilm=None,
                   dTmx: float = 25.0, maxiter: int = 20, yearG: bool = True) -> list:
    """`runpointmodela(climarrayr, tme, reqhgt, dtm, vegp, soilc, ...)` (R/Cppwrappers.R:208-263): the point model once
    per cell of the coarse climate grid.  `climarray[k]`: [crows, ccols, T]; `lats`, `lons`: [crows, ccols] (the reference
    takes them from the climate raster's CRS).  Returns the row-major list of micropoints (None where the cell has no
    data), as the reference's `pointo`."""
    cr, cc, T = np.shape(climarray["temp"])
    mxhgt = float(np.nanmax(np.asarray(vegp["hgt"], dtype=np.float64)))
    wdir = np.array([getmode(np.asarray(climarray["winddir"])[:, :, k]) for k in range(T)])
    vc = {k: block_reduce(vegp[k], cr, cc) for k in VEG_KEYS}
    st = block_reduce(soilc["soiltype"], cr, cc, "mode")[:, :, 0]
    soiltype = int(getmode(st))
    gr = float(np.nanmean(np.asarray(soilc["groundr"], dtype=np.float64)))
    P = SOILPARAMETERS
    out = []
    for i in range(cr):
        for j in range(cc):
            if np.isnan(climarray["temp"][i, j, 0]) or np.isnan(vc["hgt"][i, j, 0]):
                out.append(None)
                continue
            w = {k: np.asarray(climarray[k])[i, j, :] for k in WEATHER if k != "winddir"}
            w["winddir"] = wdir
            w["obstime"] = obstime
            m = {k: float(np.mean(vc[k][i, j, :])) for k in VEG_KEYS}                                   # .tovp
            vegp_p = np.array([m["hgt"], m["pai"], m["x"], m["clump"], m["leafr"], m["leaft"], m["leafd"], 0.97, m["gsmax"], 100.0])
            sn = int(st[i, j]) - 1                                                                       # .togp
            groundp_p = np.array([gr, 0.0, 180.0, 0.97, P["rho"][sn], P["Vm"][sn], P["Vq"][sn], P["Mc"][sn], P["b"][sn],
                                  P["psi_e"][sn], P["Smax"][sn], P["Smin"][sn], P["Smin"][sn], P["Smin"][sn], P["Smin"][sn]])
            out.append(runpointmodel(w, reqhgt, dtm, vegp, soilc, zref=zref, windhgt=windhgt,
                                     soilm=None if soilm is None else np.asarray(soilm)[i, j, :], matemp=matemp, dTmx=dTmx,
                                     maxiter=maxiter, yearG=yearG, lat=float(lats[i, j]), long=float(lons[i, j]),
                                     vegp_p=vegp_p, groundp_p=groundp_p, soiltype=soiltype, mxhgt=mxhgt))
    return out


def prepare_grid_inputs_array(micropointa: Sequence, crows: int, ccols: int, reqhgt: float, vegp: Mapping, soilc: Mapping,
                              dtm: Mapping, *, lats, lons, pai_a=None, out: Sequence = (1,) * 10, slr=None, apr=None,
                              hor=None, twi=None, wsa=None, svf=None, device: int = 0) -> dict:
    """What `.runmodel2Cpp` / `.runmodel4Cpp` prepare (R/internal.R:1175-1343) with the climate and
    point-model variables left on the coarse grid: the solver interpolates them (array_forcing == 2; the altitude
    correction is applied there too, see runmicro_array).  `lats`, `lons`:
    [rows, cols] of the fine raster (`.latslonsfromr(dtm)`).  Cells of the coarse grid without a micropoint are not
    supported (the reference fills them with NA and lets `resample` look around them)."""
    if any(m is None for m in micropointa):
        raise ValueError("every coarse cell needs a micropoint")
    first = micropointa[0]
    base = prepare_grid_inputs(first, reqhgt, vegp, soilc, dtm, pai_a=pai_a, out=out, slr=slr, apr=apr, hor=hor, twi=twi,
                               wsa=wsa, svf=svf, device=device)
    T = len(first["weather"]["temp"])

    def grid(get):
        a = np.empty((crows, ccols, T), order="F")
        for k, m in enumerate(micropointa):
            a[k // ccols, k % ccols, :] = get(m)
        return a
    clim = {k: grid(lambda m, k=k: m["weather"][k]) for k in ("temp", "relhum", "pres", "swdown", "difrad", "lwdown",
                                                                "windspeed", "winddir")}
    pm = {"soilm": "soilm", "Gp": "G", "umu": "umu", "kp": "kp", "muGp": "muGp", "dtrp": "dtrp"}
    pointm = {k: grid(lambda m, v=v: m["dfo"][v]) for k, v in pm.items()}
    base.update(climdata=clim, pointm=pointm, lat=np.asfortranarray(lats, dtype=np.float64),
                lon=np.asfortranarray(lons, dtype=np.float64), zref=float(first["zref"]),
                mat=float(np.mean([m["matemp"] for m in micropointa])))
    return base


def runmicro_array(micropointa: Sequence, crows: int, ccols: int, reqhgt: float, vegp: Mapping, soilc: Mapping, dtm: Mapping,
                   *, lats, lons, tfact: float = 1.5, altcorrect: int = 0, dtmc=None, device: int = 0, **kw) -> dict:
    """`runmicro()` for a list of micropoints from `runpointmodela` (array weather, no snow).  `altcorrect` 1 / 2 with
    `dtmc` [crows, ccols], the elevations of the climate cells: the lapse-rate correction of R/internal.R:1233-1251."""
    if reqhgt < 0:
        raise ValueError("coarse array forcing below ground needs the per-cell point-model series: not mirrored")
    a = prepare_grid_inputs_array(micropointa, crows, ccols, reqhgt, vegp, soilc, dtm, lats=lats, lons=lons, device=device, **kw)
    a["tfact"] = float(tfact)
    dfsel = a.pop("dfsel", None)
    R, Cc = a["vegp"]["hgt"].shape[:2]
    coarse = {"rowpos": api.coarse_positions(R, crows), "colpos": api.coarse_positions(Cc, ccols)}
    if altcorrect:
        coarse.update(altcorrect=int(altcorrect), dtmc=dtmc, dtm=cleanvars(vegp, soilc, dtm["z"])[2])
    order = ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp", "tfact",
             "complete", "mat", "out")
    fn = "mcf_runmicro2" if dfsel is None else "mcf_runmicro4"
    return api._run(fn, True, *[a[k] for k in order], device, 0, 0, dfsel, coarse)


# ---- snow: runsnowmodel() -> .snowmodel1 --------------------------------------------------------------------------
def cleanvegp(vegp):
    """`.cleanvegp` (R/internal.R:121-139): no plant area on zero-height cells and the other way round, so that the
    snow model does not return NAs"""
    veg = {k: as3d(vegp[k]).copy() for k in VEG_KEYS}
    hm = veg["hgt"][:, :, 0]
    for l in range(veg["pai"].shape[2]):
        pm = veg["pai"][:, :, l]
        with np.errstate(invalid="ignore"):
            hm[(pm == 0) & (hm > 0)] = 0
            pm[(hm == 0) & (pm > 0)] = 0
    return {k: (v[:, :, 0] if np.ndim(vegp[k]) == 2 else v) for k, v in veg.items()}


def sortl(vegp, sdep):
    """`.sortl` (R/internal.R:2388-2419): pai, hgt, leaft, clump averaged over the layers in force while snow lies
    (weights = number of snow-covered steps each layer serves); single-layer variables pass through"""
    sdep = np.asarray(sdep, dtype=np.float64)
    out = {}
    for k in ("pai", "hgt", "leaft", "clump"):
        a = as3d(vegp[k])
        dmx = a.shape[2]
        if dmx == 1:
            out[k] = a[:, :, 0]
            continue
        s = layer_index(dmx, len(sdep))
        sel = np.nonzero(sdep > 0)[0]
        s = s[sel] if len(sel) else s[:1]
        num, fre = np.unique(s, return_counts=True)
        # the reference sums a[,,j] * fre[j] with j = 1..length(num) — the FIRST layers, not layers `num`
        m = np.zeros(a.shape[:2])
        for j in range(len(num)):
            m = m + a[:, :, j] * fre[j]
        out[k] = m / fre.sum()
    return out


def runsnowmodel(weather: Mapping, micropoint: Mapping, vegp: Mapping, soilc: Mapping, dtm: Mapping, *,
                 snowenv: str = "Taiga", method: str = "fast", snowinitd=0.0, snowinita=0.0, zref: float = 2.0,
                 windhgt: float | None = None, stfact: float = 0.01, device: int = 0) -> dict:
    """`runsnowmodel(weather, micropoint, vegp, soilc, dtm, ...)` for data.frame weather (R/Cppwrappers.R:717-735);
    `weather` is always the complete hourly series.  A complete micropoint runs `.snowmodel1` (R/internal.R:2498-2619)
    at the point model's reference height: weather height adjustment, the snow point model (host C++), `.sortl`, the
    5-day chunk loop on the device (terrain refresh from dtm + snow, gridmodelsnow1, `.tpicalc` redistribution,
    hand-over).  A subset micropoint runs, at `zref` / `windhgt`, either that and its subset (`method = "slow"`) or
    `.snowmodelq1` (:2627-2776, `method = "fast"`, the reference's default): the point model over every hour, the grid
    model on the selected days only, the pack carried between them by the point model's balance.
    Returns Tc, Tg, groundsnowdepth, totalSWE, snowden, umu."""
    from . import snow as S
    if method not in ("fast", "slow"):
        raise ValueError('method is "fast" or "slow"')
    vegp = cleanvegp(vegp)
    w = {k: np.array(weather[k], dtype=np.float64, copy=True) for k in WEATHER if k in weather}
    tme = weather["obstime"]
    obstime = {k: np.asarray(tme[k]) for k in ("year", "month", "day", "hour")}
    subset = len(micropoint["subs"]) != micropoint["ntme"]
    if subset:
        zref = float(zref)
        windhgt = zref if windhgt is None else float(windhgt)
    else:
        zref = windhgt = float(micropoint["zref"])                     # runsnowmodel passes micropoint$zref twice
    if zref != windhgt:                                                # R/internal.R:2505-2507, 2636-2638
        w["windspeed"] = w["windspeed"] * np.log(67.8 * zref - 5.42) / np.log(67.8 * windhgt - 5.42)
    lat, long = float(micropoint["lat"]), float(micropoint["long"])
    z = np.asarray(dtm["z"], dtype=np.float64)
    hmax = float(np.nanmax(np.asarray(vegp["hgt"], dtype=np.float64)))
    if hmax > zref:                                                    # R/internal.R:2509-2526
        w.update({k: v for k, v in pointmodel.weatherhgtCpp(obstime, w, zref, zref, hmax, lat, long).items() if k in w})
        zref = hmax
    vp = sortvegp_point(vegp)                                          # (hgt, pai, x, clump, leafr, leaft, ...)
    vegpp = np.array([vp[1], vp[0], vp[5], vp[3]])                     # c(vegpp[2], vegpp[1], vegpp[6], vegpp[4])
    sdep = z * 0 + snowinitd
    sage = z * 0 + snowinita
    other_p = [0.0, 0.0, lat, long, zref, float(np.nanmean(sdep)), float(np.nanmean(sage))]
    hour_int = {**obstime, "hour": np.floor(np.asarray(obstime["hour"], dtype=np.float64))}   # `hour = tme$hour`
    fast = subset and method == "fast"
    pmod = pointmodel.pointmodelsnow(hour_int, w, vegpp, other_p, snowenv, maxiter=20 if fast else 100)
    n = len(w["temp"])
    pointm = {"Gp": pmod["G"], "Tc": pmod["Tc"], "RswabsG": pmod["RswabsG"], "RlwabsG": pmod["RlwabsG"], "umu": pmod["umu"],
              "tr": pmod["tr"]}
    vg = sortl(vegp, pmod["sdepc"][:n])
    res = dtm["res"]
    xres = res if np.isscalar(res) else res[0]
    clim = {k: w[k] for k in ("temp", "relhum", "pres", "swdown", "difrad", "lwdown", "windspeed", "winddir", "precip")}
    if fast:
        subs = np.asarray(micropoint["subs"], dtype=np.int64)          # 1-based positions, as in the reference
        ai = subs - 1
        vg["leaft"] = np.where(np.isnan(vg["leaft"]), 0.01, vg["leaft"])
        other = {"zref": zref, "lat": lat, "lon": long, "isnowdc": snowinitd * z, "isnowac": sage, "isnowag": sage}
        out = S.snowmodelq1_days(_rows(hour_int, ai), _rows(clim, ai), _rows(pointm, ai), pmod, w["temp"],
                                 np.where(w["temp"] > 2, 0.0, w["precip"]), subs, vg, other, snowenv, z, xres, stfact,
                                 device=device)
        out["umu"] = pmod["umu"][ai]
        return out
    other = {"zref": zref, "lat": lat, "lon": long, "isnowdc": sdep, "isnowac": sage, "isnowdg": sdep * 0.5, "isnowag": sage}
    out = S.snowmodel1_chunks(hour_int, clim, pointm, vg, other, snowenv, z, xres, stfact, device=device)
    out["umu"] = pmod["umu"]
    if subset:                                                         # method = "slow": the full model, then its subset
        out = subsetsnowmodel(out, micropoint["subs"])
    return out


def runsnowmodela(climarray: Mapping, obstime: Mapping, micropointa: Sequence, vegp: Mapping, soilc: Mapping, dtm: Mapping, *,
                  dtmc, lats_c, lons_c, lats, lons, altcorrect: int = 0, snowenv: str = "Taiga", method: str = "fast",
                  snowinitd: float = 0.0, snowinita: float = 0.0, zref: float = 2.0, windhgt: float | None = None,
                  stfact: float = 0.01, device: int = 0) -> dict:
    """`runsnowmodel(climarrayr, micropointa, vegp, soilc, dtm, dtmc, tme, altcorrect, ...)` for array weather
    (R/Cppwrappers.R:735-757 -> `.snowmodel2`, R/internal.R:2777-3013): the snow point model (host C++) once per cell of
    the climate grid, then `snow.snowmodel2_chunks`.  `climarray[k]`: [crows, ccols, T]; `dtmc`, `lats_c`, `lons_c`:
    [crows, ccols] of the climate grid; `lats`, `lons`: [rows, cols] of the fine raster.  A subset micropoint list with
    `method = "slow"` runs the whole series and subsets it (here `umu` too along time; the reference indexes the array as
    a vector there); `method = "fast"` runs `.snowmodelq2` (R/internal.R:3017-3283) = `snow.snowmodelq2_days`.  As in the reference every climate cell needs
    data, and vegetation taller than `zref` fails (`.snowmodel2` stops at R/internal.R:2838, `climdfr` not found)."""
    from . import snow as S
    vegp = cleanvegp(vegp)
    if any(m is None for m in micropointa):
        raise ValueError("every coarse cell needs a micropoint")
    last = micropointa[-1]                                            # the reference's loop keeps the last one's subs
    subset = len(last["subs"]) != last["ntme"]
    if method not in ("fast", "slow"):
        raise ValueError('method is "fast" or "slow"')
    fast = subset and method == "fast"
    if subset:
        zref = float(zref)
        windhgt = zref if windhgt is None else float(windhgt)
    else:
        zref = windhgt = float(micropointa[0]["zref"])
    cr, cc, T = np.shape(climarray["temp"])
    z = np.asarray(dtm["z"], dtype=np.float64)
    R, Cc = z.shape
    if float(np.nanmax(np.asarray(vegp["hgt"], dtype=np.float64))) > zref:
        raise ValueError("the array snow model needs zref at or above the tallest vegetation (the reference fails there)")
    wdir = np.array([getmode(np.asarray(climarray["winddir"])[:, :, k]) for k in range(T)])
    vc = {k: block_reduce(vegp[k], cr, cc) for k in ("pai", "hgt", "leaft", "clump")}
    if np.isnan(np.asarray(climarray["temp"])[:, :, 0]).any() or np.isnan(vc["hgt"][:, :, 0]).any():
        raise ValueError("every coarse cell needs climate data and vegetation")
    ob = {k: np.asarray(obstime[k]) for k in ("year", "month", "day", "hour")}
    clim_c = {k: np.array(climarray[k], dtype=np.float64, order="F", copy=True) for k in WEATHER if k != "winddir"}
    if zref != windhgt:
        clim_c["windspeed"] *= np.log(67.8 * zref - 5.42) / np.log(67.8 * windhgt - 5.42)
    clim_c["winddir"] = wdir
    pn = {"Gp": "G", "Tc": "Tc", "RswabsG": "RswabsG", "RlwabsG": "RlwabsG", "umu": "umu", "tr": "tr", "sdepc": "sdepc"}
    if fast:                                                          # `.snowmodelq2`: depth after the step, the melt terms
        pn.update({k: k for k in ("sublmelt", "tempmelt", "rainmelt", "sstemp", "sdenc", "sdeng")})
    pointm_c = {k: np.empty((cr, cc, T), order="F") for k in pn}
    for i in range(cr):
        for j in range(cc):
            w = {k: clim_c[k][i, j, :] for k in clim_c if k != "winddir"}
            w["winddir"] = wdir
            vegpp = [float(np.mean(vc[k][i, j, :])) for k in ("pai", "hgt", "leaft", "clump")]          # `.tovp`
            pmod = pointmodel.pointmodelsnow(ob, w, vegpp, [0.0, 0.0, float(lats_c[i, j]), float(lons_c[i, j]), zref, snowinitd,
                                                            snowinita], snowenv, maxiter=10)
            for k, v in pn.items():
                pointm_c[k][i, j, :] = pmod[v][1:T + 1] if fast and k == "sdepc" else pmod[v][:T]
    res = dtm["res"]
    xres = res if np.isscalar(res) else res[0]
    rowpos, colpos = api.coarse_positions(R, cr), api.coarse_positions(Cc, cc)
    if fast:                                                          # R/internal.R:3017-3283
        subs = np.asarray(last["subs"], dtype=np.int64)
        ai = subs - 1
        pm2_c = {k: pointm_c[k] for k in ("sublmelt", "tempmelt", "rainmelt", "sstemp", "sdenc", "sdeng")}
        pm2_c["tc"] = clim_c["temp"]
        pm2_c["snow"] = np.where(clim_c["temp"] > 2, 0.0, clim_c["precip"])
        sel = lambda d: {k: (np.asarray(v)[ai] if np.ndim(v) == 1 else np.asfortranarray(np.asarray(v)[:, :, ai]))   # noqa: E731
                         for k, v in d.items()}
        pm_s = sel({k: pointm_c[k] for k in ("Gp", "Tc", "RswabsG", "RlwabsG", "umu", "tr", "sdepc")})
        vg = sortl(vegp, np.max(pm_s["sdepc"], axis=(0, 1)))
        other = {"zref": zref, "lats": np.asarray(lats, dtype=np.float64), "lons": np.asarray(lons, dtype=np.float64),
                 "isnowdc": z * 0 + snowinitd, "isnowac": z * 0 + snowinita, "isnowag": z * 0 + snowinita}
        return S.snowmodelq2_days(sel(ob), sel(clim_c), pm_s, pm2_c, subs, vg, other, snowenv, z, np.asarray(dtmc, dtype=np.float64),
                                  xres, stfact, rowpos=rowpos, colpos=colpos, altcorrect=altcorrect, device=device)
    vg = sortl(vegp, np.max(pointm_c["sdepc"], axis=(0, 1)))
    sdep = z * 0 + snowinitd
    sage = z * 0 + snowinita
    other = {"zref": zref, "lats": np.asarray(lats, dtype=np.float64), "lons": np.asarray(lons, dtype=np.float64),
             "isnowdc": sdep, "isnowac": sage, "isnowdg": sdep * 0.5, "isnowag": sage}
    out = S.snowmodel2_chunks(ob, clim_c, pointm_c, vg, other, snowenv, z, np.asarray(dtmc, dtype=np.float64), xres, stfact,
                              rowpos=rowpos, colpos=colpos, altcorrect=altcorrect,
                              agg=10 if xres <= 100 and min(cr, cc) >= 10 else 1, device=device)
    if subset:
        i = np.asarray(last["subs"], dtype=np.int64) - 1
        out = {k: v[:, :, i] for k, v in out.items()}
    return out


# ---- snow: runmicro(snow = TRUE) -> .runmicrosnow1 ----------------------------------------------------------------
def _rows(d: Mapping, ai):
    return {k: np.asarray(v)[ai] for k, v in d.items()}


def subsetpointmodel(micropoint: Mapping, tstep: str = "month", what: str = "tmax", days=None, Tc=None) -> dict:
    """`subsetpointmodel` (R/dataprep.R:31-103): whole days of the point model — given 1-based `days`, or one day per
    month / year chosen by the point model's canopy temperature (`what` = tmax, tmin, tmedian)."""
    dfo, ob = micropoint["dfo"], micropoint["obstime"]
    if days is not None:
        days = np.asarray(days, dtype=np.int64)
        ai = (np.repeat((days - 1) * 24, 24) + np.tile(np.arange(24), days.size)).astype(np.int64)
    else:
        tc = np.asarray(dfo["Tc"] if Tc is None else Tc, dtype=np.float64)
        yr, mo, dy = (np.asarray(ob[k]).astype(int) for k in ("year", "month", "day"))

        def extract(sel):
            v = tc[sel]
            if what == "tmax":
                s2 = int(np.argmax(v))
            elif what == "tmin":
                s2 = int(np.argmin(v))
            elif what == "tmedian":
                o = np.argsort(v, kind="stable")
                s2 = int(o[len(o) // 2 - 1])                       # o[trunc(length(o) / 2)], 1-based
            else:
                raise ValueError("what must be one of tmax, tmin or tmedian")
            k = sel[s2]
            return np.nonzero((yr == yr[k]) & (mo == mo[k]) & (dy == dy[k]))[0]
        parts = []
        for y in dict.fromkeys(yr.tolist()):
            sely = np.nonzero(yr == y)[0]
            if tstep == "year":
                parts.append(extract(sely))
            elif tstep == "month":
                # `which(tme$mon[sely] == m)` indexes WITHIN the year's rows; the reference then uses those positions
                # on the whole series — the same thing for the first year only
                for m in dict.fromkeys(mo[sely].tolist()):
                    parts.append(extract(np.nonzero(mo[sely] == m)[0]))
            else:
                raise ValueError("tstep must be month or year")
        ai = np.concatenate(parts)
    out = dict(micropoint)
    out["dfo"] = _rows(dfo, ai)
    out["weather"] = _rows(micropoint["weather"], ai)
    out["obstime"] = _rows(ob, ai)
    out["subs"] = np.asarray(micropoint["subs"])[ai]
    if micropoint.get("Tbz") is not None:
        out["Tbz"] = np.asarray(micropoint["Tbz"])[ai]
    return out


def subsetpointmodela(micropointa: Sequence, tstep: str = "month", what: str = "tmax", days=None) -> list:
    """`subsetpointmodela` (R/dataprep.R:105-138): the same days for every cell of the climate grid — chosen, as there,
    from the mean canopy temperature over the cells that have a point model"""
    have = [m for m in micropointa if m is not None]
    if days is None:
        tc = np.mean([np.asarray(m["dfo"]["Tc"], dtype=np.float64) for m in have], axis=0)
        return [None if m is None else subsetpointmodel(m, tstep, what, Tc=tc) for m in micropointa]
    return [None if m is None else subsetpointmodel(m, days=days) for m in micropointa]


def subsetsnowmodel(smod: Mapping, subs) -> dict:
    """`subsetsnowmodel` (R/dataprep.R:148-160); `subs` 1-based.  (The reference tests `snowmods$umu`, which does not
    exist yet, so `umu` is always subset as a vector.)"""
    i = np.asarray(subs, dtype=np.int64) - 1
    out = {k: np.asarray(smod[k])[:, :, i] for k in ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden")}
    out["umu"] = np.asarray(smod["umu"])[i]
    return out


def sortl2(vegp, sdep, reqhgt, pai_a=None):
    """`.sortl2` (R/internal.R:2421-2482): as `.sortl` for pai, hgt, leaft, clump, leafd, plus paia and leafden"""
    sdep = np.asarray(sdep, dtype=np.float64)
    out = {}
    for k in ("pai", "hgt", "leaft", "clump", "leafd"):
        a = as3d(vegp[k])
        dmx = a.shape[2]
        if dmx == 1:
            out[k] = a[:, :, 0]
            continue
        s = layer_index(dmx, len(sdep))
        sel = np.nonzero(sdep > 0)[0]
        s = s[sel] if len(sel) else s[:1]
        num, fre = np.unique(s, return_counts=True)
        m = np.zeros(a.shape[:2])
        for j in range(len(num)):
            m = m + a[:, :, j] * fre[j]
        out[k] = m / fre.sum()
    if pai_a is not None and as3d(pai_a).shape[2] > 1:
        pai_a = sortl({"pai": pai_a, "hgt": pai_a, "leaft": pai_a, "clump": pai_a}, sdep)["pai"]
    elif pai_a is not None:
        pai_a = as3d(pai_a)[:, :, 0]
    out["leafden"], out["paia"] = foliageden(reqhgt, out["hgt"], out["pai"], pai_a)
    return out


def runmicro_snow(micropoint: Mapping, reqhgt: float, vegp: Mapping, soilc: Mapping, dtm: Mapping, smod: Mapping, *,
                  pai_a=None, tfact: float = 1.5, out: Sequence = (1,) * 10, device: int = 0,
                  _solve=None, _microsnow=None, _terrain=None) -> dict:
    """`runmicro(..., snow = TRUE, snowmod = smod)` for data.frame weather = `.runmicrosnow1` (R/internal.R:3581-3659):
    days with no snow anywhere go through the ordinary solver, days with snow through gridmicrosnow1 (which keeps the
    ordinary solver's values on snow-free cell-steps of days that have both), and the two are merged by day.
    `_solve` / `_microsnow` / `_terrain` let the tests put their checkers behind the same orchestration."""
    from . import snow as S
    solve = runmicro if _solve is None else _solve
    microsnow = S.gridmicrosnow1 if _microsnow is None else _microsnow
    veg, soil, z = cleanvars(vegp, soilc, dtm["z"])
    sm = dict(smod)
    swe = np.array(sm["totalSWE"], dtype=np.float64, copy=True)
    swe[np.isnan(swe)] = 0.0
    swe[np.isnan(z)] = np.nan                                            # mask(totalSWE, dtm)
    sm["totalSWE"] = swe
    sd = S.snowdaysfun(S.applycpp3(swe, "max", device=device), S.applycpp3(swe, "min", device=device))
    alldays = np.arange(1, len(sd["snowdays"]) + 1)
    snowdays, nosnowdays = alldays[sd["snowdays"] == 1], alldays[sd["nosnowdays"] == 1]
    rows, cols = z.shape
    if len(nosnowdays):
        moutn = solve(subsetpointmodel(micropoint, days=nosnowdays), reqhgt, vegp, soilc, dtm, pai_a=pai_a, tfact=tfact,
                      out=out, device=device)
    else:                                                                # .createblanktemplate1
        moutn = solve(subsetpointmodel(micropoint, days=[1]), reqhgt, vegp, soilc, dtm, tfact=1.5, out=out, device=device)
        moutn = {k: v * np.nan for k, v in moutn.items()}
    if not len(snowdays):
        return moutn
    mps = subsetpointmodel(micropoint, days=snowdays)
    ai = (np.repeat((snowdays - 1) * 24, 24) + np.tile(np.arange(24), snowdays.size)).astype(np.int64)
    w = dict(mps["weather"])
    w["umu"] = np.asarray(sm["umu"])[ai]
    sdept = np.zeros(micropoint["ntme"])
    sdept[np.asarray(mps["subs"]) - 1] = 1
    vg = sortl2(veg, sdept, reqhgt, pai_a)
    res = dtm["res"]
    xres = res if np.isscalar(res) else res[0]
    ter = (terrain.precompute_terrain(z, xres, micropoint["zref"], device=device) if _terrain is None
           else _terrain(z, xres, micropoint["zref"]))
    other = {"slope": ter["slope"], "aspect": ter["aspect"], "hor": ter["hor"], "skyview": ter["svfa"], "wsa": ter["wsa"],
             "lat": float(micropoint["lat"]), "lon": float(micropoint["long"]), "zref": float(micropoint["zref"]),
             "Smax": soilinit(soil)["Smax"]}
    t1 = len(snowdays) * 24
    s1 = np.arange(t1)[np.repeat(np.isin(snowdays, nosnowdays), 24)]
    s2 = np.arange(len(nosnowdays) * 24)[np.repeat(np.isin(nosnowdays, snowdays), 24)]
    micro = {}
    for k, v in moutn.items():
        a = np.full((rows, cols, t1), np.nan, order="F")
        if len(s1):
            a[:, :, s1] = np.asarray(v)[:, :, s2]
        micro[k] = a
    outm = [int(bool(v)) for v in out]
    if reqhgt == 0:
        outm = [1 if i in (0, 3, 5, 6, 7, 8, 9) else 0 for i in range(10)]
    elif reqhgt < 0:
        outm = [1 if i in (0, 3) else 0 for i in range(10)]
    smods = subsetsnowmodel(sm, ai + 1)
    mouts = microsnow(reqhgt, mps["obstime"], w, smods, micro, vg, other, float(micropoint["matemp"]), outm)
    return S.merge_snow_outputs(moutn, mouts, snowdays, nosnowdays, rows, cols)



def runmicro_snow_array(micropointa: Sequence, crows: int, ccols: int, reqhgt: float, vegp: Mapping, soilc: Mapping,
                        dtm: Mapping, smod: Mapping, *, dtmc, lats, lons, altcorrect: int = 0, pai_a=None, tfact: float = 1.5,
                        out: Sequence = (1,) * 10, device: int = 0, _solve=None, _microsnow=None, _terrain=None) -> dict:
    """`runmicro(..., snow = TRUE, snowmod = smod)` for array weather = `.runmicrosnow2` with `.prepsnowinputs2`
    (R/internal.R:3661-3742, 3445-3579): as `runmicro_snow`, with the no-snow days through the coarse-array solver and
    the snow days through gridmicrosnow2 on climate resampled to the fine raster (host numpy, as the reference does it in
    R; relative humidity from resampled vapour pressure, kept within 20 .. 100 %).  Reference behaviour kept: `.sortl2`
    sees no snow-covered step here (`micropoints$subs` of a list is NULL), so the vegetation layers are weighted as
    without snow."""
    from . import snow as S
    from .rformulas import lapserate_R, satvap_R, upsample_coarse
    if any(m is None for m in micropointa):
        raise ValueError("every coarse cell needs a micropoint")
    microsnow = S.gridmicrosnow2 if _microsnow is None else _microsnow

    def solve(mpa, **kw):
        if _solve is not None:
            return _solve(mpa, reqhgt, **kw)
        return runmicro_array(mpa, crows, ccols, reqhgt, vegp, soilc, dtm, lats=lats, lons=lons, altcorrect=altcorrect, dtmc=dtmc,
                              device=device, **kw)
    veg, soil, z = cleanvars(vegp, soilc, dtm["z"])
    hole = np.isnan(z)
    sm = dict(smod)
    swe = np.array(sm["totalSWE"], dtype=np.float64, copy=True)
    swe[np.isnan(swe)] = 0.0
    swe[hole] = np.nan
    sm["totalSWE"] = swe
    sd = S.snowdaysfun(S.applycpp3(swe, "max", device=device), S.applycpp3(swe, "min", device=device))
    alldays = np.arange(1, len(sd["snowdays"]) + 1)
    snowdays, nosnowdays = alldays[sd["snowdays"] == 1], alldays[sd["nosnowdays"] == 1]
    rows, cols = z.shape
    if len(nosnowdays):
        moutn = solve([subsetpointmodel(m, days=nosnowdays) for m in micropointa], pai_a=pai_a, tfact=tfact, out=out)
    else:                                                                # .createblanktemplate2
        moutn = {k: v * np.nan for k, v in solve([subsetpointmodel(m, days=[1]) for m in micropointa], tfact=1.5, out=out).items()}
    if not len(snowdays):
        return moutn
    mps = [subsetpointmodel(m, days=snowdays) for m in micropointa]
    h = len(snowdays) * 24
    ai = (np.repeat((snowdays - 1) * 24, 24) + np.tile(np.arange(24), snowdays.size)).astype(np.int64)

    def grid(get):
        a = np.empty((crows, ccols, h), order="F")
        for k, m in enumerate(mps):
            a[k // ccols, k % ccols, :] = get(m)
        return a
    rowpos, colpos = api.coarse_positions(rows, crows), api.coarse_positions(cols, ccols)
    up = lambda a: upsample_coarse(a, rowpos, colpos)                                          # noqa: E731
    cca = lambda a: np.where(hole[:, :, None], np.nan, up(a))                                  # noqa: E731
    wc = {k: grid(lambda m, k=k: m["weather"][k]) for k in WEATHER}
    with np.errstate(invalid="ignore"):
        ea = up(satvap_R(wc["temp"]) * (wc["relhum"] / 100))
        temp = up(wc["temp"])
        if altcorrect == 0:
            pres = up(wc["pres"])
        else:
            zc = np.nan_to_num(np.asarray(dtmc, dtype=np.float64), nan=0.0)
            pres = up(wc["pres"] / (((293 - 0.0065 * zc[:, :, None]) / 293) ** 5.26)) * (((293 - 0.0065 * z[:, :, None]) / 293) ** 5.26)
            elevd = (up(zc) - z)[:, :, None]
            temp = (5 / 1000 if altcorrect == 1 else lapserate_R(temp, ea, pres)) * elevd + temp
        relhum = np.clip((ea / satvap_R(temp)) * 100, 20.0, 100.0)
        wd = wc["winddir"] * np.pi / 180
        wu, wv = wc["windspeed"] * np.cos(wd), wc["windspeed"] * np.sin(wd)
        w = {"temp": temp, "relhum": relhum, "pres": pres, "swdown": cca(wc["swdown"]), "difrad": cca(wc["difrad"]),
             "lwdown": cca(wc["lwdown"]), "windspeed": np.sqrt(up(wu) ** 2 + up(wv) ** 2),
             "winddir": (np.arctan2(np.nanmean(wv, axis=(0, 1)), np.nanmean(wu, axis=(0, 1))) * 180 / np.pi) % 360,
             "precip": cca(wc["precip"]), "umu": cca(grid(lambda m: m["dfo"]["umu"]))}
    last = micropointa[-1]
    vg = sortl2(veg, np.zeros(last["ntme"]), reqhgt, pai_a)
    res = dtm["res"]
    xres = res if np.isscalar(res) else res[0]
    zref = float(last["zref"])
    ter = terrain.precompute_terrain(z, xres, zref, device=device) if _terrain is None else _terrain(z, xres, zref)
    other = {"slope": ter["slope"], "aspect": ter["aspect"], "hor": ter["hor"], "skyview": ter["svfa"], "wsa": ter["wsa"],
             "lat": np.asarray(lats, dtype=np.float64), "lon": np.asarray(lons, dtype=np.float64), "zref": zref,
             "Smax": soilinit(soil)["Smax"]}
    s1 = np.arange(h)[np.repeat(np.isin(snowdays, nosnowdays), 24)]
    s2 = np.arange(len(nosnowdays) * 24)[np.repeat(np.isin(nosnowdays, snowdays), 24)]
    micro = {}
    for k, v in moutn.items():
        a = np.full((rows, cols, h), np.nan, order="F")
        if len(s1):
            a[:, :, s1] = np.asarray(v)[:, :, s2]
        micro[k] = a
    outm = [int(bool(v)) for v in out]
    if reqhgt == 0:
        outm = [1 if i in (0, 3, 5, 6, 7, 8, 9) else 0 for i in range(10)]
    elif reqhgt < 0:
        outm = [1 if i in (0, 3) else 0 for i in range(10)]
    smods = {k: np.asfortranarray(np.asarray(sm[k])[:, :, ai]) for k in ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden")}
    mouts = microsnow(reqhgt, mps[0]["obstime"], w, smods, micro, vg, other, float(np.mean([m["matemp"] for m in micropointa])), outm)
    return S.merge_snow_outputs(moutn, mouts, snowdays, nosnowdays, rows, cols)


# ---- runbioclim() -> .runbioclim1 / .runbioclim3 ------------------------------------------------------------------
def biosel(obstime: Mapping, tc) -> dict:
    """`.biosel` (R/internal.R:1690-1729): the fourteen days a bioclim run models — for each month the day of median
    daily-mean temperature, then the hottest and the coldest day (of the median year when there are several).
    `seld`: 1-based day numbers, `selh`: 0-based hour indices."""
    tcd = np.asarray(tc, dtype=np.float64).reshape(-1, 24).mean(axis=1)
    mon = np.asarray(obstime["month"]).astype(int).reshape(-1, 24)[:, 12]       # the day's mean time falls on the day itself
    yr = np.asarray(obstime["year"]).astype(int).reshape(-1, 24)[:, 12]
    sel = []
    for m in range(1, 13):
        s = np.nonzero(mon == m)[0]
        o = np.argsort(tcd[s], kind="stable")
        sel.append(s[0] + o[len(o) // 2 - 1])                                   # s[1] - 1 + o[trunc(length(o) / 2)]
    mx, mn = [], []
    for y in dict.fromkeys(yr.tolist()):
        s = np.nonzero(yr == y)[0]
        mx.append(s[0] + int(np.argmax(tcd[s])))
        mn.append(s[0] + int(np.argmin(tcd[s])))
    mx = np.array(mx)[np.argsort(tcd[mx], kind="stable")]
    mn = np.array(mn)[np.argsort(tcd[mn], kind="stable")]
    k = len(mx) // 2                                                            # element trunc(n / 2) + 1, 1-based
    seld = np.array(sel + [mx[k], mn[k]]) + 1
    selh = (np.repeat((seld - 1) * 24, 24) + np.tile(np.arange(24), seld.size)).astype(np.int64)
    return {"seld": seld, "selh": selh}


def _quarter(values, months, how):
    """which.max / which.min of the circular three-month mean of the monthly aggregate (R/internal.R:1797-1802)"""
    ms = sorted(set(months.tolist()))
    agg = np.array([how(values[months == m]) for m in ms])
    sm = (np.roll(agg, 1) + agg + np.roll(agg, -1)) / 3
    return sm


def runbioclim(climdata: Mapping, reqhgt: float, vegp: Mapping, soilc: Mapping, dtm: Mapping, *, temp: str = "air",
               zref: float = 2.0, windhgt: float | None = None, soilm=None, pai_a=None, tfact: float = 1.5,
               out: Sequence = (1,) * 19, vegpisannual: bool = True, device: int = 0, _bioclim=None, _terrain=None) -> dict:
    """`runbioclim(climdata, reqhgt, vegp, soilc, dtm, ...)` for data.frame weather (R/Cppwrappers.R:628-651 ->
    `.runbioclim1` / `.runbioclim3`, R/internal.R:1776-1880 / 2083-2200): fourteen days are modelled (`biosel`) and the
    19 bioclim variables reduced from them on the device (mcf_runbioclim1 / 3).  Returns {bio1.. : [rows, cols]}."""
    ob_all = {k: np.asarray(climdata["obstime"][k]) for k in ("year", "month", "day", "hour")}
    mon_all = ob_all["month"].astype(int)
    t_all, p_all = np.asarray(climdata["temp"], dtype=np.float64), np.asarray(climdata["precip"], dtype=np.float64)
    pq = _quarter(p_all, mon_all, np.nanmean)
    tq = _quarter(t_all, mon_all, np.nansum)
    wq, dq, hq, cq = int(np.argmax(pq)) + 1, int(np.argmin(pq)) + 1, int(np.argmax(tq)) + 1, int(np.argmin(tq)) + 1
    sel = biosel(ob_all, t_all)
    w2 = {k: np.asarray(climdata[k])[sel["selh"]] for k in WEATHER}
    w2["obstime"] = {k: v[sel["selh"]] for k, v in ob_all.items()}
    veg, soil, z = cleanvars(vegp, soilc, dtm["z"])
    mp = runpointmodel(w2, reqhgt, dtm, vegp, soilc, zref=zref, windhgt=windhgt, soilm=soilm, yearG=False)
    layered = vegpdmx(veg) > 1
    ter_kw = {}
    if _terrain is not None:
        t = _terrain(z, dtm["res"] if np.isscalar(dtm["res"]) else dtm["res"][0], mp["zref"])
        ter_kw = dict(slr=t["slope"], apr=t["aspect"], hor=t["hor"], svf=t["svfa"], wsa=t["wsa"])
    static = {k: as3d(v)[:, :, 0] for k, v in veg.items()} if layered else vegp
    a = prepare_grid_inputs(mp, reqhgt, static, soilc, dtm, pai_a=None if layered else pai_a, device=device, **ter_kw)
    if layered:                                                                 # .sortvegp2
        n = len(t_all)
        seld = sel["seld"] % 365 if vegpisannual else sel["seld"]
        nd = 365 if vegpisannual else int(round(n / 24))
        v14 = {}
        for k in VEG_KEYS:
            arr = as3d(veg[k])
            dmx = arr.shape[2]
            if dmx == 1:
                v14[k] = np.repeat(arr, 14, axis=2)
            else:
                sidx = layer_index(dmx, nd)
                sidx = np.concatenate([sidx, sidx[-1:]])
                v14[k] = arr[:, :, sidx[np.clip(seld, 1, len(sidx)) - 1] - 1]      # (seld = 0 would drop the day in R)
        pa = None if pai_a is None else intr(pai_a, mp["ntme"], mp["subs"])
        v14["leafden"], v14["paia"] = foliageden(reqhgt, v14["hgt"], v14["pai"], pa)
        a["vegp"] = v14
    mon2 = np.asarray(w2["obstime"]["month"]).astype(int)

    def selq(iq):                                                               # .getselq(iq, tme) - 1
        imn, imx = (12 if iq == 1 else iq - 1), (1 if iq == 12 else iq + 1)
        return np.sort(np.concatenate([np.nonzero(mon2 == m)[0] for m in (imn, iq, imx)]))
    args = {k: a[k] for k in ("obstime", "climdata", "pointm", "vegp", "soilc", "reqhgt", "zref", "lat", "lon", "Sminp", "Smaxp")}
    kw = dict(tfact=float(tfact), mat=a["mat"], out=[int(bool(v)) for v in out], wetq=selq(wq), dryq=selq(dq), hotq=selq(hq),
              colq=selq(cq), air=(temp == "air"))
    if _bioclim is not None:
        return _bioclim(layered, args, kw)
    fn = api.runbioclim3Cpp if layered else api.runbioclim1Cpp
    res = fn(**args, **kw, device=device)
    na = np.isnan(z)
    for v in res.values():
        v[na] = np.nan                                                          # mask(bior, dtm)
    return res


# ---- runmicro_big(): tiles of a large raster, one netCDF file each ---------------------------------------------------
def tile_size(nt: int, toverlap: int = 0) -> int:
    """the reference's automatic tile size (R/Cppwrappers.R:469-471): about 2e7 cell-steps per tile"""
    osize = np.sqrt(20000000 / nt) - 2 * toverlap
    sizeo = np.array([10, 20, 50, 100, 200, 500, 1000, 2000])
    return int(sizeo[np.argmin(np.abs(osize - sizeo))])


def tile_window(rw: int, cl: int, rows: int, cols: int, tilesize: int, toverlap: int):
    """`.croprast` (R/internal.R:1663-1676) in cell indices for unit-free overlap (the reference subtracts `toverlap` in
    map units): rows [r0, r1), cols [c0, c1) of tile (rw, cl), 1-based tile numbers, clipped to the raster"""
    r0, r1 = max((rw - 1) * tilesize - toverlap, 0), min(rw * tilesize + toverlap, rows)
    c0, c1 = max((cl - 1) * tilesize - toverlap, 0), min(cl * tilesize + toverlap, cols)
    return r0, r1, c0, c1


def runmicro_big(micropoint, reqhgt: float, pathout: str, vegp: Mapping, soilc: Mapping, dtm: Mapping, *,
                 tilesize: int | None = None, toverlap: int = 0, pai_a=None, tfact: float = 1.5,
                 vars: Sequence[str] | None = None, days_per_chunk: int = 5, device: int = 0, rank: int = 0,
                 world: int = 1, crows: int | None = None, ccols: int | None = None, lats=None, lons=None, dtmc=None,
                 altcorrect: int = 0) -> list:
    """`runmicro_big(micropoint, reqhgt, pathout, vegp, soilc, dtm, ..., writeasnc = TRUE)` for data.frame weather
    (R/Cppwrappers.R:446-541): slope, aspect, wetness index, horizons, sky view and wind shelter once for the WHOLE raster
    (wind shelter from the surface model dtm + hgt at 8 m, as there), then tile by tile the solver and
    `microut/area_RR_CC.nc`.  Each tile is solved in day chunks straight into its file (`pipeline.run_to_nc`): no tile's
    output ever exists on the host, and the tile size is only a file layout — the default keeps the reference's.
    `dtm` needs "xmin", "ymax" besides "z" / "res" / "lat" / "long" for the files' coordinates.  Returns the files written.
    With `world` > 1 (one process per GPU) the tiles are dealt round-robin: tile k goes to rank k % world; the universal
    variables are computed by every rank (seconds), no exchange is needed.
    (In the reference this function stops at an undefined `svfi`, R/Cppwrappers.R:520; what it sets out to do is done.)
    Array weather (`runmicro_big(micropointa, ..., dtm, dtmc, altcorrect)`, vignette "Running the model over large
    areas"): `micropoint` is `runpointmodela`'s list for a `crows` x `ccols` climate grid over the whole raster, `lats`,
    `lons` [rows, cols]; every tile interpolates the coarse arrays inside the solver at its own place in that grid, so
    tiles join without a seam in the forcing."""
    import os
    from . import pipeline
    if reqhgt < 0:
        raise ValueError("below ground the whole series has to be resident: use runmicro() per tile and writetonc()")
    z_all = np.asarray(dtm["z"], dtype=np.float64)
    rows, cols = z_all.shape
    res = dtm["res"] if np.isscalar(dtm["res"]) else dtm["res"][0]
    array = not isinstance(micropoint, Mapping)
    if array and (crows is None or ccols is None or lats is None or lons is None or len(micropoint) != crows * ccols):
        raise ValueError("array weather needs crows, ccols (matching the list of micropoints), lats and lons")
    first = micropoint[0] if array else micropoint
    nt = len(first["weather"]["temp"])
    ts = tile_size(nt, toverlap) if tilesize is None else int(tilesize)
    os.makedirs(os.path.join(pathout, "microut"), exist_ok=True)
    # universal variables, R/Cppwrappers.R:482-499
    ter = terrain.precompute_terrain(z_all, res, first["zref"], what=("slope", "aspect", "hor", "svfa"), device=device)
    slr, apr = ter["slope"], ter["aspect"]
    slr[np.isnan(z_all)] = np.nan
    apr[np.isnan(z_all)] = np.nan
    twi = terrain.topidx(z_all, res)
    dsm = z_all + np.nan_to_num(as3d(vegp["hgt"])[:, :, 0], nan=0.0)
    wsa = terrain.precompute_terrain(dsm, res, 8.0, what=("wsa",), device=device)["wsa"]
    written = []
    k_tile = 0
    for rw in range(1, -(-rows // ts) + 1):
        for cl in range(1, -(-cols // ts) + 1):
            r0, r1, c0, c1 = tile_window(rw, cl, rows, cols, ts, toverlap)
            zi = z_all[r0:r1, c0:c1]
            if np.count_nonzero(~np.isnan(zi)) <= 1:
                continue
            k_tile += 1
            if (k_tile - 1) % world != rank:
                continue
            crop = lambda a: np.asarray(a)[r0:r1, c0:c1]                           # noqa: E731
            dtmi = dict(dtm, z=zi)
            vegi, soili = {k: crop(v) for k, v in vegp.items()}, {k: crop(v) for k, v in soilc.items()}
            pre = dict(pai_a=None if pai_a is None else crop(pai_a), slr=crop(slr), apr=crop(apr), hor=crop(ter["hor"]),
                       twi=crop(twi), wsa=crop(wsa), svf=crop(ter["svfa"]), device=device)
            if array:
                a = prepare_grid_inputs_array(micropoint, crows, ccols, reqhgt, vegi, soili, dtmi, lats=crop(lats), lons=crop(lons),
                                              **pre)
                a["coarse"] = {"rowpos": api.coarse_positions(rows, crows)[r0:r1], "colpos": api.coarse_positions(cols, ccols)[c0:c1]}
                if altcorrect:
                    a["coarse"].update(altcorrect=int(altcorrect), dtmc=dtmc, dtm=cleanvars(vegi, soili, zi)[2])
            else:
                a = prepare_grid_inputs(micropoint, reqhgt, vegi, soili, dtmi, **pre)
            a["tfact"] = float(tfact)
            fo = os.path.join(pathout, "microut", f"area_{rw:02d}_{cl:02d}.nc")
            ext = {"xmin": dtm["xmin"] + c0 * res, "xmax": dtm["xmin"] + c1 * res, "ymin": dtm["ymax"] - r1 * res,
                   "ymax": dtm["ymax"] - r0 * res, "res": res, "crs": dtm.get("crs", "")}
            pipeline.run_to_nc(a, fo, ext, vars=vars, days_per_chunk=days_per_chunk, device=device, array_forcing=array)
            written.append(fo)
    return written
