"""Oracle of the reference's fast snow method `.snowmodelq1` (TEST INFRASTRUCTURE): numpy restatement of the day loop
(R/internal.R:2690-2776) on top of terrain_oracle.py, snow_oracle.c (gridmodelsnow1) and snowdriver_oracle.tpicalc, with
`canintfrac` / `meltmu` / `canopysnowintCpp` restated from src/microclimfCpp.cpp:5417-5492, 3713-3739.

PARITY: the terra parts are unpinned as in terrain_oracle.py; the chain as a whole is held against the red curve of the
reference's vignette figure image14p (tests/test_frontend_gpu.py).  R-level behaviours kept: `a:b` counts down when
a > b; `dtms <- dtm + sdepg[,,ed]` reads zeros (the day's depths are stored afterwards); snow ages are not handed on;
`snowinitd * dtm` as the initial depth; `x[x < 0] <- 0` leaves NA alone.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O
from . import snowdriver_oracle as SD
from . import terrain_oracle as TO


def canopysnowint(hgt, pai, uf, prec, tc, Li, Sh=6.2):
    """canopysnowintCpp, cpp:3713-3739"""
    hgt = max(hgt, 0.001)
    pai = max(pai, 0.001)
    Be = np.sqrt(0.003 + (0.2 * pai) / 2.0)
    uh = uf / Be
    a = pai / hgt
    Lc = (0.25 * a) ** -1.0
    Lm = 2.0 * Be ** 3.0 * Lc
    k1 = Be / Lm
    uzm = (uh / (hgt * k1)) * (1 - np.exp(-k1 * hgt))
    uzm = max(uzm, uf)
    rhos = 67.92 + 51.25 * np.exp(tc / 2.59)
    S = Sh * (0.26 + 46 / rhos)
    Lstr = S * pai
    Z = np.arctan(uzm / 0.8)
    kc = 1.0 / (2.0 * np.cos(Z))
    Cp = 1.0 - np.exp(-kc * pai)
    k2 = Cp / Lstr
    I1 = (Lstr - Li) * (1.0 - np.exp(-k2 * prec))
    return min(I1 * 0.678, prec)


def canintfrac(hgt, pai, uf, prec, tc, Li):
    """cpp:5417-5450"""
    hgt = np.asarray(hgt, dtype=np.float64)
    pai = np.asarray(pai, dtype=np.float64)
    frac = np.empty(hgt.shape)
    for idx in np.ndindex(hgt.shape):
        if np.isnan(hgt[idx]):
            frac[idx] = np.nan
        elif prec > 0.0:
            frac[idx] = canopysnowint(hgt[idx], pai[idx], uf, prec, tc, Li) / prec
        else:
            frac[idx] = 0.5
    return frac


def meltmu(skyview, stemp, tc):
    """cpp:5454-5492"""
    skyview = np.asarray(skyview, dtype=np.float64)
    stemp = np.asarray(stemp, dtype=np.float64)
    tc = np.asarray(tc, dtype=np.float64)
    dhp = 0.0
    for v in stemp:
        if v > 0.0:
            dhp += v
    mu = np.ones(skyview.shape)
    if dhp > 0.0:
        for idx in np.ndindex(skyview.shape):
            if np.isnan(skyview[idx]):
                mu[idx] = np.nan
                continue
            dhm = 0.0
            for k in range(stemp.size):
                s2 = (stemp[k] - tc[k]) * skyview[idx] + tc[k]
                if s2 > 0.0:
                    dhm += s2
            mu[idx] = dhm / dhp
    return mu


def _colon(a, b):
    return (np.arange(a, b + 1) if a <= b else np.arange(a, b - 1, -1)) - 1


def snowmodelq1_days(obstime, climdata, pointm, pmod, temp_all, snow_all, subs, vegp, other, snowenv, dtm, res, tfact=0.02):
    dtm = np.asarray(dtm, dtype=np.float64)
    R, Cc = dtm.shape
    subs = np.asarray(subs, dtype=np.int64)
    n = subs.size
    zref = float(other["zref"])
    nanmask = np.isnan(dtm)
    slope, aspect = TO.slope_aspect(dtm, res, aspect_na=180.0)
    hor = TO.horizons24(dtm, res)
    oth = dict(other)
    oth.update(slope=np.where(nanmask, np.nan, slope), aspect=np.where(nanmask, np.nan, aspect), hor=hor, skyview=TO.skyview(hor),
               wsa=TO.windsheltera(dtm, zref, 10 if res <= 100 else 1, res))
    snow_all = np.asarray(snow_all, dtype=np.float64)
    temp_all = np.asarray(temp_all, dtype=np.float64)
    pos = snow_all[snow_all > 0]
    msnow = pos.mean() if pos.size else np.nan
    mtemp = np.mean(np.asarray(climdata["temp"], dtype=np.float64))
    intfrac = canintfrac(vegp["hgt"], vegp["pai"], 2, msnow, mtemp, 0)
    isnowdc = np.array(other["isnowdc"], dtype=np.float64)
    isnowdg = (1 - intfrac) * isnowdc
    na = np.nan
    Tc = np.full((R, Cc, n), na)
    Tg = Tc.copy(); sdepc = Tc.copy(); sden = Tc.copy()
    sdepg = np.zeros((R, Cc, n))
    ped = 0
    sbtn = None
    with np.errstate(invalid="ignore"):
        for day in range(n // 24):
            st = day * 24
            s = slice(st, st + 24)
            if subs[st] - 1 > 1:
                sbtn = _colon(ped + 1, int(subs[st]) - 1)
                mu = meltmu(oth["skyview"], pmod["sstemp"][sbtn], temp_all[sbtn])
                melt = np.sum(pmod["sublmelt"][sbtn]) + np.sum(pmod["rainmelt"][sbtn]) + mu * np.sum(pmod["tempmelt"][sbtn])
                balancec = np.sum(snow_all[sbtn] / 1000) - melt
                balanceg = (1 - intfrac) * np.sum(snow_all[sbtn] / 1000) - np.exp(-np.asarray(vegp["pai"])) * melt
            else:
                balancec = balanceg = 0.0
            isnowdc = isnowdc + balancec * (1000 / np.mean(pmod["sdenc"][sbtn]))
            isnowdg = isnowdg + balanceg * (1000 / np.mean(pmod["sdeng"][sbtn]))
            isnowdc[isnowdc < 0] = 0
            isnowdg[isnowdg < 0] = 0
            oth["isnowdc"], oth["isnowdg"] = isnowdc, isnowdg
            smod = O.run_snowmodel({k: np.asarray(v)[s] for k, v in obstime.items()},
                                   {k: np.asarray(v)[s] for k, v in climdata.items()},
                                   {k: np.asarray(v)[s] for k, v in pointm.items()}, vegp, oth, snowenv)
            dsnow = smod["sdepc"] - isnowdc[:, :, None]
            dsnowg = smod["sdepg"] - isnowdg[:, :, None]
            dsnowc = dsnow - dsnowg
            dtms = dtm + sdepg[:, :, st + 23]
            tpr = 10 * np.mean(np.asarray(climdata["windspeed"])[s]) ** 0.5
            af = int(np.round(tpr / res))
            tpi = SD.tpicalc(af, min(R, Cc), dtms, tfact)
            dsnowg2 = dsnowg * tpi[:, :, None]
            dsnowc2 = dsnowc + dsnowg2
            sdc = dsnowc2 + isnowdc[:, :, None]
            sdg = dsnowg2 + isnowdg[:, :, None]
            sdc[sdc < 0] = 0
            sdg[sdg < 0] = 0
            Tc[:, :, s] = smod["Tc"]; Tg[:, :, s] = smod["Tg"]; sden[:, :, s] = smod["sden"]
            sdepc[:, :, s] = sdc
            sdepg[:, :, s] = sdg
            ped = int(subs[st + 23])
            isnowdc = sdc[:, :, 23].copy()
            isnowdg = sdg[:, :, 23].copy()
    return {"Tc": Tc, "Tg": Tg, "groundsnowdepth": sdepg, "totalSWE": sdepc * sden, "snowden": sden}
