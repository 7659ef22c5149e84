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


def meltmu2(mu, stemp, tc):
    """cpp:5495-5527"""
    mu = np.asarray(mu, dtype=np.float64)
    stemp = np.asarray(stemp, dtype=np.float64)
    tc = np.asarray(tc, dtype=np.float64)
    out = np.empty(mu.shape)
    for i, j in np.ndindex(mu.shape):
        if np.isnan(mu[i, j]):
            out[i, j] = np.nan
            continue
        dhp = dhm = 0.0
        for k in range(stemp.shape[2]):
            if stemp[i, j, k] > 0.0:
                dhp += stemp[i, j, k]
            s2 = (stemp[i, j, k] - tc[i, j, k]) * mu[i, j] + tc[i, j, k]
            if s2 > 0.0:
                dhm += s2
        out[i, j] = dhm / dhp if dhp > 0.0 else 0.5
    return out


def snowmodelq2_days(obstime, clim_c, pointm_c, pm2_c, subs, vegp, other, snowenv, dtm, dtmc, res, tfact, rowpos, colpos,
                     altcorrect=0):
    """`.snowmodelq2`, R/internal.R:3108-3283 (arguments as microclimf_amd.snow.snowmodelq2_days)"""
    from . import coarse_oracle as CO
    dtm = np.asarray(dtm, dtype=np.float64)
    R, Cc = dtm.shape
    nanmask = np.isnan(dtm)
    subs = np.asarray(subs, dtype=np.int64)
    n = subs.size
    zc = np.nan_to_num(np.asarray(dtmc, dtype=np.float64), nan=0.0)
    zref = float(other["zref"])

    def cca(a, mask=True):
        f = CO.upsample(a, rowpos, colpos)
        if mask:
            f[nanmask] = np.nan
        return f
    temp = cca(clim_c["temp"])
    relhum = cca(clim_c["relhum"])
    if altcorrect == 0:
        pres = cca(clim_c["pres"], False)
    else:
        ea = CO.satvap_R(temp) * relhum / 100
        psl = cca(np.asarray(clim_c["pres"]) / (((293 - 0.0065 * zc[:, :, None]) / 293) ** 5.26), False)
        pres = psl * (((293 - 0.0065 * dtm[:, :, None]) / 293) ** 5.26)
        elevd = (CO.upsample(zc[:, :, None], rowpos, colpos)[:, :, 0] - dtm)[:, :, None]
        tcdif = elevd * (5 / 1000) if altcorrect == 1 else CO.lapserate(temp, ea, pres) * elevd
        temp = tcdif + temp
        relhum = (ea / CO.satvap_R(temp)) * 100
    with np.errstate(invalid="ignore"):
        relhum[relhum > 100] = 100
    wd = np.asarray(clim_c["winddir"], dtype=np.float64) * np.pi / 180
    wu = np.asarray(clim_c["windspeed"]) * np.cos(wd)
    wv = np.asarray(clim_c["windspeed"]) * np.sin(wd)
    wuv, wvv = np.nanmean(wu, axis=(0, 1)), np.nanmean(wv, axis=(0, 1))
    clim = {"temp": temp, "relhum": relhum, "pres": pres, "difrad": cca(clim_c["difrad"]), "swdown": cca(clim_c["swdown"]),
            "lwdown": cca(clim_c["lwdown"]), "precip": cca(clim_c["precip"]),
            "windspeed": np.sqrt(cca(wu, False) ** 2 + cca(wv, False) ** 2), "winddir": (np.arctan2(wvv, wuv) * 180 / np.pi) % 360}
    pointm = {k: cca(pointm_c[k]) for k in ("Gp", "Tc", "RswabsG", "RlwabsG", "umu", "tr")}
    sstemp_f, tc_f = cca(pm2_c["sstemp"]), cca(pm2_c["tc"])
    vg = dict(vegp)
    vg["leaft"] = np.where(np.isnan(vg["leaft"]), 0.01, vg["leaft"])
    slope, aspect = TO.slope_aspect(dtm, res, aspect_na=180.0)
    hor = TO.horizons24(dtm, res)
    oth = dict(other)
    oth.update(slope=np.where(nanmask, np.nan, slope), aspect=np.where(nanmask, np.nan, aspect), hor=hor, skyview=TO.skyview(hor),
               wsa=TO.windsheltera(dtm, zref, 10 if res <= 100 else 1, res))
    snow_c = np.asarray(pm2_c["snow"], dtype=np.float64)
    pos = snow_c[snow_c > 0]
    msnow = pos.mean() if pos.size else np.nan
    mtemp = np.nanmean(tc_f)
    intfrac = canintfrac(vg["hgt"], vg["pai"], 2, msnow, mtemp, 0)
    isnowdc = np.array(other["isnowdc"], dtype=np.float64)
    isnowdg = (1 - intfrac) * isnowdc
    Tc = np.full((R, Cc, n), np.nan)
    Tg = Tc.copy(); sdepc = Tc.copy(); sden = Tc.copy()
    sdepg = np.zeros((R, Cc, n))
    ped = 0

    def resamplemelt(a, sbtn):
        return CO.upsample(np.asarray(a)[:, :, sbtn].sum(axis=2)[:, :, None], rowpos, colpos)[:, :, 0]
    with np.errstate(invalid="ignore", divide="ignore"):
        for day in range(n // 24):
            st = day * 24
            s = slice(st, st + 24)
            if subs[st] - 1 > 1:
                sbtn = _colon(ped + 1, int(subs[st]) - 1)
                mu = meltmu2(oth["skyview"], sstemp_f[:, :, sbtn], tc_f[:, :, sbtn])
                melt = resamplemelt(pm2_c["sublmelt"], sbtn) + resamplemelt(pm2_c["rainmelt"], sbtn) + mu * resamplemelt(pm2_c["tempmelt"], sbtn)
                snowsum = resamplemelt(pm2_c["snow"], sbtn)
                balancec = snowsum / 1000 - melt
                balanceg = (1 - intfrac) * snowsum / 1000 - np.exp(-np.asarray(vg["pai"])) * melt
                sdec = resamplemelt(pm2_c["sdenc"], sbtn) / len(sbtn)
                sdeg = resamplemelt(pm2_c["sdeng"], sbtn) / len(sbtn)
                isnowdc = isnowdc + balancec * (1000 / sdec)
                isnowdg = isnowdg + balanceg * (1000 / sdeg)
            isnowdc[isnowdc < 0] = 0
            isnowdg[isnowdg < 0] = 0
            oth["isnowdc"], oth["isnowdg"] = isnowdc, isnowdg
            c1 = {k: (np.asarray(v)[s] if k == "winddir" else np.asfortranarray(v[:, :, s])) for k, v in clim.items()}
            p1 = {k: np.asfortranarray(v[:, :, s]) for k, v in pointm.items()}
            smod = O.run_snowmodel({k: np.asarray(v)[s] for k, v in obstime.items()}, c1, p1, vg, oth, snowenv, array_forcing=True)
            dsnow = smod["sdepc"] - isnowdc[:, :, None]
            dsnowg = smod["sdepg"] - isnowdg[:, :, None]
            dsnowc = dsnow - dsnowg
            dtms = dtm + sdepg[:, :, st + 23]
            wss = np.sqrt(wuv[s] ** 2 + wvv[s] ** 2)
            af = int(np.round(10 * np.mean(wss) ** 0.5 / res))
            tpi = SD.tpicalc(af, min(R, Cc), dtms, tfact)
            dsnowg2 = dsnowg * tpi[:, :, None]
            sdc = dsnowc + dsnowg2 + isnowdc[:, :, None]
            sdg = dsnowg2 + isnowdg[:, :, None]
            sdc[sdc < 0] = 0
            sdg[sdg < 0] = 0
            Tc[:, :, s] = smod["Tc"]; Tg[:, :, s] = smod["Tg"]; sden[:, :, s] = smod["sden"]
            sdepc[:, :, s] = sdc
            sdepg[:, :, s] = sdg
            ped = int(subs[st + 23])
            isnowdc = sdc[:, :, 23].copy()
            isnowdg = sdg[:, :, 23].copy()
        out = {"Tc": Tc, "Tg": Tg, "groundsnowdepth": sdepg, "totalSWE": sdepc * sden, "snowden": sden, "umu": pointm["umu"]}
    for v in out.values():
        v[nanmask] = np.nan
    return out
