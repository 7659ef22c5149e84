"""Oracle of the second half of `.snowmodel2` (TEST INFRASTRUCTURE): numpy restatement of R/internal.R:2862-3013 — `.cca`
resampling and the altitude correction (coarse_oracle.py), terrain of dtm + ground snow (terrain_oracle.py), gridmodelsnow2
(snow_oracle.c), `.tpicalc` (snowdriver_oracle.py), redistribution and hand-over.  PARITY UNPINNED for the terra parts
(resample / aggregate / terrain), as in those files; no reference test or published figure covers this route.
R-level behaviours kept: `other$isnowdg` is never updated; `if (af < 2) af <- 2`; `vegp$leaft[is.na] <- 0.001`;
`relhum > 100 <- 100`; `.cleansmod` masks every returned array by the dtm."""
from __future__ import annotations

import numpy as np

from . import coarse_oracle as CO
from . import oracle as O
from . import snowdriver_oracle as SD
from . import terrain_oracle as TO


def snowmodel2_chunks(obstime, clim_c, pointm_c, vegp, other, snowenv, dtm, dtmc, res, tfact, rowpos, colpos, altcorrect=0,
                      agg=10, chunk_steps=120):
    dtm = np.asarray(dtm, dtype=np.float64)
    R, Cc = dtm.shape
    h = len(np.asarray(obstime["year"]))
    nanmask = np.isnan(dtm)
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
        es = CO.satvap_R(temp)
        ea = es * relhum / 100
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
    vg = dict(vegp)
    vg["leaft"] = np.where(np.isnan(vg["leaft"]), 0.001, vg["leaft"])
    oth = dict(other)
    isnowdg = np.asarray(other["isnowdg"], dtype=np.float64)
    dtms = dtm + isnowdg
    na = np.nan
    outs = {k: np.full((R, Cc, h), na) for k in ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden")}
    for ch in range(h // chunk_steps):
        st, ed = ch * chunk_steps, min((ch + 1) * chunk_steps, h)
        sl = slice(st, ed)
        slope, aspect = TO.slope_aspect(dtms, res, aspect_na=180.0)
        hor = TO.horizons24(dtms, res)
        oth.update(slope=np.where(nanmask, np.nan, slope), aspect=np.where(nanmask, np.nan, aspect), hor=hor,
                   skyview=TO.skyview(hor), wsa=TO.windsheltera(dtms, zref, agg, res))
        c1 = {k: (np.asarray(v)[sl] if k == "winddir" else np.asfortranarray(v[:, :, sl])) for k, v in clim.items()}
        p1 = {k: np.asfortranarray(v[:, :, sl]) for k, v in pointm.items()}
        smod = O.run_snowmodel({k: np.asarray(v)[sl] for k, v in obstime.items()}, c1, p1, vg, oth, snowenv, array_forcing=True)
        wss = np.sqrt(wuv[sl] ** 2 + wvv[sl] ** 2)
        af = int(np.round(10 * np.mean(wss) ** 0.5 / res))
        if af < 2:
            af = 2
        tpi = SD.tpicalc(af, min(R, Cc), dtms, tfact)[:, :, None]
        with np.errstate(invalid="ignore"):
            asd = isnowdg[:, :, None]
            dsnow = smod["sdepg"] - asd
            dsnow2 = np.where(dsnow < 0, dsnow, dsnow * tpi)
            asc = np.asarray(oth["isnowdc"], dtype=np.float64)[:, :, None]
            cdsnow = smod["sdepc"] - asc - dsnow
            tot = asc + cdsnow + dsnow2
            outs["Tc"][:, :, sl] = smod["Tc"]
            outs["Tg"][:, :, sl] = smod["Tg"]
            outs["totalSWE"][:, :, sl] = tot * smod["sden"]
            outs["groundsnowdepth"][:, :, sl] = asd + dsnow2
            outs["snowden"][:, :, sl] = smod["sden"]
        oth["isnowdc"] = tot[:, :, -1]
        oth["isnowac"] = np.nan_to_num(smod["agec"])
        oth["isnowag"] = np.nan_to_num(smod["ageg"])
        dtms = dtm + outs["groundsnowdepth"][:, :, ed - 1]
    outs["umu"] = pointm["umu"].copy()                     # `umu = pointm$umu`: the whole series, beyond the last chunk too
    for k in outs:
        outs[k][nanmask] = np.nan
    return outs
