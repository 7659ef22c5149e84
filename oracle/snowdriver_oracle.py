"""Oracle of `.snowmodel1`'s chunk loop (TEST INFRASTRUCTURE): numpy restatement of
R/internal.R:2553-2617 on top of terrain_oracle.py (terrain refresh) and snow_oracle.c
(gridmodelsnow1), plus `.tpicalc` (R/internal.R:2471-2485).

PARITY UNPINNED for the terra parts (aggregate/resample/terrain), as in terrain_oracle.py: block
means from the top-left corner over non-NA cells (`na.rm = TRUE`), bilinear interpolation between
block centres clamped at the outermost centres.  R-level behaviours kept: `other$isnowdg` is never
updated inside the loop; `round()` is half-to-even; `tpic[tpic < 0.05] <- 0.1`; `1:n5days`
truncates (at least one chunk); arrays beyond the last chunk stay NA.
"""
from __future__ import annotations

import numpy as np

from . import oracle as O
from . import terrain_oracle as TO


def block_mean_narm(a, s: int):
    x, y = a.shape
    nx, ny = -(-x // s), -(-y // s)
    out = np.empty((nx, ny))
    for i in range(nx):
        for j in range(ny):
            blk = a[i * s:(i + 1) * s, j * s:(j + 1) * s]
            ok = ~np.isnan(blk)
            out[i, j] = blk[ok].sum() / ok.sum() if ok.any() else np.nan
    return out


def tpicalc(af: int, me: int, dtm, tfact: float):
    """.tpicalc, R/internal.R:2471-2485."""
    dtm = np.asarray(dtm, dtype=np.float64)
    if af < me / 2:
        dtmc = TO.bilinear_from_blocks(block_mean_narm(dtm, af), af, *dtm.shape)
    else:
        dtmc = dtm * 0 + np.nanmean(dtm)
    tpi = dtmc - dtm
    with np.errstate(invalid="ignore"):
        tpic = np.exp(tpi * tfact)
        tpic[tpic < 0.05] = 0.1
        tpic[tpic > 10] = 10
    return tpic / np.nanmean(tpic)


def snowmodel1_chunks(obstime, climdata, pointm, vegp, other, snowenv, dtm, res, tfact=0.02, chunk_steps=120, handover=None):
    """`handover(chunk, isnowdc)` (optional) may return a replacement for the pack depth handed to the next chunk — the hook
    by which a test aligns the loop's ill-conditioned gate (`sdepcp > 0` on a rounding residue) with another run's."""
    dtm = np.asarray(dtm, dtype=np.float64)
    R, Cc = dtm.shape
    h = len(np.asarray(obstime["year"]))
    nch = max(1, h // chunk_steps)
    na = O.load().orc_na_real()
    outs = {k: np.full((R, Cc, h), na, order="F") for k in ("Tc", "Tg", "groundsnowdepth", "totalSWE", "snowden")}
    oth = dict(other)
    isnowdg = np.asarray(other["isnowdg"], dtype=np.float64)
    dtms = dtm + isnowdg
    zref = float(other["zref"])
    ss = 10 if res <= 100 else 1
    nanmask = np.isnan(dtm)
    for ch in range(nch):
        st, ed = ch * chunk_steps, min((ch + 1) * chunk_steps, h)
        slope, aspect = TO.slope_aspect(dtms, res, aspect_na=180.0)
        slope = np.where(nanmask, np.nan, slope)
        aspect = np.where(nanmask, np.nan, aspect)
        hor = TO.horizons24(dtms, res)
        oth.update(slope=slope, aspect=aspect, hor=hor, skyview=TO.skyview(hor),
                   wsa=TO.windsheltera(dtms, zref, ss, res))
        sl = slice(st, ed)
        smod = O.run_snowmodel({k: np.asarray(v)[sl] for k, v in obstime.items()},
                               {k: np.asarray(v)[sl] for k, v in climdata.items()},
                               {k: np.asarray(v)[sl] for k, v in pointm.items()}, vegp, oth, snowenv)
        tpr = 10 * np.mean(np.asarray(climdata["windspeed"])[sl]) ** 0.5
        af = int(np.round(tpr / res))                     # numpy rounds half to even, like R
        tpi = tpicalc(af, min(R, Cc), dtms, tfact)[:, :, None]
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
            if handover is not None:
                oth["isnowdc"] = handover(ch, oth["isnowdc"])
            oth["isnowac"] = np.nan_to_num(smod["agec"])
            oth["isnowag"] = np.nan_to_num(smod["ageg"])
            dtms = dtm + outs["groundsnowdepth"][:, :, ed - 1]
    return outs
