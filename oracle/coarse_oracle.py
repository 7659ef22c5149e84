"""TEST INFRASTRUCTURE — what `.runmodel2Cpp` (R/internal.R:1175-1343, altcorrect = 0) does to its coarse climate and
point-model arrays before calling runmicro2Cpp, restated with numpy: bilinear resampling to the fine raster (terra's
`resample`, parity unpinned; here: the four neighbouring coarse cell centres, edge replication), then `.satvap`,
`.dewpoint`, wind speed from resampled components and the raster-mean wind direction.  The expanded arrays go to the
array-forcing oracle; the product (array_forcing == 2) never materialises them."""
import numpy as np

from microclimf_amd.synthetic import dewpoint_R, satvap_R


def upsample(a, rowpos, colpos):
    a = np.asarray(a, dtype=np.float64)
    cr, cc = a.shape[:2]
    fr, fc = np.floor(rowpos), np.floor(colpos)
    r0, c0 = fr.astype(int), fc.astype(int)
    r1, c1 = np.minimum(r0 + 1, cr - 1), np.minimum(c0 + 1, cc - 1)
    wy, wx = (rowpos - fr)[:, None, None], (colpos - fc)[None, :, None]
    top = (1.0 - wx) * a[np.ix_(r0, c0)] + wx * a[np.ix_(r0, c1)]
    bot = (1.0 - wx) * a[np.ix_(r1, c0)] + wx * a[np.ix_(r1, c1)]
    return np.asfortranarray((1.0 - wy) * top + wy * bot)


def lapserate(tc, ea, pk):
    """`.lapserate`, R/internal.R:545-550"""
    rv = 0.622 * ea / (pk - ea)
    return 9.8076 * (1 + (2501000 * rv) / (287 * (tc + 273.15))) / (1003.5 + (0.622 * 2501000 ** 2 * rv) / (287 * (tc + 273.15) ** 2))


def expand(climdata, pointm, rowpos, colpos, altcorrect=0, dtmc=None, dtm=None):
    """coarse {temp, relhum, pres, swdown, difrad, lwdown, windspeed, winddir} + pointm -> runmicro2Cpp's lists;
    altcorrect / dtmc / dtm: the altitudinal correction of R/internal.R:1233-1251"""
    up = lambda k: upsample(climdata[k], rowpos, colpos)          # noqa: E731
    tc = up("temp")
    es = satvap_R(tc)
    ea = es * up("relhum") / 100
    pk_fine = None
    if altcorrect:
        zc = np.nan_to_num(np.asarray(dtmc, dtype=np.float64), nan=0.0)[:, :, None]
        z = np.asarray(dtm, dtype=np.float64)[:, :, None]
        psl = upsample(climdata["pres"] / (((293 - 0.0065 * zc) / 293) ** 5.26), rowpos, colpos)
        pk_fine = np.asfortranarray(psl * (((293 - 0.0065 * z) / 293) ** 5.26))
        elevd = upsample(zc, rowpos, colpos) - z
        lr = 5 / 1000 if altcorrect == 1 else lapserate(tc, ea, pk_fine)
        tc_corrected = np.asfortranarray(lr * elevd + tc)
    wd = np.asarray(climdata["winddir"], dtype=np.float64) * np.pi / 180
    wu, wv = climdata["windspeed"] * np.cos(wd), climdata["windspeed"] * np.sin(wd)
    wuv, wvv = np.nanmean(wu, axis=(0, 1)), np.nanmean(wv, axis=(0, 1))
    with np.errstate(invalid="ignore", divide="ignore"):
        tdew = dewpoint_R(ea, tc)
    if altcorrect:
        tc = tc_corrected                                      # es, ea, tdew above are from the uncorrected field
    clim = {"tc": tc, "es": es, "ea": ea, "tdew": np.asfortranarray(tdew), "pk": pk_fine if altcorrect else up("pres"),
            "swdown": up("swdown"),
            "difrad": up("difrad"), "lwdown": up("lwdown"),
            "windspeed": np.asfortranarray(np.sqrt(upsample(wu, rowpos, colpos) ** 2 + upsample(wv, rowpos, colpos) ** 2)),
            "winddir": (np.arctan2(wvv, wuv) * 180 / np.pi) % 360}
    pm = {k: upsample(v, rowpos, colpos) for k, v in pointm.items()}
    return clim, pm
